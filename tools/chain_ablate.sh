#!/bin/bash
# GPU box: time of the chain kernels with parts switched off.  TEMP_CHAIN_DBG (forward) bits: 1 no weight loads, 2 no MFMAs,
# 4 no HBM stores, 8 no input-gate loads, 16 no gate arithmetic; TEMP_CHAIN_DBG_BWD bits: 1 no weight loads, 2 no MFMAs, 4 no HBM
# stores, 8 no saved-plane loads.  Results are wrong by design; only the kernel times matter.
which=${1:-fwd}
if [ "$which" = fwd ]; then
  for d in 0 1 2 3 4 8 12 28 31; do
    echo "== TEMP_CHAIN_DBG=$d"
    TEMP_CHAIN_DBG=$d python bench.py --kernel-table --no-cpu-baseline --train-loop-steps 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "k_gru_chain_fwd"
  done
else
  for d in 0 1 2 3 4 8 12 15; do
    echo "== TEMP_CHAIN_DBG_BWD=$d"
    TEMP_CHAIN_DBG_BWD=$d python bench.py --kernel-table --no-cpu-baseline --train-loop-steps 0 --steps 10 --warmup 3 2>&1 >/dev/null | grep "k_gru_chain_bwd"
  done
fi
