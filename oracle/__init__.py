"""Test infrastructure: CPU oracle of the TeMP snapshot-encoder hot path (see temp_oracle.py).
Never imported by the product package `temp_amd`."""
