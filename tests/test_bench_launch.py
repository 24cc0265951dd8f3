"""`python bench.py --gpus N` without a launcher starts its own N ranks (the reference's axis is
Lightning DDP, models/TKG_Module.py:162-179, launcher_2gpu.sh:8)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_launch_command_is_torchrun_on_localhost():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_plain_gpus2_forms_a_two_rank_group():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""          # gloo leg also on a GPU box
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["rendezvous"] == 2 and out["rank_sum"] == 1.0 and out["gpus"] == 2
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
