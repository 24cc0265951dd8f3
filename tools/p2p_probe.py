import os, sys, torch, torch.distributed as dist
mode = sys.argv[1]
if mode == "self":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1)
    a = torch.arange(12, dtype=torch.float32, device="cuda").view(3, 4)
    b = torch.zeros(3, 4, device="cuda")
    ops = [dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, a, 0)]
    for r in dist.batch_isend_irecv(ops): r.wait()
    torch.cuda.synchronize()
    print("self send ok:", torch.equal(a, b))
    dist.destroy_process_group()
else:
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", rank, "allreduce on shared device:", x.tolist())
    dist.destroy_process_group()
