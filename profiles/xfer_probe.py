import os, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
a = torch.full((1024,), float(rank + 1), device=dev)
b = torch.zeros(1024, device=dev)
try:
    ops = [dist.P2POp(dist.isend, a, 1 - rank), dist.P2POp(dist.irecv, b, 1 - rank)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    print(rank, "p2p cuda over gloo ok", float(b[0]))
except Exception as e:
    print(rank, "p2p cuda over gloo FAILED", type(e).__name__, str(e)[:200])
dist.barrier()
