"""Developer probes for the transports under rgbd_recon_amd.dist:
  (default, 2 ranks, gloo)  point-to-point on device tensors works but is not stream-ordered
  (arg 'self', 1 rank, nccl) does RCCL accept a send/recv pair to the own rank?"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "self":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    a = torch.arange(1024, device=dev, dtype=torch.float32)
    b = torch.zeros(1024, device=dev)
    try:
        ops = [dist.P2POp(dist.isend, a, rank), dist.P2POp(dist.irecv, b, rank)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        torch.cuda.synchronize()
        print("self send/recv over RCCL ok", bool(torch.equal(a, b)))
    except Exception as e:
        print("self send/recv over RCCL FAILED", type(e).__name__, str(e)[:300])
    dist.destroy_process_group()
    sys.exit(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
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
