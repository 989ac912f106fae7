#!/usr/bin/env python3
"""Can two ranks of an RCCL communicator share ONE GPU on this pool?  (NCCL proper refuses: "Duplicate GPU detected".)
    python3 profiles/rccl_same_gpu_probe.py            parent: starts two ranks of itself
If it works, the whole N = 2 code path (RcclComm, rgbdr_halo_exchange_async, rgbdr_shard_allgather between two processes)
can be exercised on the one-GPU boxes."""
import os
import subprocess
import sys

if "RANK" not in os.environ:
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        for k, v in (a.split("=", 1) for a in sys.argv[1:]):
            env[k] = v
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        try:
            rc = p.wait(timeout=180) or rc
        except subprocess.TimeoutExpired:
            p.kill()
            rc = 124
    sys.exit(rc)

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
x = torch.full((1024,), float(rank + 1), device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank %d all_reduce -> %.1f" % (rank, x[0].item()), flush=True)
y = torch.full((1 << 20,), float(rank), device=dev)
z = torch.empty_like(y)
ops = [dist.P2POp(dist.isend, y, 1 - rank), dist.P2POp(dist.irecv, z, 1 - rank)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
print("rank %d send/recv -> %.1f" % (rank, z[0].item()), flush=True)
dist.destroy_process_group()
