"""Several ranks, one Z slab each, end to end through rgbd_recon_amd.dist: integrate,
halo exchange, slab ray-march with first-hit MIN and frame compositing.  The box has
one GPU, so the ranks share cuda:0 and use gloo with host staging; on a multi-GPU node
the same functions run over RCCL (bench.py --gpus N)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,G,limit", [(2, 64, 0.03), (3, 96, 0.05)])
def test_slab_ranks_render_the_whole_volume(world, G, limit, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "slab_worker.py"), str(tmp_path), str(G),
           str(limit)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for n in range(2):
        ref = np.load(os.path.join(str(tmp_path), "whole_v%d.npz" % n))
        assert (ref["depth"] < 1).mean() > 0.02
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), "slab_r%d_v%d.npz" % (rank, n)))
            for key in ("color", "depth", "ns"):
                assert same_bits(got[key], ref[key]), (n, rank, key)
