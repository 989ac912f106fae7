"""Developer probe: thread scaling of the CPU oracle's integrate on the GPU box's host."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package, load_oracle
load_package(); orc = load_oracle()
from rgbd_recon_amd import synth
import numpy as np
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a",
      "affinity:", len(os.sched_getaffinity(0)))
N, W, H, G, rows = 4, 512, 424, 256, 32
scene = synth.Scene(N, W, H, lut_res=(64, 53, 64))
inv = [synth.inverse_lut(s, (G, G, rows)) for s in scene.sensors]
sil = [np.ones((H, W), np.float32) for _ in range(N)]
db = [np.full((H, W, 2), 0.5, np.float32) for _ in range(N)]
q = [np.ones((H, W), np.float32) for _ in range(N)]
for t in (1, 8, 32, 64, 128, 256):
    orc.set_threads(t)
    orc.integrate(inv, sil, db, q, (G, G, rows), 0.01)
    t0 = time.perf_counter()
    for _ in range(3):
        orc.integrate(inv, sil, db, q, (G, G, rows), 0.01)
    dt = (time.perf_counter() - t0) / 3
    print("threads %3d: %.3f s  %.1f Mvox/s" % (t, dt, G * G * rows / dt / 1e6))
