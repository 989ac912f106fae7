"""Creates and destroys N small contexts (1 / 3 / 5 / 8 sensors in turn), runs one frame in each and compares every image
with the oracle: the loop that exposed the intermittent corruption / hang with physically contiguous LUT arenas
(profiles/r03_notes).  Test infrastructure (loads the oracle); run on the GPU box under `timeout`:
    timeout 200 python3 profiles/context_churn_probe.py 40"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from __graft_entry__ import load_oracle, load_package
orc = load_oracle(); load_package()
from rgbd_recon_amd import capi, synth
import test_parity_gpu as T
class P: pass
pkg = P(); pkg.capi, pkg.synth = capi, synth
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    n = [1, 3, 5, 8][it % 4]
    scene, ctx, inv = T.build(pkg, n=n, wh=(64, 53), G=32, lut_res=(16, 13, 16))
    ctx.step(scene.depth, scene.color)
    ref = T.oracle_run(orc, scene, ctx, inv)
    for name, which in T.IMG.items():
        for i in range(n):
            got = ctx.readback_image(which, i)
            r = ref[name][i]
            d = ~((got == r) | (np.isnan(got) & np.isnan(r)))
            if d.any():
                bad += 1
                idx = np.argwhere(d)
                print("iter", it, "n", n, name, "sensor", i, "differ", int(d.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(), "got", got[tuple(idx[0])], "want", r[tuple(idx[0])])
                again = ctx.readback_image(which, i)
                print("   re-read equal to first read:", bool(np.array_equal(again, got, equal_nan=True)), " re-read equals oracle:", bool(np.all((again == r) | (np.isnan(again) & np.isnan(r)))))
    ctx.close()
print("bad images", bad)
