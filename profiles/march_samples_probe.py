import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
N, W, H = 4, 512, 424
os.environ.setdefault("RGBDR_ARENA_TRIALS", "1")
sc = synth.Scene(N, W, H, lut_res=(128, 106, 128), seed=1234)
for G in (os.environ.get("RGBDR_DISPLAY_GRID", "512"),):   # the two grids of profiles/display_frame_only.py
    if G == "ref":
        bmax = (1.0, 2.2, 1.0)
        rc = capi.Context(capi.make_config(N, (W, H), bbox_max=bmax, voxel_size=0.01, brick_size=0.1), 0)
    else:
        bmax = synth.BBOX_MAX
        rc = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / 512, brick_size=8 * 2.0 / 512), 0)
    for i in range(N):
        rc.set_calibration(i, sc.xyz[i], sc.lut_res, sc.uv[i], sc.lut_res, (0.5, 4.5))
        if G == "ref":
            rc.set_inverse_calibration(i, rc.generate_inverse_lut(i, (286, 315, 286)), (286, 315, 286))
        else:
            rc.synth_inverse_calibration(i, sc.pinhole(i))
    rc.step(sc.depth, sc.color)
    view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, bmax)
    view.skip_space = 1
    col, dep, ns = rc.raymarch(view)
    num = np.rint(ns / 0.0027).astype(np.int64)
    hit = dep < 1
    print("pixels", num.size, "hit", hit.mean(), "zero-sample", (num == 0).mean())
    print("samples: total %.3g, by hit rays %.3g, by miss rays %.3g" % (num.sum(), num[hit].sum(), num[~hit].sum()))
    for q in (50, 90, 99, 99.9, 100):
        print("  percentile %5.1f of nonzero rays: %d samples (hit rays: %d, miss rays: %d)" % (q, np.percentile(num[num > 0], q), np.percentile(num[hit], q), np.percentile(num[(~hit) & (num > 0)], q) if ((~hit) & (num > 0)).any() else 0))
    # per 8x8 wave square: max samples in the square (what the wavefront runs for)
    sq = num[:720, :1280].reshape(90, 8, 160, 8).max(axis=(1, 3))
    print("wave squares: %d, with work %d; sum of per-square max %.3g; max %d" % (sq.size, (sq > 0).sum(), sq.sum(), sq.max()))
    blk = num[:720, :1280].reshape(90, 8, 160, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    order = np.argsort(-blk.max(axis=1))
    print("the ten longest wave squares: (max samples, lanes with at least half of it, lanes with any, median of the marching lanes)")
    for k in order[:10]:
        b = blk[k]
        print("   ", int(b.max()), int((b >= b.max() / 2).sum()), int((b > 0).sum()), int(np.median(b[b > 0])))
    rounds = np.ceil(blk.max(axis=1) / 8).astype(int)
    hist = np.bincount(rounds)
    print("squares by rounds of 8 samples:", {int(r): int(c) for r, c in enumerate(hist) if c and r > 0})
    for cap in (16, 32, 64):
        left = num[num > cap] - cap
        print("rays with more than %d samples: %d; what they have left: median %d, 90 %% %d, max %d, total %.3g" %
              (cap, left.size, np.median(left) if left.size else 0, np.percentile(left, 90) if left.size else 0, left.max() if left.size else 0, left.sum()))
    lane_rounds = np.ceil(blk / 8).sum()
    print("rounds: %d summed over the squares (what the SIMDs issue), %.0f summed over the lanes / 64 (what the rays need): lane use %.2f"
          % (rounds.sum(), lane_rounds / 64, lane_rounds / 64 / rounds.sum()))
    print("  issue bound of the march at I instructions per round: rounds x I x 4 cycles / 1024 SIMDs / 2.4 GHz = %.1f us per 1000 instructions"
          % (rounds.sum() * 1000 * 4 / 1024 / 2.4e3))
    # with rounds of 8 per lane until at most 8 lanes are left, then the lanes left share the wavefront (64 / n samples per round)
    est = []
    for b in blk[order[:200]]:
        s_ = np.sort(b[b > 0])[::-1]
        if s_.size == 0:
            continue
        r, done = 0, 0
        # lanes sorted by length: after k rounds of 8 the lanes with more than 8 k samples are still marching
        k = 0
        while (s_ > 8 * k).sum() > 8:
            k += 1
        left = s_[s_ > 8 * k] - 8 * k
        coop = 0
        while left.size:
            per = 64 // left.size
            left = left - per
            left = left[left > 0]
            coop += 1
        est.append((k, coop))
    print("longest squares: (per-lane rounds, then cooperative rounds) for the ten longest:", est[:10], " worst total:", max(a + b for a, b in est))
    rc.close()
