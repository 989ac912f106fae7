"""Developer probe: ray-march the benchmark scene (4 sensors -> 512^3) at 1280x720,
time the kernel and write PNGs of the colour / shaded / normal views."""
import sys, time, os, zlib, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from rgbd_recon_amd import capi, synth
import numpy as np

def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

N, W, H = 4, 512, 424
G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
scene = synth.Scene(N, W, H, lut_res=(128, 106, 128))
ctx = capi.Context(capi.make_config(N, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G), 0)
for i in range(N):
    ctx.set_calibration(i, scene.xyz[i], scene.lut_res, scene.uv[i], scene.lut_res, (0.5, 4.5))
    ctx.synth_inverse_calibration(i, scene.pinhole(i))
ctx.step(scene.depth, scene.color)
ctx.enable_timers(True)
OUTD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUTD, exist_ok=True)
for mode, name in ((0, "color"), (1, "shaded"), (2, "normal")):
    view = capi.make_view((2.4, 1.8, 2.1), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0), 45.0, 1280, 720, synth.BBOX_MIN, synth.BBOX_MAX, shade_mode=mode)
    color, depth, ns = ctx.raymarch(view)
    color, depth, ns = ctx.raymarch(view)
    print(name, "draw %.3f ms" % (ctx.timer_ns("draw") * 1e-6), "hit %.3f" % (depth < 1).mean(), "mean samples/ray %.1f" % (ns.mean() / 0.0027))
    img = color[..., :3].copy()
    if mode == 2: img = img * 0.5 + 0.5
    img[depth >= 1] = (0.1, 0.1, 0.12)
    write_png(OUTD + "/raymarch_%s_%d.png" % (name, G), (np.clip(img[::-1], 0, 1) * 255).astype(np.uint8))
