"""The C++ host mirror (rgbd-recon_amd/host/rgbdr_host.hpp: CalibVolumes /
NetKinectArray / ReconIntegration over the C ABI) driven by the example frame
loop, fed with calibration volumes and a recorded .stream frame in the reference's
on-disk formats."""
import os
import subprocess

import numpy as np
import pytest

from conftest import count_diff, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rgbd-recon_amd", "host", "frame_loop")


def test_frame_loop_is_built_and_linked_against_the_c_abi():
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "librgbdr_hip.so" in out and "not found" not in out
    assert subprocess.run([EXE], capture_output=True).returncode == 2      # usage


@pytest.mark.gpu
def test_frame_loop_matches_oracle(pkg, orc, tmp_path):
    synth = pkg.synth
    n, W, H, G = 2, 64, 53, 32
    scene = synth.Scene(n, W, H, lut_res=(16, 13, 16), seed=77)
    inv = scene.inverse((G, G, G))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "recordings"))
    for i in range(n):
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz" % i), scene.xyz[i], 3) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_uv" % i), scene.uv[i], 2) == 0
        assert orc.lut_write(os.path.join(d, "s%d.cv_xyz_inv" % i), inv[i], 4) == 0
        # .stream: frames of [colorsize][depthsize] (NetKinectArray.cpp:745-763); two frames, the first is read
        with open(os.path.join(d, "recordings", "s%d.stream" % i), "wb") as f:
            for k in range(2):
                f.write(scene.color[i].tobytes())
                f.write((scene.depth[i] if k == 0 else np.zeros_like(scene.depth[i])).tobytes())
    out = os.path.join(d, "out.tsdf")
    r = subprocess.run([EXE, d, str(n), str(W), str(H), str(G), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("res 32 32 32 bricks 64")
    got = np.fromfile(out, dtype=np.float32).reshape(G, G, G)
    g = pkg.capi.compute_geometry(pkg.capi.make_config(n, (W, H), voxel_size=2.0 / G, brick_size=8 * 2.0 / G))
    ref = orc.run_pipeline(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, brick_size=g.brick_size,
                           bv=g.brick_voxels, res_bricks=tuple(g.res_bricks))
    assert same_bits(got, ref["tsdf"]), "%d voxels differ" % count_diff(got, ref["tsdf"])
