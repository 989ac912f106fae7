#!/usr/bin/env python3
"""Writes tests/golden/shader_passes_<case>.npz: what the TEXT of the reference's pre_* / tsdf_integration shaders,
compiled as C++ (oracle/build_shader_ref.py -> oracle/_ref/libref_shaders.so, build container only), produces for the
scenes of shader_cases.py.  The fixtures are data (images, counters, volumes); no reference text is stored.
They are NOT output of a run of the reference: texture sampling and the driver-defined built-ins are stand-ins bound
to the oracle's conventions (oracle/glsl_runtime.hpp).  What they add over the oracle's own golden files is that
every arithmetic statement between two fetches was executed from the reference's text.

    python tests/golden/make_shader_golden.py        (needs /root/reference; `make -C oracle shaders` first)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), HERE]
from __graft_entry__ import load_oracle, load_package  # noqa: E402

load_oracle()
load_package()
import shader_cases  # noqa: E402
import shader_ref  # noqa: E402
from rgbd_recon_amd import capi, synth  # noqa: E402


def run_case(name):
    scene, cfg, geo, inv, inv_res = shader_cases.build(synth, capi, name)
    flags = shader_cases.CASES[name][5]
    G = shader_cases.CASES[name][3]
    out = shader_ref.run_frame(scene, synth.BBOX_MIN, synth.BBOX_MAX, (G, G, G), inv, limit=cfg.tsdf_limit,
                               brick_size=geo.brick_size, res_bricks=tuple(geo.res_bricks), filter_textures=bool(flags & 1),
                               processed=bool(flags & 2), refine=bool(flags & 4), compress=name in shader_cases.COMPRESSED_DEPTH)
    assert out["bricks_out_of_range"] == 0, "a marked position left the brick grid: undefined in the shader"
    assert out["offcentre_lookups"] == 0, "a texel-centre sampler was used off centre"
    return scene, inv, out


def run_views(name, scene, inv, frame):
    """the frame's volume and images through tsdf_raymarch.fs (+ depth peels from the ORACLE for skip_space: the
    instanced brick cubes of bricks.{vs,gs,fs} need a rasteriser and are not compiled) and the fill shaders"""
    import pyoracle
    n = scene.N
    scn, cfg, geo, _, _ = shader_cases.build(synth, capi, name)
    ids, ratio = pyoracle.update_occupied(frame["counters"], cfg.min_voxels_per_brick)
    mask = np.zeros(frame["counters"].shape, np.uint8)
    mask[ids] = 1
    out = {}
    for key, eye, mode, skip, fill in shader_cases.VIEW_CASES[name]:
        view = shader_cases.make_view(capi, synth, eye, mode, skip)
        peels = pyoracle.depth_peels(bytes(view), synth.BBOX_MIN, geo.brick_size, tuple(geo.res_bricks), frame["counters"], mask) if skip else None
        col, dep, ns = shader_ref.raymarch(bytes(view), frame["tsdf"], inv, scene.uv, [scene.color[i] for i in range(n)],
                                           frame["depth_b"], frame["quality"], limit=cfg.tsdf_limit, peels=peels)
        out[key + "_color"], out[key + "_depth"], out[key + "_samples"] = col, dep, ns
        if fill:
            out[key + "_filled_color"], out[key + "_filled_depth"] = shader_ref.fill_colors(col, dep)
    return out


def main():
    assert shader_ref.available(), "oracle/_ref/libref_shaders.so is missing: make -C oracle shaders"
    for name in shader_cases.CASES:
        scene, inv, out = run_case(name)
        arrays = {k: np.stack(out[k]) for k in shader_cases.IMAGES}
        arrays["counters"] = out["counters"]
        arrays["tsdf"] = out["tsdf"]
        arrays["inputs_sha256"] = np.frombuffer(shader_cases.digest(scene, inv).encode(), dtype=np.uint8)
        path = os.path.join(HERE, "shader_passes_%s.npz" % name)
        np.savez_compressed(path, **arrays)
        print("%-40s %7.1f KiB  surface voxels %d  counted %d" % (name, os.path.getsize(path) / 1024,
              int(np.sum(np.abs(out["tsdf"]) < 0.01)), int(out["counters"].sum())))
        if name in shader_cases.VIEW_CASES:            # tsdf_raymarch.fs / the hole-filling shaders on that frame
            views = run_views(name, scene, inv, out)
            path = os.path.join(HERE, "shader_views_%s.npz" % name)
            np.savez_compressed(path, **views)
            print("%-40s %7.1f KiB  %s" % ("  views", os.path.getsize(path) / 1024,
                  ", ".join("%s: %d px hit" % (k[:-6], int((v < 1).sum())) for k, v in views.items() if k.endswith("_depth"))))


if __name__ == "__main__":
    main()
