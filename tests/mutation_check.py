#!/usr/bin/env python3
"""Mutation check of the oracle's known-answer tests (TEST INFRASTRUCTURE).

Until round 3 the pass arithmetic of the oracle was pinned only by reading the shaders; now the reference's GLSL runs
on Mesa in the build container (tests/test_gl_ref.py) -- within llvmpipe's ulps.  The KATs below remain
the exact, driver-independent check: a transcription slip in oracle/rgbdr_oracle.c would be mirrored by the kernels and pass
every HIP-vs-oracle test.  The analytic KATs (tests/test_oracle_kat.py, tests/test_oracle_kat_passes.py,
tests/test_bricks_cpu.py) are the independent check -- this script shows they have teeth: each entry of
MUTANTS flips one detail of the reference (a quirk, a constant, a comparison) in a scratch copy of the
oracle source, builds it, and runs the KATs against it through RGBDR_ORACLE_LIB; the mutant must make at
least one KAT fail ("killed").  The unmodified copy must pass (control).

  python tests/mutation_check.py            all mutants, 4 at a time
  python tests/mutation_check.py -k brick   only mutants whose name contains "brick"
tests/test_mutation.py runs it inside the CPU suite.
"""
import argparse
import concurrent.futures
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oracle", "rgbdr_oracle.c")
KATS = ["tests/test_oracle_kat.py", "tests/test_oracle_kat_passes.py", "tests/test_bricks_cpu.py", "tests/test_oracle_kat_view.py"]

# (name, text in oracle/rgbdr_oracle.c -- must occur exactly once --, replacement)
MUTANTS = [
    # pre_morph.fs
    ("morph: valid range 0.5 < d < 4.5", "return d > 0.5f && d < 4.5f;", "return d >= 0.5f && d <= 4.5f;"),
    # pre_depth.fs
    ("uncompress: threshold is d < scaled_near", "if (d < scaled_near) return 0.0f;", "if (d <= scaled_near) return 0.0f;"),
    ("uncompress: sqrt mapping d*d", "return (d * d + 0.15f * scaled_near) * scale + p->near_;",
     "return (d + 0.15f * scaled_near) * scale + p->near_;"),
    ("uncompress: offset 0.15 * scaled_near", "return (d * d + 0.15f * scaled_near) * scale + p->near_;",
     "return (d * d + 0.5f * scaled_near) * scale + p->near_;"),
    ("bilateral: spatial weights are not clamped at 0", "float gauss_space = 1.0f - len * inv_k;",
     "float gauss_space = fmaxf(0.0f, 1.0f - len * inv_k);"),
    ("bilateral: spatial kernel radius 6", "const float inv_k = 1.0f / 6.0f; /* dist_space_max_inv, pre_depth.fs:37 */",
     "const float inv_k = 1.0f / 7.0f;"),
    ("bilateral: range threshold 0.35 * depth / 4.5", "float d_dmax = depth0 / 4.5f;", "float d_dmax = depth0 / 4.0f;"),
    ("bilateral: taps outside [cv_min_ds, cv_max_ds] are skipped",
     "if ((depth_s < p->cv_min_ds) || (depth_s > p->cv_max_ds) || (depth_range > dist_range_max)) continue;",
     "if (depth_range > dist_range_max) continue;"),
    ("bilateral: quality divides by all 169 taps", "out_depth_rg[o * 2 + 1] = w_range / num_samples;",
     "out_depth_rg[o * 2 + 1] = w_range / fmaxf(accepted, 1.0f);"),
    ("bilateral: range weight 1 - dr / dr_max", "float gauss_range = 1.0f - fminf(depth_range, dist_range_max) * dist_range_max_inv;\n          float w_s",
     "float gauss_range = 1.0f;\n          float w_s"),
    # inc_bricks.glsl
    ("mark_brick: increment tests d_abs.x", "const uint32_t inc = (dabs[0] > p->brick_size * 0.1f) ? 1u : 0u;",
     "const uint32_t inc = (min_v > p->brick_size * 0.1f) ? 1u : 0u;"),
    ("mark_brick: increment threshold 0.1 * brick_size", "const uint32_t inc = (dabs[0] > p->brick_size * 0.1f) ? 1u : 0u;",
     "const uint32_t inc = (dabs[0] > p->brick_size * 0.2f) ? 1u : 0u;"),
    ("mark_brick: every maximal axis is selected (ties)", "float mc = (dabs[a] < min_v) ? 0.0f : 1.0f;",
     "float mc = (dabs[a] < min_v || (a > 0 && dabs[0] >= min_v)) ? 0.0f : 1.0f;"),
    ("mark_brick: neighbour is clamped to the grid", "nb[a] = clampi(idx[a] + off, 0, p->res_bricks[a] - 1);",
     "nb[a] = idx[a] + off; if (nb[a] < 0 || nb[a] >= p->res_bricks[a]) return 0;"),
    ("mark_brick: the home brick is counted", "  bricks[(size_t)idx[2] * ry * rx + (size_t)idx[1] * rx + idx[0]] += 1u;",
     "  bricks[(size_t)idx[2] * ry * rx + (size_t)idx[1] * rx + idx[0]] += inc;"),
    # pre_normal.fs
    ("normal: cross(b - t, l - r) orientation", "cross3(a, b, c);", "cross3(b, a, c);"),
    ("normal: b - t, not t - b", "float a[3] = {wb[0] - wt[0], wb[1] - wt[1], wb[2] - wt[2]};",
     "float a[3] = {wt[0] - wb[0], wt[1] - wb[1], wt[2] - wb[2]};"),
    ("normal: invalid neighbours take the centre depth", "dr = unit_outside(dr) ? depth : dr;", "dr = dr;"),
    ("normal/quality: d >= 1 is outside", "static inline int unit_outside(float d) { return (d <= 0.0f) || (d >= 1.0f); }",
     "static inline int unit_outside(float d) { return (d <= 0.0f) || (d > 1.0f); }"),
    # pre_quality.fs
    ("quality: range threshold 0.35 * depth", "float dist_range_max = 0.35f * (depth / 1.0f);",
     "float dist_range_max = 0.35f * (depth / 4.5f);"),
    ("quality: invalid taps count as border", "if (unit_outside(ds) || (depth_range > dist_range_max)) {",
     "if (depth_range > dist_range_max) {"),
    ("quality: lateral term 1 - border / 169", "float lateral_quality = 1.0f - border_samples / num_samples;",
     "float lateral_quality = 1.0f;"),
    ("quality: exponent 6 on the range term", "q *= orc_pow6(w_range / num_samples);", "q *= orc_pow2(w_range / num_samples);"),
    ("quality: divided by depth * 6.5", "q /= depth * 6.5f;", "q /= depth * 6.0f;"),
    ("quality: angle squared", "q *= orc_pow2(angle);", "q *= fabsf(angle);"),
    # inc_color.glsl
    ("lab: the reference's extra / 255", "float r = pivot_rgb(rgb[0] / 255.0f);", "float r = pivot_rgb(rgb[0]);"),
    # pre_boundary.fs
    ("boundary: quality threshold 0.65", "0.65f", None),       # every 0.65f of the boundary pass -> 0.60f
    ("boundary: dx <= 0 is outside", "if (dx <= 0.0f) { /* :90-100 */", "if (dx < 0.0f) { /* :90-100 */"),
    ("boundary: fewer than 8 confident neighbours -> colour distance 1",
     "float color_dist = (num_samples < 16.0f * 0.5f) ? 1.0f : total_dist / num_samples;",
     "float color_dist = (num_samples < 1.0f) ? 1.0f : total_dist / num_samples;"),
    ("boundary: colour distance threshold 0.5", "if (color_dist > 0.5f || !refine) {", "if (color_dist > 5.5f || !refine) {"),
    ("boundary: refined edges keep silhouette 0", "      } else if (!(dy > 0.65f)) { /* :102-113 */\n        sil = 0.0f;",
     "      } else if (!(dy > 0.65f)) { /* :102-113 */\n        sil = 1.0f;"),
    # sampling
    ("sampling: LINEAR is offset by half a texel", "float t = s * (float)n - 0.5f;", "float t = s * (float)n;"),
    ("sampling: NEAREST is floor(s * n)", "float f = floorf(s * (float)n);", "float f = floorf(s * (float)n + 0.5f);"),
    # tsdf_integration.vs
    ("integrate: sdist >= limit leaves the voxel untouched", "} else if (sdist >= limit) {", "} else if (sdist > limit) {"),
    ("integrate: silhouette only overwrites an untouched voxel", "if (weighted_tsd >= limit) {", "if (1) {"),
    # divideBox / containedVoxels
    ("bricks: float upper bound of containedVoxels (orc_brick_voxel_mask)",
     "for (unsigned x = (unsigned)(pos[0] / stepv[0]); (float)x < (pos[0] + bs[0]) / stepv[0]; ++x)\n              for (unsigned z = (unsigned)(pos[2] / stepv[2]); (float)z < (pos[2] + bs[2]) / stepv[2]; ++z) {\n                const size_t id",
     "for (unsigned x = (unsigned)(pos[0] / stepv[0]); x < (unsigned)((pos[0] + bs[0]) / stepv[0]); ++x)\n              for (unsigned z = (unsigned)(pos[2] / stepv[2]); (float)z < (pos[2] + bs[2]) / stepv[2]; ++z) {\n                const size_t id"),
    # tsdf_raymarch.fs (f-2)
    ("raymarch: step is limit / 2", "const float limit = p->limit, sd = limit * 0.5f;", "const float limit = p->limit, sd = limit * 0.25f;"),
    ("raymarch: secant prev / (density - prev)", "const float f = prev / (density - prev);", "const float f = prev / (density + prev);"),
    ("raymarch: refinement starts one step back", "for (int a = 0; a < 3; ++a) sp[a] = (sp[a] - step[a]) - step[a] * f;",
     "for (int a = 0; a < 3; ++a) sp[a] = sp[a] - step[a] * f;"),
    ("raymarch: density > 0 is inside, 0 is not", "if (density > 0.0f) {", "if (density >= 0.0f) {"),
    ("raymarch: the first sample is assumed outside (-limit)", "      float prev = -limit;\n      unsigned num = 0;", "      float prev = 0.0f;\n      unsigned num = 0;"),
    ("raymarch: camera inside the cube starts at the camera", "t_near = t_near < 0.0f ? 0.0f : t_near;", "t_near = t_near;"),
    ("raymarch: gradient points to smaller density", "d[a] = tsdf_sample(tsdf, res, pp) - tsdf_sample(tsdf, res, pm);",
     "d[a] = tsdf_sample(tsdf, res, pm) - tsdf_sample(tsdf, res, pp);"),
    ("raymarch: sample count image scale 0.0027", "out_samples[o] = (float)num * 0.0027f;", "out_samples[o] = (float)num * 0.0028f;"),
    ("raymarch: blend weight quality / (dist + 0.01)", "        tw += q / (dist + 0.01f);", "        tw += q / (dist + 0.1f);"),
    ("raymarch: quality only within limit of the surface", "if (dist < limit) tex2d_linear(quality[i], 1, 1, p->W, p->H, pcal[0], pcal[1], &q);",
     "if (dist < 3.0f * limit) tex2d_linear(quality[i], 1, 1, p->W, p->H, pcal[0], pcal[1], &q);"),
    ("raymarch: fallback blend weighs by 1 / distance", "          tc2[k] += col[k] / dist;", "          tc2[k] += col[k];"),
    ("raymarch: fallback blend has alpha -1", "          diff[3] = -1.0f;", "          diff[3] = 1.0f;"),
    ("raymarch: camera mode is white without weights", "rgba[k] = (cwt <= 0.0f) ? 1.0f : cw[k] / cwt;", "rgba[k] = (cwt <= 0.0f) ? 0.0f : cw[k] / cwt;"),
    ("raymarch: gl_FragDepth divides by -z", "/ -vp[2] * 0.5f + 0.5f;", "/ vp[2] * 0.5f + 0.5f;"),
    ("raymarch: fragments at the far plane fail the depth test", "if (!(dclamped < 1.0f)) continue;", "if (0) continue;"),
    ("raymarch: phong specular exponent 20", "sl = r16 * r4;", "sl = r16;"),
    ("raymarch: phong ambient 0.2", "rgba[k] = (ld[k] * 0.2f) * 0.5f", "rgba[k] = (ld[k] * 0.3f) * 0.5f"),
    # getStartPos (f-4)
    ("start pos: sample budget is distance / step", "fmaxs = ceilf(distance3(pf, pb) / sd);", "fmaxs = ceilf(distance3(pf, pb) / limit);"),
    ("start pos: a culled front face starts at the near plane", "dr = (dr >= dbk) ? 0.0f : dr; /* gl_DepthRange.near */", "dr = (dr > dbk) ? 0.0f : dr;"),
    ("start pos: the back face depth is -g", "k == 0 ? dr : -dg, 1.0f};", "k == 0 ? dr : dg, 1.0f};"),
    ("start pos: no face at all means no samples", "        if (dr >= 1.0f) {\n          pb[0] = pf[0];", "        if (0) {\n          pb[0] = pf[0];"),
    # bricks.{gs,fs}
    ("peels: neighbour culls a face when its counter > 10", "return counters[id] > 10u;", "return counters[id] >= 10u;"),
    ("peels: across the grid's boundary the neighbour is the brick the uint index wraps to",
     "  if (id < 0 || id >= nb) return 0;\n  return counters[id] > 10u;",
     "  if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= g->res_bricks[0] || c[1] >= g->res_bricks[1] || c[2] >= g->res_bricks[2]) return 0;\n  return counters[id] > 10u;"),
    ("peels: blue keeps the nearest BACK face only", "            r = fminf(r, z);\n            gneg = fminf(gneg, -z);\n          }\n          if (prev_list",
     "            r = fminf(r, z);\n            gneg = fminf(gneg, -z);\n            b = fminf(b, z);\n          }\n          if (prev_list"),
    ("peels: green is MIN of -z (the farthest face)", "            gneg = fminf(gneg, -z);\n            b = fminf(b, z);\n          }\n        }\n      }\n    }\n    first = 0;",
     "            gneg = fmaxf(gneg, -z);\n            b = fminf(b, z);\n          }\n        }\n      }\n    }\n    first = 0;"),
    ("peels: cleared to (1, 0, 1, 0)", "  out[0] = 1.0f;\n  out[1] = 0.0f;\n  out[2] = 1.0f;\n  out[3] = 0.0f;\n  /* world-space ray",
     "  out[0] = 1.0f;\n  out[1] = 0.0f;\n  out[2] = 0.0f;\n  out[3] = 0.0f;\n  /* world-space ray"),
    # calibration_inverter.cpp (f-3)
    ("inverter: weights are 1 / distance", "const float w = 1.0f / sqrtf(bd[k]);", "const float w = 1.0f / bd[k];"),
    ("inverter: output is (index + 0.5) / dims", "o[1] = (wi[1] / tw + 0.5f) / (float)ry;", "o[1] = (wi[1] / tw) / (float)ry;"),
    ("inverter: samples start half a voxel inside the box", "start[a] = bbox_min[a] + step[a] * 0.5f;", "start[a] = bbox_min[a];"),
    ("inverter: points outside the frustum are -1", "o[0] = o[1] = o[2] = o[3] = -1.0f;", "o[0] = o[1] = o[2] = o[3] = 0.0f;"),
    ("inverter: eight neighbours", "int k = n < 8 ? n : 7;", "int k = n < 8 ? n : 7; if (n >= 4 && !(d2 < bd[3])) continue; if (k > 3) k = 3;"),
    # tsdf_inpaint.fs / tsdf_colorfill.fs / framebuffer_transfer.fs (f-2)
    ("inpaint: only samples at or behind the mean depth", "if (samples[k][0] >= 0.0f && samples[k][3] >= depth_av) {", "if (samples[k][0] >= 0.0f && samples[k][3] <= depth_av) {"),
    ("inpaint: reads the atlas squeezed to 2/3 in x", "const int pix = (int)((float)lx * (2.0f / 3.0f)), piy = (int)((float)ly * 1.0f);",
     "const int pix = (int)((float)lx * 1.0f), piy = (int)((float)ly * 1.0f);"),
    ("inpaint: alpha <= 0 is a hole", "          if (c[3] <= 0.0f) {\n            c[0] = -1.0f;", "          if (c[3] < 0.0f) {\n            c[0] = -1.0f;"),
    ("inpaint: hole over a surface depth is marked alpha -1", "          oc[2] = 0.0f;\n          oc[3] = -1.0f;", "          oc[2] = 0.0f;\n          oc[3] = 0.0f;"),
    ("inpaint: 4 x 4 taps from -1 to +2", "(int)((float)x - 4.0f * 0.5f + 1.0f), ty", "(int)((float)x - 4.0f * 0.5f), ty"),
    ("colorfill: first LOD with alpha > 0", "        if (c[3] > 0.0f) break;", "        if (c[3] >= 0.0f) break;"),
    ("colorfill: blends LOD + 1 and LOD + 2", "const int l = level + 1 + k; /* level+1 -> p1, level+2 -> p2 */", "const int l = level + k;"),
    ("colorfill: depth comes from LOD 0", "out_dep[(size_t)py * L->W + px] = d0;", "out_dep[(size_t)py * L->W + px] = d;"),
    ("lod atlas: 1.5 W wide", "L->FW = (int)((float)W * 1.5f);", "L->FW = (int)((float)W * 2.0f);"),
]


def apply(src, old, new, name):
    if new is None:                      # "boundary: quality threshold": every 0.65f of the boundary pass
        assert src.count(old) >= 1, name
        return src.replace(old, "0.60f")
    old, new = old.replace("\\n", "\n"), new.replace("\\n", "\n")
    n = src.count(old)
    assert n == 1, "mutant %r: its anchor text occurs %d times in rgbdr_oracle.c (must be 1)" % (name, n)
    return src.replace(old, new)


def build_and_run(name, text, workdir):
    base = os.path.join(workdir, "".join(ch if ch.isalnum() else "_" for ch in name)[:60])
    with open(base + ".c", "w") as f:
        f.write(text)
    cc = ["gcc", "-O1", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fvisibility=hidden", "-std=c11", "-w",
          "-shared", "-o", base + ".so", base + ".c", "-lm"]
    r = subprocess.run(cc, capture_output=True, text=True)
    if r.returncode != 0:
        return name, None, r.stderr[-2000:]
    env = dict(os.environ, RGBDR_ORACLE_LIB=base + ".so", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", "not asan"] + KATS,
                       cwd=ROOT, env=env, capture_output=True, text=True)
    failed = [l for l in r.stdout.splitlines() if l.startswith("FAILED")]
    return name, r.returncode, (failed[0] if failed else r.stdout[-300:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-k", default="", help="only mutants whose name contains this")
    ap.add_argument("-j", type=int, default=4)
    args = ap.parse_args()
    src = open(SRC).read()
    # the "accepted" counter the quality mutant refers to
    src_m = src
    jobs = [("control (unmodified oracle)", src)]
    for name, old, new in MUTANTS:
        if args.k and args.k not in name:
            continue
        text = src_m
        if "fmaxf(accepted" in (new or ""):
            text = text.replace("float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f, num_samples = 0.0f;",
                                "float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f, num_samples = 0.0f, accepted = 0.0f;")
            text = text.replace("          w_range += gauss_range;\n        }\n      }\n      float filtered",
                                "          w_range += gauss_range;\n          accepted += 1.0f;\n        }\n      }\n      float filtered")
            assert "accepted += 1.0f" in text
        jobs.append((name, apply(text, old, new, name)))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(args.j) as pool:
        for name, rc, info in pool.map(lambda j: build_and_run(j[0], j[1], tmp), jobs):
            if name.startswith("control"):
                ok = rc == 0
                print("%-72s %s" % (name, "passes" if ok else "FAILS: " + info))
            else:
                ok = rc is not None and rc != 0
                print("%-72s %s" % (name, ("killed by " + info.replace("FAILED ", "")) if ok else
                                    ("DID NOT BUILD: " + info if rc is None else "SURVIVED")))
            bad += 0 if ok else 1
    print("%d mutants, %d not killed" % (len(jobs) - 1, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
