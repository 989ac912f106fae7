#!/usr/bin/env python3
"""Fills README.md's @PLACEHOLDERS@ from a bench line (profiles/<tag>_bench.json): the README's tables are never typed by
hand.   usage: python profiles/fill_readme.py r05   (README.md.in -> README.md)"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.load(open(os.path.join(root, "profiles", tag + "_bench.json")))
extra_file = os.path.join(root, "profiles", tag + "_bench_extra.json")
if os.path.exists(extra_file):                      # the line + what it moved to bench_extra.json = the object of rounds 1-5
    for k, v in json.load(open(extra_file)).items():
        if k == "cpu_baseline":
            j["cpu_baseline"] = dict(j.get("cpu_baseline") or {}, **v)
        elif k == "roofline_box":
            j["roofline"]["box"] = v
        elif k == "roofline_tables":
            j["roofline"].update(v)
        else:
            j[k] = v
r, sc, pp, hf = j["roofline"], j["scenes"], j["post_pass"], j["host_fed"]
order = ("static", "moving", "dense", "dense_moving")


def row(fn):
    return " | ".join(fn(sc[k]) for k in order)


box = r.get("box") or {}
sub = {
    "FRAME": "%.3f" % j["ms_per_step"], "GVOX": "%.1f" % (j["value"] / 1e3), "FPS": "%.0f" % j["frames_per_s"],
    "INT": "%.4f" % r["avg_launch_ms"], "TBS": "%.2f" % (r["achieved"] / 1e3), "FRAC": "%.3f" % r["frac"],
    "BOX": "%.3f" % r["frac_of_box_stream"], "FIRST": "%.3f" % r["frac_first_placement"],
    "SCLK": "%.0f" % box["sclk_MHz"]["median"] if box.get("sclk_MHz") else "n/a", "POWER": "%.0f" % box["power_W"]["median"] if box.get("power_W") else "n/a",
    "CPU": "%.0f" % j["cpu_baseline"]["value"],
    "PRE": row(lambda m: "%.3f" % m["pre_chain_ms"]),
    "BRICK": row(lambda m: "%.3f (%.1f %%)" % (m["bricked"]["ms_per_step"], 100 * m["bricked"]["occupied_ratio"])),
    "SKIP": row(lambda m: "%.3f (%.0f %%)" % (m["background_skip"]["ms_per_step"], 100 * m["background_skip"]["frac_decided"])),
    "ELIDE": row(lambda m: "%.3f" % m["store_elision"]["ms_per_step"]),
    "REFDEF": "%.3f | %.3f" % (j["reference_defaults"]["ms_per_frame"], j["reference_defaults"]["ms_per_frame_moving"]),
    "FRACS": " / ".join("%.3f (%.3f)" % (sc[k]["full_sweep"]["roofline_frac"], sc[k]["full_sweep"]["frac_of_box_stream"]) for k in order),
    "MAPPED": "%.3f" % hf["ms_per_step_mapped_buffer"], "PAGEABLE": "%.3f" % hf["ms_per_step_pageable_upload"],
    "MARCH": "%.2f" % pp["raymarch_ms"], "MARCHSKIP": "%.3f" % pp["raymarch_skip_space_ms"], "PEELS": "%.3f" % pp["brickdraw_ms"],
    "FILL": "%.2f" % pp["holefill_ms"], "VIEWSKIP": "%.3f" % (pp["raymarch_skip_space_ms"] + pp["brickdraw_ms"]),
    "INV": "%.0f" % j["inverse_lut"]["inverse_lut_generate_ms"], "INVG": "%.2f" % j["inverse_lut"]["Gvoxels_per_s"],
}
for key, g in (("DISPLAY_REF", "reference_box"), ("DISPLAY_512", "grid_512")):
    d = j["default_display_frame"][g]
    st = d["stages_ms"]
    sub[key] = "**%.3f** / %.3f | %.3f | %.3f | %.3f | %.3f | %.3f" % (d["ms_per_frame"], d["ms_per_frame_moving"], st["pre_chain"], st["integrate"],
                                                                        st["depth_peels"], st["raymarch"], st["holefill"])
    sub[key + "_PIPE"] = "%.3f" % d["ms_per_frame_pipelined"] if "ms_per_frame_pipelined" in d else "n/a"
text = open(os.path.join(root, "README.md.in")).read()
for k, v in sub.items():
    text = text.replace("@%s@" % k, v)
assert "@" not in text.replace("@pytest", ""), [w for w in text.split() if w.startswith("@")]
open(os.path.join(root, "README.md"), "w").write(text)
print("README.md written from profiles/%s_bench.json" % tag)
