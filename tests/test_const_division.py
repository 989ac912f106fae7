"""The constant divisions of the Lab conversion (kernels_pre.hip: divc) are three instructions instead of a
correctly rounded division; tests/const_division_check.c compares the two for every binary32 dividend.  Here
a strided sample of it (every 1021st bit pattern, all exponents) -- the full run takes minutes."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constant_division_matches_ieee_division_in_the_range_the_kernel_uses(tmp_path):
    exe = str(tmp_path / "cdc")
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "tests", "const_division_check.c"),
                    "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "1021"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 6 and all(l.endswith(": 0") for l in lines), r.stdout
