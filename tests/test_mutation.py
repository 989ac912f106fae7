"""The KATs kill every listed mutant of the oracle (tests/mutation_check.py): the oracle's pass arithmetic
is pinned only by reading the shaders, so its analytic tests must be able to see a transcription slip."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_oracle_mutant_is_killed():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mutation_check.py"), "-j", "4"], capture_output=True,
                       text=True, timeout=900)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0, "\n".join(l for l in lines if "killed by" not in l) + r.stderr[-2000:]
    assert lines[0].startswith("control") and lines[0].endswith("passes")
    assert len([l for l in lines if "killed by" in l]) >= 30
