"""The numeric constants of the shaders on the hot path, read from the reference tree at test time (this test is
skipped where /root/reference does not exist) and compared with what the oracle restates -- a mechanical check of
the one kind of transcription slip that a known-answer test written from the same reading would share.  For every
constant: (shader file, a regular expression around the literal in the reference's text) -> the value, and the
literal as the oracle spells it at the line that restates that expression (oracle/rgbdr_oracle.c); the mutation
check (tests/mutation_check.py) shows that the known-answer tests are sensitive to these constants in the oracle."""
import os
import re

import pytest

REF = "/root/reference/glsl"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present on this machine")

# (shader, regex in the reference with one group = the literal, value, regex the oracle must match)
CONSTANTS = [
    ("pre_depth.fs", r"const int kernel_size = (\d+);", 6, r"bilateral_filter, pre_depth\.fs[^\n]*\n(?:[^\n]*\n){4}\s*for \(int y = -6; y < 7; \+\+y\) \{\n\s*for \(int x = -6; x < 7; \+\+x\)"),
    ("pre_depth.fs", r"const float max_depth = ([\d.]+)f; // Kinect V2\s+float d_dmax", 4.5, r"d_dmax = depth0 / 4\.5f"),
    ("pre_depth.fs", r"dist_range_max = ([\d.]+)f \* d_dmax", 0.35, r"dist_range_max = 0\.35f \* d_dmax"),
    ("pre_quality.fs", r"const int kernel_size = (\d+);", 6, r"border_samples = 0\.0f, num_samples = 0\.0f;\n\s*for \(int y = -6; y < 7; \+\+y\) \{\n\s*for \(int x = -6; x < 7; \+\+x\)"),
    ("pre_quality.fs", r"const float max_depth = ([\d.]+)f; // Kinect V2", 1.0, r"0\.35f \* \(depth / 1\.0f\)"),
    ("pre_quality.fs", r"dist_range_max = ([\d.]+)f \* d_dmax", 0.35, r"0\.35f \* \(depth / 1\.0f\)"),
    ("pre_quality.fs", r"quality_strong /= depth \* ([\d.]+)f;", 6.5, r"q /= depth \* 6\.5f"),
    ("pre_boundary.fs", r"const int kernel_size = (\d+);", 2, r"num_samples < 16\.0f \* 0\.5f"),
    ("pre_boundary.fs", r"const float min_range = ([\d.]+)f;", 0.65, r"!\(dy > 0\.65f\)"),
    ("pre_boundary.fs", r"const float max_color_dist = ([\d.]+)f;", 0.5, r"color_dist > 0\.5f \|\| !refine"),
    ("pre_boundary.fs", r"num_samples < total_samples \* ([\d.]+)f\) return 1\.0f", 0.5, r"num_samples < 16\.0f \* 0\.5f\) \? 1\.0f"),
    ("pre_boundary.fs", r"depth\.y = ([\d.]+)f;\s+out_Silhouette = 0\.0f;\s+\}\s+else", 0.1, r"= 0\.1f"),
    ("pre_morph.fs", r"#else\s+const float min_depth = ([\d.]+)f;", 0.5, r"d > 0\.5f && d < 4\.5f"),
    ("pre_morph.fs", r"#else\s+const float min_depth = [\d.]+f;\s+const float max_depth = ([\d.]+)f;", 4.5, r"d > 0\.5f && d < 4\.5f"),
    ("pre_morph.fs", r"const float max_dist = ([\d.]+);", 0.2, r"fabsf\(average_depth - ds\) < 0\.2f"),
    ("inc_color.glsl", r"white_reference = vec3\(([\d.]+),", 95.047, r"X / 95\.047f"),
    ("inc_color.glsl", r"white_reference = vec3\([\d.]+, ([\d.]+),", 100.0, r"Y / 100\.0+f"),
    ("inc_color.glsl", r"white_reference = vec3\([\d.]+, [\d.]+, ([\d.]+)\)", 108.883, r"Z / 108\.883f"),
    ("inc_color.glsl", r"const float epsilon = ([\d.]+)f;", 0.008856, r"n > 0\.008856f"),
    ("inc_color.glsl", r"const float kappa\s+= ([\d.]+)f;", 903.3, r"903\.3f \* n \+ 16\.0f\) / 116\.0f"),
    ("inc_color.glsl", r"n > ([\d.]+) \? pow\(\(n \+ 0\.055\) / 1\.055, 2\.4\) : n / 12\.92\) \* 100\.0", 0.04045,
     r"n > 0\.04045f \? powf\(\(n \+ 0\.055f\) / 1\.055f, 2\.4f\) : n / 12\.92f\) \* 100\.0f"),
    ("inc_color.glsl", r"xyz_col\[0\] = r \* ([\d.]+) \+ g \* 0\.3576 \+ b \* 0\.1805", 0.4124, r"0\.4124f"),
    ("inc_color.glsl", r"xyz_col\[2\] = r \* 0\.0193 \+ g \* 0\.1192 \+ b \* ([\d.]+)", 0.9505, r"0\.9505f"),
    ("inc_color.glsl", r"lab_col\[0\] = max\(0\.0, (\d+)\*y -16\)", 116, r"fmaxf\(0\.0f, 116\.0f \* y - 16\.0f\)"),
    ("inc_bricks.glsl", r"\(d_abs\.x > brick_size \* ([\d.]+)\) \? 1u : 0u", 0.1, r"dabs\[0\] > p->brick_size \* 0\.1f\) \? 1u : 0u"),
    ("tsdf_integration.vs", r"if \(sdist <= -(limit) \)", "limit", r"sdist <= -limit"),
    ("tsdf_integration.vs", r"else if \(sdist >= (limit) \)", "limit", r"sdist >= limit"),
    ("tsdf_integration.vs", r"if \(weighted_tsd >= (limit)\)", "limit", r"tsd >= limit|weighted_tsd >= limit"),
]


@pytest.fixture(scope="module")
def oracle_text():
    return open(os.path.join(ROOT, "oracle", "rgbdr_oracle.c")).read()


@pytest.mark.parametrize("shader,pattern,value,oracle_pattern", CONSTANTS, ids=lambda v: str(v)[:40])
def test_constant_of_the_reference_shader_is_the_oracles(oracle_text, shader, pattern, value, oracle_pattern):
    text = open(os.path.join(REF, shader)).read()
    m = re.search(pattern, text)
    assert m, "the reference's %s no longer matches %r" % (shader, pattern)
    got = m.group(1)
    if isinstance(value, str):
        assert got == value
    else:
        assert float(got) == pytest.approx(float(value), rel=0, abs=0), (shader, got, value)
    assert re.search(oracle_pattern, oracle_text), "oracle/rgbdr_oracle.c does not restate %r" % oracle_pattern
