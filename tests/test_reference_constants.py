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
    # consumers of the volume (f-2, f-4)
    ("tsdf_raymarch.fs", r"float sampleDistance = limit \* ([\d.]+)f;", 0.5, r"sd = limit \* 0\.5f"),
    ("tsdf_raymarch.fs", r"const float IsoValue = ([\d.]+);", 0.0, r"if \(density > 0\.0f\)"),
    ("tsdf_raymarch.fs", r"float prev_density = -(limit);", "limit", r"float prev = -limit;"),
    ("tsdf_raymarch.fs", r"float samples = float\(num_samples\) \* ([\d.]+);", 0.0027, r"\(float\)num \* 0\.0027f"),
    ("tsdf_raymarch.fs", r"total_color \+= colors\[i\] \* quality / \(distances\[i\] \+ ([\d.]+)\);", 0.01, r"col\[k\] \* q / \(dist \+ 0\.01f\)"),
    ("tsdf_raymarch.fs", r"total_weight \+= quality / \(distances\[i\] \+ ([\d.]+)\);", 0.01, r"tw \+= q / \(dist \+ 0\.01f\)"),
    ("tsdf_raymarch.fs", r"if\(distances\[i\] < (limit)\) \{\s+quality = texture\(kinect_qualities", "limit", r"if \(dist < limit\) tex2d_linear\(quality"),
    ("tsdf_raymarch.fs", r"\) / -view_pos\.z \* ([\d.]+)f \+ 0\.5f;", 0.5, r"/ -vp\[2\] \* 0\.5f \+ 0\.5f"),
    ("tsdf_raymarch.fs", r"if\(total_weight <= 0\.0\) total_color = vec3\(([\d.]+)\);", 1.0, r"\(cwt <= 0\.0f\) \? 1\.0f : cw\[k\] / cwt"),
    ("tsdf_raymarch.fs", r"return vec4\(total_color2, (-[\d.]+)\);", -1.0, r"diff\[3\] = -1\.0f;"),
    ("shading.glsl", r"LightPosition = vec3\(([\d.]+)f, 1\.0f, 1\.0f\)", 1.5, r"lp\[3\] = \{1\.5f, 1\.0f, 1\.0f\}"),
    ("shading.glsl", r"LightDiffuse = vec3\(1\.0f, ([\d.]+)f, 0\.7f\)", 0.9, r"ld\[3\] = \{1\.0f, 0\.9f, 0\.7f\}"),
    ("shading.glsl", r"LightAmbient = LightDiffuse \* ([\d.]+)f;", 0.2, r"\(ld\[k\] \* 0\.2f\) \* 0\.5f"),
    ("shading.glsl", r"const float ks = ([\d.]+)f;", 0.5, r"1\.0f \* 0\.5f \* sl"),
    ("shading.glsl", r"const float n = ([\d.]+)f;", 20.0, r"pow\(reflectedAngle, 20\)[^\n]*\n[^\n]*r16 = r8 \* r8;\n\s*sl = r16 \* r4;"),
    ("shading.glsl", r"solid_diffuse = vec3\(([\d.]+)f\)", 0.5, r"\+ ld\[k\] \* 0\.5f \* dc"),
    ("shading.glsl", r"vec3\((\d+),26,28\) / 255\.0f", 228, r"\{228, 26, 28\}"),
    ("shading.glsl", r"vec3\(55,126,(\d+)\) / 255\.0f", 184, r"\{55, 126, 184\}"),
    ("shading.glsl", r"vec3\(77,(\d+),74\) / 255\.0f", 175, r"\{77, 175, 74\}"),
    ("shading.glsl", r"vec3\(152,78,(\d+)\) / 255\.0f", 163, r"\{152, 78, 163\}"),
    ("shading.glsl", r"vec3\(255,(\d+),0\) / 255\.0f", 127, r"\{255, 127, 0\}"),
    ("inc_bricks.glsl", r"return bricks\[index\] > (\d+)u;", 10, r"return counters\[id\] > 10u;"),
    ("bricks.fs", r"gl_FrontFacing \? ([\d.]+) : gl_FragCoord\.z", 1.0, r"float r = 1\.0f, gneg = 0\.0f, b = 1\.0f;"),
    ("tsdf_inpaint.fs", r"const int kernel_size = (\d+);", 4, r"for \(int x = 0; x < 4; \+\+x\)\n\s*for \(int y = 0; y < 4; \+\+y\) \{\n[^\n]*4\.0f \* 0\.5f \+ 1\.0f"),
    ("tsdf_inpaint.fs", r"\* vec2\(([\d.]+) / 3\.0 , 1\.0\)\);", 2.0, r"\(float\)lx \* \(2\.0f / 3\.0f\)"),
    ("tsdf_inpaint.fs", r"if \(color\.a <= ([\d.]+)\) \{\s+color\.r = -1\.0;", 0.0, r"if \(c\[3\] <= 0\.0f\) \{\n\s*c\[0\] = -1\.0f;"),
    ("tsdf_inpaint.fs", r"out_FragColor = vec4\(0\.0, 0\.0, 0\.0, (-[\d.]+)\);", -1.0, r"oc\[3\] = -1\.0f;"),
    ("tsdf_colorfill.fs", r"if \(out_FragColor\.a > ([\d.]+)\) break;", 0.0, r"if \(c\[3\] > 0\.0f\) break;"),
    ("tsdf_colorfill.fs", r"vec2\(texture_offsets\[lod\]\) \+ ([\d.]+), vec2\(texture_offsets\[lod\] \+ texture_resolutions\[lod\]\) - 0\.5\)", 0.5,
     r"fc_clampf\(ox \+ rx \* ptx, ox \+ 0\.5f, \(ox \+ rx\) - 0\.5f\)"),
    # the offline inverter (f-3), C++
    ("../framework/calibration/calibration_inverter.cpp", r"curr_calib_search\.search\(sample_pos, (\d+)\);", 8, r"float bd\[8\];\n\s*int bi\[8\]"),
    ("../framework/calibration/calibration_inverter.cpp", r"sample_start = bbox_translation \+ sample_step \* ([\d.]+)f;", 0.5,
     r"start\[a\] = bbox_min\[a\] \+ step\[a\] \* 0\.5f;"),
    ("../framework/calibration/calibration_inverter.cpp", r"\(weighted_index \+ glm::fvec3\{([\d.]+)f\}\) / curr_calib_dims", 0.5,
     r"o\[0\] = \(wi\[0\] / tw \+ 0\.5f\) / \(float\)rx;"),
    ("../framework/calibration/calibration_inverter.cpp", r"float weight = ([\d.]+)f / glm::distance\(curr_point, sample\.pos\);", 1.0,
     r"const float w = 1\.0f / sqrtf\(bd\[k\]\);"),
    ("../framework/calibration/calibration_inverter.cpp", r"\] = glm::fvec4\{(-[\d.]+)f\};\s+continue;", -1.0,
     r"o\[0\] = o\[1\] = o\[2\] = o\[3\] = -1\.0f;"),
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
