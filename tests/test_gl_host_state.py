"""The GL state oracle/gl_ref.py sets up for the run of the reference's shaders on Mesa is a restatement of the reference's
HOST code.  This test reads that host code where it lies (build container only) and checks, statement by statement, that
the formats, filters, texture / image / buffer bindings, uniforms, clear values and vertex data the harness uses are the
ones the source states -- the counterpart of tests/test_reference_constants.py for the part of the Mesa run that is not
executed from the reference's text.  Each entry: (file, regular expression that must match the source, what gl_ref.py does
with it).  Where gl_ref.py keeps the value in a variable, the captured group is compared with that variable."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "oracle")]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/framework"), reason="reference checkout absent (build container only)")

NKA = "framework/NetKinectArray.cpp"
RI = "framework/reconstruction/recon_integration.cpp"
CV = "framework/calibration/CalibVolumes.cpp"
KC = "source/kinect_client.cpp"

# (file, pattern, note)
STATEMENTS = [
    # ---- NetKinectArray::init: formats and filters of the pass targets ----
    (NKA, r"m_textures_color->image3D\(0, GL_RGB32F,", "Lab target RGB32F"),
    (NKA, r"m_textures_quality->image3D\(0, GL_LUMINANCE32F_ARB,", "quality LUMINANCE32F"),
    (NKA, r"m_textures_normal->image3D\(0, GL_RGB32F,", "normals RGB32F"),
    (NKA, r"m_textures_silhouette->image3D\(0, GL_R32F,", "silhouette R32F"),
    (NKA, r"m_textures_depth->image3D\(0, GL_RG32F,", "filtered depth RG32F"),
    (NKA, r"m_textures_depth_b->image3D\(0, GL_RG32F,", "boundary depth RG32F"),
    (NKA, r"m_textures_depth2\.front->image3D\(0, GL_LUMINANCE32F_ARB,", "morph ping LUMINANCE32F"),
    (NKA, r"m_textures_depth2\.back->image3D\(0, GL_LUMINANCE32F_ARB,", "morph pong LUMINANCE32F"),
    (NKA, r"new TextureArray\(m_resolution_depth\.x, m_resolution_depth\.y, m_numLayers, GL_LUMINANCE32F_ARB, GL_RED, GL_FLOAT\)", "raw depth f32"),
    (NKA, r"new TextureArray\(m_resolution_depth\.x, m_resolution_depth\.y, m_numLayers, GL_LUMINANCE, GL_RED, GL_UNSIGNED_BYTE\)", "raw depth u8"),
    (NKA, r"new TextureArray\(m_resolution_color\.x, m_resolution_color\.y, m_numLayers, GL_RGB, GL_RGB, GL_UNSIGNED_BYTE\)", "colour RGB8"),
    (NKA, r"m_depthArray_raw->setMAGMINFilter\(GL_NEAREST\)", "raw depth NEAREST"),
    (NKA, r"m_textures_depth_b->setParameter\(GL_TEXTURE_MIN_FILTER, GL_NEAREST\)", "depth_b NEAREST"),
    (NKA, r"m_textures_depth->setParameter\(GL_TEXTURE_MAG_FILTER, GL_NEAREST\)", "depth NEAREST"),
    (NKA, r"m_textures_depth2\.front->setParameter\(GL_TEXTURE_MIN_FILTER, GL_NEAREST\)", "depth2 NEAREST"),
    (NKA, r"m_textures_depth2\.back->setParameter\(GL_TEXTURE_MAG_FILTER, GL_NEAREST\)", "depth2 NEAREST"),
    (NKA, r"glm::fvec2 tex_size_inv\{1\.0f/m_resolution_depth\.x, 1\.0f/m_resolution_depth\.y\}", "texSizeInv = 1 / size in float"),
    (NKA, r'm_programs\.at\("quality"\)->setUniform\("camera_positions", m_calib_vols->getCameraPositions\(\)\)', "camera positions"),
    # ---- processDepth / processTextures: what is attached and set per pass ----
    (NKA, r'setUniform\("mode", 0u\);\s*for\(unsigned i = 0; i < m_calib_files->num\(\); \+\+i\)\{\s*m_fbo->attachTextureLayer\(GL_COLOR_ATTACHMENT0, m_textures_depth2\.back, 0, i\)', "morph mode 0 into depth2.back"),
    (NKA, r'setUniform\("mode", 1u\);\s*m_textures_depth2\.swapBuffers\(\);', "swap, then mode 1"),
    (NKA, r"if\(m_use_processed_depth\) \{\s*m_textures_depth2\.front->bindActive\(getTextureUnit\(\"raw_depth\"\)\)", "processed depth rebinds raw_depth"),
    (NKA, r"attachTextureLayer\(GL_COLOR_ATTACHMENT0, m_textures_depth, 0, i\);\s*m_fbo->attachTextureLayer\(GL_COLOR_ATTACHMENT1, m_textures_color, 0, i\)", "filter pass targets"),
    (NKA, r'setUniform\("scaled_near", scale/255\.0f\)', "scaled_near"),
    (NKA, r"attachTextureLayer\(GL_COLOR_ATTACHMENT0, m_textures_depth_b, 0, i\);\s*m_fbo->attachTextureLayer\(GL_COLOR_ATTACHMENT1, m_textures_silhouette, 0, i\)", "boundary pass targets"),
    (NKA, r'm_textures_depth_b->bindActive\(getTextureUnit\("depth"\)\);\s*// normals', "normals read depth_b"),
    (NKA, r"attachTextureLayer\(GL_COLOR_ATTACHMENT0, m_textures_normal, 0, i\)", "normal target"),
    (NKA, r"attachTextureLayer\(GL_COLOR_ATTACHMENT0, m_textures_quality, 0, i\)", "quality target"),
    # ---- CalibVolumes ----
    (CV, r"volume_xyz_inv->image3D\(0, GL_RGBA32F,", "inverse LUT RGBA32F"),
    (CV, r"volume_xyz->image3D\(0, GL_RGB32F,", "xyz LUT RGB32F"),
    (CV, r"volume_uv->image3D\(0, GL_RG32F,", "uv LUT RG32F"),
    (CV, r"units\[i\] = m_start_texture_unit \+ i \* 2;", "xyz units start + 2 i"),
    (CV, r"units\[i\] = m_start_texture_unit \+ i \* 2 \+ 1;", "uv units start + 2 i + 1"),
    (CV, r"units\[i\] = m_start_texture_unit_inv \+ i;", "inverse units start + i"),
    (CV, r"std::vector<glm::fvec4> bbox_ext\{glm::fvec4\{min, 1\.0f\}, glm::fvec4\{max, 1\.0f\}\}", "BBox UBO = two vec4"),
    (CV, r"m_buffer_bbox->bindBase\(GL_UNIFORM_BUFFER, 2\)", "BBox UBO binding 2"),
    # ---- ReconIntegration ----
    (RI, r'm_program_integration->setUniform\("kinect_silhouettes",5\)', "silhouettes on unit 5"),
    (RI, r'm_program->setUniform\("depth_peels", 17\)', "peels on unit 17"),
    (RI, r'm_program->setUniform\("volume_tsdf", 29\)', "volume sampler on unit 29"),
    (RI, r'm_program_inpaint->setUniform\("texture_color", 15\);\s*m_program_inpaint->setUniform\("texture_depth", 16\)', "fill textures on 15 / 16"),
    (RI, r"m_view_depth->setClearColor\(glm::fvec4\{1\.0f, 0\.0f, 1\.0f, 0\.0f\}\)", "peel target cleared to (1, 0, 1, 0)"),
    (RI, r"m_volume_tsdf->clearImage\(0, GL_RED, GL_FLOAT, &negative\)", "volume cleared to -limit"),
    (RI, r"m_volume_tsdf->bindImageTexture\(start_image_unit, 0, GL_TRUE, 0, GL_WRITE_ONLY, GL_R32F\)", "volume image binding"),
    (RI, r"m_volume_tsdf->image3D\(0, GL_R32F,", "volume R32F"),
    (RI, r"m_tex_num_samples->bindImageTexture\(start_image_unit \+ 1, 0, GL_FALSE, 0, GL_WRITE_ONLY, GL_R32F\)", "sample-count image"),
    (RI, r"glBlendEquation\(GL_MIN\);\s*UnitCube::drawInstanced\(m_bricks_occupied\.size\(\)\)", "peels: MIN blending, instanced cubes"),
    (RI, r"m_buffer_bricks->bindRange\(GL_SHADER_STORAGE_BUFFER, 3, 0,", "brick SSBO binding 3"),
    (RI, r"m_buffer_occupied->bindRange\(GL_SHADER_STORAGE_BUFFER, 4, 0,", "occupied SSBO binding 4"),
    (RI, r"std::memcpy\(&bricks\[0\], &m_brick_size, sizeof\(float\)\);\s*std::memcpy\(&bricks\[4\], &m_res_bricks, sizeof\(unsigned\) \* 3\)", "SSBO header layout"),
    (RI, r"if\(m_active_bricks\[i\] >= m_min_voxels_per_brick\)", "occupied = counter >= min_voxels"),
    (RI, r"glDepthFunc\(GL_ALWAYS\);", "fill pyramid: depth func ALWAYS"),
    (RI, r"glDepthFunc\(GL_LESS\);\s*// tranfer to default framebuffer", "colorfill: depth func LESS"),
    (RI, r'setUniform\("resolution_inv", 1\.0f / glm::fvec2\{m_view_inpaint->resolution_full\(\)\}\)', "resolution_inv from the full atlas"),
    # ---- ViewLod, View, UnitCube, ScreenQuad, VolumeSampler ----
    ("framework/rendering/view_lod.cpp", r"m_resolution_full = glm::uvec2\{width \* 1\.5f, height\}", "atlas 1.5 W x H"),
    ("framework/rendering/view_lod.cpp", r"m_tex_color->image2D\(0, GL_RGBA32F,", "atlas colour RGBA32F"),
    ("framework/rendering/view_lod.cpp", r"m_tex_depth->image2D\(0, GL_DEPTH_COMPONENT32,", "atlas depth DEPTH_COMPONENT32"),
    ("framework/rendering/view_lod.cpp", r"m_tex_color->setParameter\(GL_TEXTURE_WRAP_S, GL_MIRRORED_REPEAT\)", "atlas MIRRORED_REPEAT"),
    ("framework/rendering/view_lod.cpp", r"m_tex_depth->setParameter\(GL_TEXTURE_MIN_FILTER, GL_NEAREST\)", "atlas depth NEAREST"),
    ("framework/rendering/view_lod.cpp", r"glClearColor\(0\.0,1\.0,0\.0,0\.0\)", "atlas cleared to (0, 1, 0, 0)"),
    ("framework/rendering/view.cpp", r"View\{width, height, \{GL_RGBA32F\}, depth\}", "peel target RGBA32F"),
    ("framework/rendering/unit_cube.cpp", r"3, 2, 6, 7, 4, 2, 0,\s*3, 1, 6, 5, 4, 1, 0", "cube strip"),
    ("framework/rendering/screen_quad.cpp", r"-1\.0f, -1\.0f, 0\.0f, 0\.0f,\s*3\.0f, -1\.0f, 2\.0f, 0\.0f,\s*-1\.0f, 3\.0f, 0\.0f, 2\.0f", "screen triangle"),
    ("framework/rendering/volume_sampler.cpp", r"m_pos_voxels\.emplace_back\(\( x\+ 0\.5f\) \* stepX, \(y \+ 0\.5f\) \* stepY, \(z \+ 0\.5f\) \* stepZ\)", "voxel centres"),
    ("framework/rendering/volume_sampler.cpp", r"drawArrays\(GL_POINTS, 0, m_dimensions\.x \* m_dimensions\.y \* m_dimensions\.z\)", "points"),
    # ---- the default mode (round 4): DXT colour layers, bricks on, the grid and the bricks' index lists ----
    (NKA, r"if\(m_calib_files->isCompressedRGB\(\) == 1\)\{[^}]*?new TextureArray\(m_resolution_color\.x, m_resolution_color\.y, m_numLayers, GL_COMPRESSED_RGBA_S3TC_DXT1_EXT, GL_COMPRESSED_RGBA_S3TC_DXT1_EXT, GL_UNSIGNED_BYTE, m_colorsize\)", "DXT1 colour layers"),
    (NKA, r"isCompressedRGB\(\) == 5\)\{[^}]*?GL_COMPRESSED_RGBA_S3TC_DXT5_EXT, GL_COMPRESSED_RGBA_S3TC_DXT5_EXT, GL_UNSIGNED_BYTE, m_colorsize\)", "DXT5 colour layers"),
    ("framework/rendering/TextureArray.cpp", r"if\(m_storage > 0\) \{\s*m_texture->compressedImage3D\(0, m_internalFormat, m_width, m_height, m_depth, 0, m_storage \* m_depth,", "compressed array: storage x layers"),
    ("framework/rendering/TextureArray.cpp", r"m_texture\{globjects::Texture::createDefault\(GL_TEXTURE_2D_ARRAY\)\}", "colour array: createDefault = LINEAR, CLAMP_TO_EDGE"),
    (RI, r",m_use_bricks\{true\}", "bricks on by default"),
    (RI, r"if \(m_use_bricks\) \{\s*for\(auto const& index : m_bricks_occupied\) \{[^}]*?m_sampler\.sample\(m_bricks\[index\]\.indices\);", "one indexed draw per occupied brick"),
    ("framework/rendering/volume_sampler.cpp", r"m_va_samples->drawElements\(GL_POINTS, indices\.size\(\), GL_UNSIGNED_INT, indices\.data\(\)\)", "indexed points, u32 indices"),
    (RI, r"m_res_volume = glm::ceil\(glm::fvec3\{m_bbox\.getPMax\(\)\[0\] - m_bbox\.getPMin\(\)\[0\],", "res = ceil(extent / voxel)"),
    (RI, r"m_brick_size = m_voxel_size \* glm::round\(size / m_voxel_size\)", "brick size = voxel * round(size / voxel)"),
    (RI, r"while\(size\.z - start\.z  \+ min\.z > 0\.0f\) \{\s*while\(size\.y - start\.y  \+ min\.y > 0\.0f\) \{\s*while\(size\.x - start\.x  \+ min\.x > 0\.0f\)", "divideBox loop conditions"),
    (RI, r"m_bricks\.emplace_back\(start, glm::min\(glm::fvec3\{m_brick_size\}, size - start \+ min\)\)", "brick size clipped at the box"),
    (RI, r"curr_brick\.indices = m_sampler\.containedVoxels\(\(curr_brick\.pos - min\) / size, curr_brick\.size / size\)", "normalised brick -> containedVoxels"),
    (RI, r"start\.x \+= m_brick_size;", "start accumulates in float"),
    ("framework/rendering/volume_sampler.cpp", r"glm::fvec3 step\{1\.0f / glm::fvec3\{m_dimensions\}\};\s*for\(unsigned y = pos\.y / step\.y; y < \(pos\.y \+ size\.y\) / step\.y; \+\+y\) \{\s*for\(unsigned x = pos\.x / step\.x; x < \(pos\.x \+ size\.x\) / step\.x; \+\+x\) \{\s*for\(unsigned z = pos\.z / step\.z; z < \(pos\.z \+ size\.z\) / step\.z; \+\+z\)", "containedVoxels loop bounds and order"),
    ("framework/rendering/volume_sampler.cpp", r"indices\.push_back\(z \* m_dimensions\.x \* m_dimensions\.y \+ y \* m_dimensions\.x \+ x\)", "linear index"),
    ("framework/calibration/KinectCalibrationFile.cpp", r"_iscompressedrgb\(1\)", "compress_rgb defaults to 1 (DXT1)"),
    # ---- camera positions (gl_ref.host_camera_pos) ----
    ("framework/calibration/frustum.cpp", r"glm::fvec3 center_near\(\(m_corners\[0\] \+ m_corners\[1\] \+ m_corners\[2\] \+ m_corners\[3\]\) / 4\.0f\)", "near centre"),
    ("framework/calibration/frustum.cpp", r"closestPoint\(m_corners\[0\], m_corners\[0\] - m_corners\[4\], center_near, view_dir\)", "edge ray against the view axis"),
    ("framework/calibration/frustum.cpp", r"return \(p3 \+ p4 \+ p5 \+ p6\) / 4\.0f;", "mean of the four closest points"),
    ("framework/calibration/frustum.cpp", r"float sc = \(b \* e - c \* d\) / \(a \* c - b \* b\);\s*float tc = \(a \* e - b \* d\) / \(a \* c - b \* b\);", "closestPoint parameters"),
    ("framework/calibration/frustum.cpp", r"return \(pc \+ qc\) \* 0\.5f;", "midpoint of the two closest points"),
    (CV, r"points_corner\[2\] = \(curr_volume\(end_points\.x, end_points\.y, 0\)\);", "corner order"),
    (CV, r"points_corner\[7\] = \(curr_volume\(end_points\.x, 0,\s*end_points\.z\)\);", "corner order (far)"),
    # ---- the application ----
    (KC, r"glEnable\(GL_DEPTH_TEST\);\s*glDepthFunc\(GL_LESS\)", "depth test LESS"),
    (KC, r"g_buffer_shading->bindBase\(GL_UNIFORM_BUFFER, 1\)", "Settings UBO binding 1"),
    (KC, r"float\s+g_clear_color\[4\] = \{0\.0,0\.0,0\.0,0\.0\}", "window clear colour"),
]


def src(path, code_only=False):
    with open(os.path.join(REF, path)) as f:
        text = f.read()
    if code_only:            # drop lines that are commented out
        text = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("//"))
    return text


@pytest.mark.parametrize("path,pattern,note", STATEMENTS, ids=[s[2] for s in STATEMENTS])
def test_the_reference_host_code_states_what_the_harness_sets(path, pattern, note):
    assert re.search(pattern, src(path)), "%s: `%s` not found in %s -- gl_ref.py restates something the source does not say" % (note, pattern, path)


def test_binding_numbers_of_the_harness_are_the_sources():
    import gl_ref
    kc, nka, ri = src(KC), src(NKA, code_only=True), src(RI)
    assert int(re.search(r"g_nka->setStartTextureUnit\((\d+)\)", kc).group(1)) == gl_ref.NKA_UNIT
    assert int(re.search(r"g_cv->setStartTextureUnit\((\d+)\)", kc).group(1)) == gl_ref.CV_UNIT
    assert int(re.search(r"g_cv->setStartTextureUnitInv\((\d+)\)", kc).group(1)) == gl_ref.CV_INV_UNIT
    assert int(re.search(r"static int start_image_unit = (\d+);", ri).group(1)) == gl_ref.IMAGE_UNIT
    for name, off in re.findall(r'm_texture_unit_offsets\["(\w+)"\] = m_start_texture_unit(?: \+ (\d+))?;', nka):
        assert gl_ref.UNITS[name] == gl_ref.NKA_UNIT + int(off or 0), name
    for name, unit in re.findall(r'm_texture_unit_offsets\.emplace\("(\w+)", (\d+)\);', nka):
        assert gl_ref.UNITS[name] == int(unit), name
    for uni, unit in re.findall(r'm_program_integration->setUniform\("(kinect_\w+)",(\d)\)', ri):
        assert {"kinect_colors": 1, "kinect_depths": 2, "kinect_qualities": 3, "kinect_normals": 4, "kinect_silhouettes": 5}[uni] == int(unit)
    # geometry the harness draws
    verts = re.findall(r"([01])\.0f, ([01])\.0f, ([01])\.0f", re.search(r"std::vector<float> vertices\{(.*?)\};", src("framework/rendering/unit_cube.cpp"), re.S).group(1))
    assert np.array_equal(np.array(verts, np.float32).reshape(-1), gl_ref.CUBE)
    strip = [int(v) for v in re.search(r"indices \{\s*([\d,\s]+)\}", src("framework/rendering/unit_cube.cpp")).group(1).replace("\n", " ").split(",") if v.strip()]
    assert strip == gl_ref.CUBE_STRIP.tolist()
    assert int(re.search(r",m_min_voxels_per_brick\{(\d+)\}", ri).group(1)) == 10
