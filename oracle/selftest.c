/* selftest.c -- TEST INFRASTRUCTURE: drives every oracle entry point on small seeded
 * inputs so the C restatement can be run under AddressSanitizer / UBSan on the CPU
 * (`make -C oracle asan`; GPU sanitizers are not available on the target pool).
 * Prints a checksum per function; exits non-zero only if a sanitizer aborts. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rgbdr_oracle.c"

static uint32_t rng_state = 12345u;
static float frand(void)
{
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)(rng_state >> 8) / 16777216.0f;
}
static double checksum(const float* a, size_t n)
{
  double s = 0;
  for (size_t i = 0; i < n; ++i)
    if (a[i] == a[i] && fabsf(a[i]) < 1e30f) s += a[i] * (double)((i % 7) + 1);
  return s;
}

int main(void)
{
  enum { W = 40, H = 33, R = 8, G = 16, N = 2 };
  float* depth = malloc(sizeof(float) * W * H);
  uint8_t* color = malloc(W * H * 3);
  float* xyz = malloc(sizeof(float) * R * R * R * 3);
  float* uv = malloc(sizeof(float) * R * R * R * 2);
  float* inv = malloc(sizeof(float) * G * G * G * 4);
  for (int i = 0; i < W * H; ++i) {
    depth[i] = frand() < 0.03f ? 0.0f : 2.0f + 0.002f * frand() + 0.004f * (float)(i % W);
    if (i == 7) depth[i] = NAN;
    if (i == 9) depth[i] = INFINITY;
    if (i == 11) depth[i] = -2.0f;
    for (int k = 0; k < 3; ++k) color[3 * i + k] = (uint8_t)(frand() * 255.0f);
  }
  for (int z = 0; z < R; ++z)
    for (int y = 0; y < R; ++y)
      for (int x = 0; x < R; ++x) {
        size_t o = ((size_t)z * R + y) * R + x;
        float d = 0.5f + 4.0f * (z + 0.5f) / R;
        xyz[3 * o] = 0.3f * ((x + 0.5f) / R * d - 0.5f * d);
        xyz[3 * o + 1] = 1.0f + 0.3f * ((y + 0.5f) / R * d - 0.5f * d);
        xyz[3 * o + 2] = d - 2.0f;
        uv[2 * o] = (x + 0.5f) / R;
        uv[2 * o + 1] = (y + 0.5f) / R;
      }
  for (int i = 0; i < G * G * G; ++i) {
    inv[4 * i] = ((i % G) + 0.5f) / G * 1.2f - 0.1f;
    inv[4 * i + 1] = (((i / G) % G) + 0.5f) / G * 1.2f - 0.1f;
    inv[4 * i + 2] = 0.37f + 0.02f * ((i / (G * G)) + 0.5f) / G;
    inv[4 * i + 3] = 1.0f;
    if (i % 17 == 0) inv[4 * i] = inv[4 * i + 1] = inv[4 * i + 2] = inv[4 * i + 3] = -1.0f;
    if (i == 5) inv[4 * i] = NAN;
    if (i == 6) inv[4 * i + 1] = 1e30f;
  }
  orc_set_threads(2);
  float* m0 = malloc(sizeof(float) * W * H);
  float* m1 = malloc(sizeof(float) * W * H);
  orc_morph(depth, W, H, 0, m0);
  orc_morph(m0, W, H, 1, m1);
  printf("morph %.6f\n", checksum(m1, W * H));
  orc_depth_params dp = {W, H, W, H, {R, R, R}, {R, R, R}, 0.5f, 4.5f, {-1, 0, -1}, {1, 2, 1}, 1, 0, 0.5f, 4.5f};
  float* rg = malloc(sizeof(float) * W * H * 2);
  float* lab = malloc(sizeof(float) * W * H * 3);
  orc_pre_depth(m1, color, xyz, uv, &dp, rg, lab);
  printf("pre_depth %.6f %.6f\n", checksum(rg, W * H * 2), checksum(lab, W * H * 3));
  float* db = malloc(sizeof(float) * W * H * 2);
  float* sil = malloc(sizeof(float) * W * H);
  orc_boundary(rg, lab, W, H, 1, db, sil);
  printf("boundary %.6f %.6f\n", checksum(db, W * H * 2), checksum(sil, W * H));
  orc_normal_params np = {W, H, {R, R, R}, {-1, 0, -1}, {1, 2, 1}, 0.25f, {8, 8, 8}};
  uint32_t bricks[512] = {0};
  float* nrm = malloc(sizeof(float) * W * H * 3);
  orc_normal(db, xyz, &np, nrm, bricks);
  unsigned bs = 0;
  for (int i = 0; i < 512; ++i) bs += bricks[i];
  printf("normal %.6f bricks %u\n", checksum(nrm, W * H * 3), bs);
  float cam[3];
  const int res[3] = {R, R, R};
  orc_camera_pos(xyz, res, cam);
  float* q = malloc(sizeof(float) * W * H);
  orc_quality(db, nrm, xyz, res, cam, W, H, q);
  printf("quality %.6f cam %.4f %.4f %.4f\n", checksum(q, W * H), cam[0], cam[1], cam[2]);
  orc_integrate_params ip = {N, W, H, {G, G, G}, 0.01f};
  const float* invs[N] = {inv, inv};
  const int inv_res[6] = {G, G, G, G, G, G};
  const float* sils[N] = {sil, sil};
  const float* dbs[N] = {db, db};
  const float* qs[N] = {q, q};
  uint8_t mask[8] = {1, 0, 1, 1, 0, 1, 1, 1};
  float* tsdf = malloc(sizeof(float) * G * G * G);
  /* brick mode: the voxels divideBox + containedVoxels list for the occupied bricks (a box whose last
   * bricks reach one index past the x and y ends: the aliasing / out-of-buffer paths run too) */
  uint8_t* vmask = (uint8_t*)malloc((size_t)G * G * G);
  int vrb[3];
  const float vbmin[3] = {-1, 0, -1}, vbmax[3] = {1, 2, 1};
  orc_brick_voxel_mask(vbmin, vbmax, 1.0f, (int[3]){G, G, G}, mask, 0, G, vmask, vrb);
  orc_integrate(&ip, invs, inv_res, sils, dbs, qs, vmask, 0, G, tsdf);
  {
    const float omin[3] = {-1.0f, 0.0f, -1.0f}, omax[3] = {1.4f, 2.4f, 1.0f};
    int od[3] = {121, 121, 100}, orb[3];
    uint8_t* occ = (uint8_t*)malloc(24 * 25 * 20);
    uint8_t* om = (uint8_t*)malloc((size_t)121 * 121 * 100);
    memset(occ, 1, 24 * 25 * 20);
    size_t outside = orc_brick_voxel_mask(omin, omax, 0.02f * roundf(0.1f / 0.02f), od, occ, 0, 100, om, orb);
    printf("overflow box: bricks %d %d %d, %zu index triples outside\n", orb[0], orb[1], orb[2], outside);
    free(occ);
    free(om);
  }
  free(vmask);
  printf("integrate %.6f\n", checksum(tsdf, G * G * G));
  uint32_t ids[512];
  float ratio;
  printf("occupied %u\n", orc_update_occupied(bricks, 512, 3, ids, &ratio));
  float planes[24];
  orc_frustum_planes(xyz, res, planes);
  const float bmin[3] = {-1, 0, -1}, bmax[3] = {1, 2, 1};
  const int vr[3] = {6, 5, 7};
  float* iv = malloc(sizeof(float) * 6 * 5 * 7 * 4);
  orc_inverse_volume(xyz, res, bmin, bmax, vr, 0, 7, iv);
  printf("inverse_volume %.6f\n", checksum(iv, 6 * 5 * 7 * 4));
  int rb[3];
  orc_divide_box(bmin, bmax, 0.1f, rb);
  printf("divide_box %d %d %d\n", rb[0], rb[1], rb[2]);
  uint8_t blk[((W + 3) / 4) * ((H + 3) / 4) * 16];
  for (size_t i = 0; i < sizeof(blk); ++i) blk[i] = (uint8_t)(frand() * 255.0f);
  uint8_t* rgb = malloc(W * H * 3);
  orc_decode_dxt(blk, W, H, 1, rgb);
  orc_decode_dxt(blk, W, H, 5, rgb);
  printf("dxt %u\n", rgb[0] + rgb[W * H * 3 - 1]);
  /* ray-march + hole fill with a simple view: identity-ish matrices looking down -z */
  orc_view v;
  memset(&v, 0, sizeof(v));
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(v.modelview, I, 64);
  memcpy(v.projection, I, 64);
  memcpy(v.normal_matrix, I, 64);
  memcpy(v.gl_normal_matrix_inv, I, 64);
  memcpy(v.vol_to_world, I, 64);
  memcpy(v.vol_to_world_inv, I, 64);
  memcpy(v.modelview_inv, I, 64);
  memcpy(v.img_to_eye, I, 64);
  v.img_to_eye[0] = 1.0f / 24;
  v.img_to_eye[5] = 1.0f / 18;
  v.camera_pos[0] = 0.5f;
  v.camera_pos[1] = 0.5f;
  v.camera_pos[2] = -1.5f;
  v.width = 24;
  v.height = 18;
  orc_raymarch_params rp = {N, W, H, W, H, {G, G, G}, 0.01f};
  const float* uvs[N] = {uv, uv};
  const int uv_res[6] = {R, R, R, R, R, R};
  const uint8_t* cols[N] = {color, color};
  float* oc = malloc(sizeof(float) * 24 * 18 * 4);
  float* od = malloc(sizeof(float) * 24 * 18);
  float* on = malloc(sizeof(float) * 24 * 18);
  for (int mode = 0; mode < 4; ++mode) {
    v.shade_mode = mode;
    orc_raymarch(&v, &rp, tsdf, invs, inv_res, uvs, uv_res, cols, dbs, qs, NULL, oc, od, on);
  }
  {
    orc_brick_grid bg = {{0.0f, 0.0f, 0.0f}, 0.125f, {8, 8, 8}};
    float* peels = malloc(sizeof(float) * 24 * 18 * 4);
    uint8_t mask512[512];
    for (int i = 0; i < 512; ++i) mask512[i] = bricks[i] >= 3u;
    orc_depth_peels(&v, &bg, bricks, mask512, peels);
    v.skip_space = 1;
    v.shade_mode = 0;
    orc_raymarch(&v, &rp, tsdf, invs, inv_res, uvs, uv_res, cols, dbs, qs, peels, oc, od, on);
    v.skip_space = 0;
    printf("peels %.6f\n", checksum(peels, 24 * 18 * 4));
  }
  printf("raymarch %.6f %.6f\n", checksum(oc, 24 * 18 * 4), checksum(on, 24 * 18));
  float* fc = malloc(sizeof(float) * 24 * 18 * 4);
  float* fd = malloc(sizeof(float) * 24 * 18);
  orc_fill_colors(oc, od, 24, 18, fc, fd, NULL);
  printf("fill_colors %.6f\n", checksum(fc, 24 * 18 * 4));
  puts("selftest done");
  return 0;
}
