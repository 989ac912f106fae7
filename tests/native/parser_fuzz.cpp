// The host mirror's parsers (rgbdr_host.hpp: .ks scene files, sensor .yml files, .stream frames -- the data formats either
// side of the path) with hostile inputs under AddressSanitizer + UBSan + float-cast-overflow: truncated
// files, missing tokens, negative / huge / non-numeric numbers.  An exception is a fine answer, undefined behaviour is not.
//   g++ -std=c++14 -O1 -g -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=all
//       tests/native/parser_fuzz.cpp -Lrgbd-recon_amd -lrgbdr_hip -Wl,-rpath,$PWD/rgbd-recon_amd
#include <cstdio>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../rgbd-recon_amd/host/rgbdr_host.hpp"

using namespace rgbdr::host;

static void put(const std::string& path, const std::string& text)
{
  std::ofstream f(path, std::ios::binary);
  f << text;
}

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  const std::string dir = std::string(argv[1]) + "/";
  std::mt19937 rng(99);
  const char* numbers[] = {"512", "424", "-1", "-512,", "0", "0,", "4294967296,", "1e30", "1e30,", "nan", "nan,", "inf,", "abc", "abc,", ",", "", "3.5,", "7,",
                           "99999999999999999999,", "-0.0,", "1280,", "1080"};
  const char* keys[] = {"rgb_size:", "depth_size:", "near_far:", "compress_rgb:", "compress_depth:", "[", "]", "kinect", "bbx", "unknown:"};
  long parsed = 0, thrown = 0;
  for (int it = 0; it < 4000; ++it) {
    // a sensor yml of random tokens (sometimes a sane prefix first), sometimes cut off anywhere
    std::string yml;
    if (rng() % 2) yml = "rgb_size: [ 1280, 1080 ]\ndepth_size: [ 512, 424 ]\nnear_far: [ 0.5, 4.5 ]\n";
    const int n = (int)(rng() % 24);
    for (int k = 0; k < n; ++k) {
      yml += rng() % 3 ? numbers[rng() % (sizeof numbers / sizeof numbers[0])] : keys[rng() % (sizeof keys / sizeof keys[0])];
      yml += rng() % 5 ? " " : "\n";
    }
    if (rng() % 4 == 0 && !yml.empty()) yml.resize(rng() % yml.size());
    put(dir + "s0.yml", yml);
    try {
      CalibrationFiles cf = parseCalibrationFiles({dir + "s0.yml"});
      (void)colorFrameBytes(cf);
      (void)depthFrameBytes(cf);
      ++parsed;
    } catch (const std::exception&) {
      ++thrown;
    }
    // a .ks file of random tokens
    std::string ks;
    const int m = (int)(rng() % 16);
    for (int k = 0; k < m; ++k) {
      const unsigned r = rng() % 4;
      ks += r == 0 ? "kinect" : (r == 1 ? "bbx" : (r == 2 ? numbers[rng() % (sizeof numbers / sizeof numbers[0])] : "s0.yml"));
      ks += rng() % 5 ? " " : "\n";
    }
    put(dir + "scene.ks", ks);
    try {
      KsFile k = parseKs(dir + "scene.ks");
      (void)k.calib_filenames.size();
      ++parsed;
    } catch (const std::exception&) {
      ++thrown;
    }
  }
  try {
    parseKs(dir + "scene.txt");
    return 1;
  } catch (const std::invalid_argument&) {
  }
  try {
    parseKs("noextension");
    return 1;
  } catch (const std::exception&) {
  }
  // .stream frames: short files, zero sizes, indices past the end
  std::vector<unsigned char> buf(64), col(64), dep(64);
  put(dir + "s.stream", std::string(100, 'x'));
  for (size_t cs : {(size_t)0, (size_t)1, (size_t)40, (size_t)64})
    for (size_t ds : {(size_t)0, (size_t)7, (size_t)64})
      for (size_t idx : {(size_t)0, (size_t)1, (size_t)5, (size_t)1 << 40}) {
        try {
          readStreamFrame(dir + "s.stream", cs, ds, idx, col.data(), dep.data());
          ++parsed;
        } catch (const std::exception&) {
          ++thrown;
        }
      }
  std::printf("parser fuzz: %ld inputs parsed, %ld refused\n", parsed, thrown);
  return 0;
}
