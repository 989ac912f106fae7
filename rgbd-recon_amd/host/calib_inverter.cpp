// calib_inverter -- the reference's offline tool source/calib_inverter.cpp (a CGAL k-d tree +
// OpenMP job there) over the C ABI: reads the sensors and the bounding box from a `.ks` scene
// file, the forward volumes "<yml base>.cv_xyz" next to the sensor files, computes the inverse
// calibration volumes on the device (rgbdr_generate_inverse_lut) at ceil(bbox / voxel_size)
// texels and writes "<ks dir>/<basename>.cv_xyz_inv" -- same command line, same file names,
// same file format.
//   calib_inverter <scene.ks> [-s voxel_size]        (default 0.007 m, as in the reference)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rgbdr_host.hpp"

using namespace rgbdr::host;

int main(int argc, char** argv)
{
  float voxel_size = 0.007f;
  std::string ks_name;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-s" && i + 1 < argc)
      voxel_size = (float)std::atof(argv[++i]);
    else
      ks_name = a;
  }
  if (ks_name.empty() || !(voxel_size > 0.0f)) {
    std::fprintf(stderr, "usage: %s <scene.ks> [-s voxel_size]\n", argv[0]);
    return 2;
  }
  try {
    const KsFile ks = parseKs(ks_name);  // throws std::invalid_argument{"No .ks file specified"} like the tool
    if (ks.calib_filenames.empty()) throw std::invalid_argument{"no `kinect` entries in " + ks_name};
    CalibrationFiles cf = parseCalibrationFiles(ks.calib_filenames);
    uint32_t res[3];
    for (int a = 0; a < 3; ++a) res[a] = (uint32_t)std::ceil((ks.bbox.pmax[a] - ks.bbox.pmin[a]) / voxel_size);
    // the context's own TSDF grid is not used here: keep it small
    const float extent = std::fmax(ks.bbox.pmax[0] - ks.bbox.pmin[0], std::fmax(ks.bbox.pmax[1] - ks.bbox.pmin[1], ks.bbox.pmax[2] - ks.bbox.pmin[2]));
    Backend be(cf, ks.bbox, 0.01f, extent / 32.0f, extent / 4.0f);
    CalibVolumes cv(be, ks.calib_filenames);
    std::printf("using resolution %u, %u, %u\n", res[0], res[1], res[2]);
    cv.writeInverseCalibs(ks.resource_path, res);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "calib_inverter: %s\n", e.what());
    return 1;
  }
  return 0;
}
