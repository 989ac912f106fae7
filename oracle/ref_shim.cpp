// ref_shim.cpp -- TEST INFRASTRUCTURE.  C wrappers around the few pieces of the
// reference that compile stand-alone from /root/reference (header-only glm is
// vendored there): the calibration-volume reader/writer
// (framework/calibration/calibration_volume.hpp), the LUT record types and the CPU
// trilinear helper (framework/DataTypes.{h,cpp}).  Built by oracle/Makefile into
// oracle/_ref/libref_shim.so *from the sources where they lie*; nothing from the
// reference is copied into this repository.  Used by tests/test_oracle_ref.py to
// pin the oracle's file format, record layouts and trilinear interpolation, and the
// host mirror's sensor-yml scanner and .stream reader against the reference's own
// (framework/calibration/{calibration_files,KinectCalibrationFile}.cpp over the vendored
// gloost math classes, framework/io/FileBuffer.cpp).
#include <cassert>  // calibration_volume.hpp uses assert() without including it
#include <cstdint>
#include <cstring>
#include <vector>

#include <DataTypes.h>
#include <calibration/calibration_volume.hpp>
#include <squish.h>  // external/squish: the reference's own CPU decoder of its DXT colour frames
#include <string>

#include <FileBuffer.h>
#include <KinectCalibrationFile.h>
#include <calibration_files.hpp>

extern "C" {

int ref_sizeof_xyz() { return (int)sizeof(kinect::xyz); }
int ref_sizeof_uv() { return (int)sizeof(kinect::uv); }
int ref_sizeof_fvec4() { return (int)sizeof(glm::fvec4); }

// write through CalibrationVolume<T>::write
int ref_lut_write(const char* path, const uint32_t* res, const float* limits, const float* data, int floats)
{
  glm::uvec3 r{res[0], res[1], res[2]};
  glm::fvec2 l{limits[0], limits[1]};
  size_t n = (size_t)res[0] * res[1] * res[2];
  if (floats == 3) {
    std::vector<kinect::xyz> v(n);
    std::memcpy(v.data(), data, n * sizeof(kinect::xyz));
    kinect::CalibrationVolume<kinect::xyz>(r, l, v).write(path);
  } else if (floats == 2) {
    std::vector<kinect::uv> v(n);
    std::memcpy(v.data(), data, n * sizeof(kinect::uv));
    kinect::CalibrationVolume<kinect::uv>(r, l, v).write(path);
  } else if (floats == 4) {
    std::vector<glm::fvec4> v(n);
    std::memcpy(v.data(), data, n * sizeof(glm::fvec4));
    kinect::CalibrationVolume<glm::fvec4>(r, l, v).write(path);
  } else {
    return -1;
  }
  return 0;
}

// read through CalibrationVolume<T>::read; data may be null to query the header
int ref_lut_read(const char* path, uint32_t* res, float* limits, float* data, int floats)
{
  auto fill = [&](auto const& vol) {
    res[0] = vol.res().x;
    res[1] = vol.res().y;
    res[2] = vol.res().z;
    limits[0] = vol.depthLimits().x;
    limits[1] = vol.depthLimits().y;
    if (data) std::memcpy(data, vol.volume().data(), vol.volume().size() * sizeof(vol.volume()[0]));
  };
  if (floats == 3) fill(kinect::CalibrationVolume<kinect::xyz>(path));
  else if (floats == 2) fill(kinect::CalibrationVolume<kinect::uv>(path));
  else if (floats == 4) fill(kinect::CalibrationVolume<glm::fvec4>(path));
  else return -1;
  return 0;
}

// element accessor operator()(x,y,z): pins the x-fastest record order
void ref_lut_at_xyz(const char* path, unsigned x, unsigned y, unsigned z, float* out)
{
  kinect::CalibrationVolume<kinect::xyz> vol(path);
  kinect::xyz const& v = vol(x, y, z);
  out[0] = v.x;
  out[1] = v.y;
  out[2] = v.z;
}

// kinect::getTrilinear on un-normalised texel coordinates
void ref_get_trilinear(const float* data, unsigned w, unsigned h, unsigned d, float x, float y, float z, float* out)
{
  kinect::xyz r = kinect::getTrilinear((kinect::xyz*)data, w, h, d, x, y, z);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
}

// squish::DecompressImage as NetKinectArray::writeCurrentTexture calls it
// (framework/NetKinectArray.cpp:633); mode 1 = kDxt1, 5 = kDxt5; rgba out
void ref_squish_decompress(unsigned char* rgba, int width, int height, const void* blocks, int mode)
{
  squish::DecompressImage(rgba, width, height, blocks, mode == 1 ? squish::kDxt1 : squish::kDxt5);
}

// kinect::CalibrationFiles over `n` sensor yml files: out = {width, height, widthC, heightC,
// isCompressedRGB, isCompressedDepth}, near_far = {near_0, far_0, near_1, ...}
int ref_calibration_files(const char** paths, int n, unsigned* out, float* near_far)
{
  std::vector<std::string> names(paths, paths + n);
  kinect::CalibrationFiles cf(names);
  out[0] = cf.getWidth();
  out[1] = cf.getHeight();
  out[2] = cf.getWidthC();
  out[3] = cf.getHeightC();
  out[4] = cf.isCompressedRGB();
  out[5] = cf.isCompressedDepth() ? 1u : 0u;
  for (int i = 0; i < n; ++i) {
    near_far[2 * i] = cf.getCalibs()[i].getNear();
    near_far[2 * i + 1] = cf.getCalibs()[i].getFar();
  }
  return (int)cf.num();
}

// sys::FileBuffer as NetKinectArray::readFromFiles drives it (NetKinectArray.cpp:724-764):
// open("r"), setLooping, then `frames` times read(colorsize) + read(depthsize) into dst.
// Returns the number of bytes the reads delivered in total, -1 if the file does not open.
long ref_stream_read(const char* path, int looping, unsigned colorsize, unsigned depthsize, int frames, unsigned char* dst)
{
  sys::FileBuffer fb(path);
  if (!fb.open("r")) return -1;
  fb.setLooping(looping != 0);
  long total = 0;
  for (int k = 0; k < frames; ++k) {
    total += fb.read(dst, colorsize);
    dst += colorsize;
    total += fb.read(dst, depthsize);
    dst += depthsize;
  }
  fb.close();
  return total;
}

}  // extern "C"
