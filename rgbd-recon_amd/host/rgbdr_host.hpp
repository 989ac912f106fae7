// rgbdr_host.hpp -- C++ host-side mirror of the reference's operator surface for
// the TSDF-fusion hot path, layered on the C ABI of include/rgbdr.h.
//
// The reference drives the path through three in-process classes
//   kinect::CalibVolumes      framework/calibration/CalibVolumes.hpp:21-81
//   kinect::NetKinectArray    framework/NetKinectArray.h:36-116
//   kinect::Reconstruction    framework/reconstruction/reconstruction.hpp:11-37 (the base g_recons holds)
//   kinect::ReconIntegration  framework/reconstruction/recon_integration.hpp:35-103
// wired together in source/kinect_client.cpp:240-255 and called per frame in the
// order of kinect_client.cpp:572-602.  The classes below keep those names, method
// names, argument meaning and error behaviour (std::invalid_argument /
// std::out_of_range / std::runtime_error instead of status codes), so a
// maintainer can swap `kinect::` for `rgbdr::host::` at the call sites; what was
// an OpenGL texture unit becomes state of the shared rgbdr_ctx.
//
// Header-only; link with -lrgbdr_hip.  No OpenGL, no HIP types.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rgbdr.h"

namespace rgbdr {
namespace host {

// gloost::BoundingBox stand-in (getPMin / getPMax, external/gloost/BoundingBox.h:64-104)
struct BoundingBox {
  std::array<float, 3> pmin{{-1.0f, 0.0f, -1.0f}}, pmax{{1.0f, 2.2f, 1.0f}};  // default of kinect_client.cpp:208-209
  const std::array<float, 3>& getPMin() const { return pmin; }
  const std::array<float, 3>& getPMax() const { return pmax; }
};

// The scalars kinect::CalibrationFiles exposes to the hot path
// (framework/calibration/calibration_files.hpp:11-44); parseCalibrationFiles()
// below fills it from the sensor .yml files.
struct CalibrationFiles {
  std::vector<std::string> filenames;  // per-sensor yml paths ("kinect <yml>" lines of the .ks file)
  unsigned width = 512, height = 424, widthC = 512, heightC = 424;
  bool compressedDepth = false;
  int compressedRGB = 0;  // 0 RGB8, 1 DXT1 (the reference's yml default), 5 DXT5
  std::vector<float> near_, far_;
  unsigned num() const { return (unsigned)filenames.size(); }
  unsigned getWidth() const { return width; }
  unsigned getHeight() const { return height; }
  unsigned getWidthC() const { return widthC; }
  unsigned getHeightC() const { return heightC; }
  bool isCompressedDepth() const { return compressedDepth; }
  int isCompressedRGB() const { return compressedRGB; }
};

// --- scene / sensor description files -------------------------

// The `.ks` scene file as source/kinect_client.cpp:194-238 reads it: whitespace
// tokens, `kinect <yml>` (relative to the .ks directory unless absolute) and
// `bbx x0 y0 z0 x1 y1 z1`; everything else is ignored.
struct KsFile {
  std::vector<std::string> calib_filenames;
  BoundingBox bbox;
  std::string resource_path;  // directory of the .ks file, with trailing '/'
};

inline KsFile parseKs(std::string const& file_name)
{
  const std::string ext = file_name.substr(file_name.find_last_of(".") + 1);
  if (ext != "ks") throw std::invalid_argument{"No .ks file specified"};
  KsFile ks;
  const size_t slash = file_name.find_last_of("/\\");
  ks.resource_path = (slash == std::string::npos ? std::string(".") : file_name.substr(0, slash)) + '/';
  std::ifstream in(file_name);
  if (!in) throw std::invalid_argument{"cannot open " + file_name};
  std::string token;
  while (in >> token) {
    if (token == "kinect") {
      in >> token;
      if (token[0] == '/' || (token.size() > 1 && token[1] == ':'))
        ks.calib_filenames.push_back(token);
      else
        ks.calib_filenames.push_back(ks.resource_path + token);
    } else if (token == "bbx") {
      in >> ks.bbox.pmin[0] >> ks.bbox.pmin[1] >> ks.bbox.pmin[2] >> ks.bbox.pmax[0] >> ks.bbox.pmax[1] >> ks.bbox.pmax[2];
    }
  }
  return ks;
}

// The sensor `.yml` token scanner of KinectCalibrationFile::parse
// (framework/calibration/KinectCalibrationFile.cpp:166-353): a key token, then a
// stand-alone "[" token, then values of which all but the last carry a trailing
// comma.  Only the keys the hot path needs are read; defaults as in the
// reference (:88-95): near 0.3, far 7.0, compress_rgb 1 (DXT1), compress_depth 0.
// The reference converts the tokens with static_cast<unsigned>(float) (KinectCalibrationFile.cpp:612-628 feeding
// :190-240), which is undefined for negative, huge and non-numeric values; a damaged or hostile yml gets 0 here --
// a size the backend then refuses -- instead (tests/native/parser_fuzz.cpp).
inline unsigned toUnsigned(float v) { return (v >= 0.0f && v < 4294967296.0f) ? (unsigned)v : 0u; }

inline CalibrationFiles parseCalibrationFiles(std::vector<std::string> const& calib_filenames)
{
  if (calib_filenames.empty()) throw std::invalid_argument{"no calibration files"};
  CalibrationFiles cf;
  cf.filenames = calib_filenames;
  auto advance = [](std::ifstream& f, const char* what) {
    std::string t;
    while (f >> t)
      if (t == what) return;
  };
  auto komma = [](std::ifstream& f) {  // getNextTokenAsFloat: drop the trailing comma
    std::string t;
    f >> t;
    return (float)std::atof(t.substr(0, t.empty() ? 0 : t.size() - 1).c_str());
  };
  auto plain = [](std::ifstream& f) {  // getNextFloat
    std::string t;
    f >> t;
    return (float)std::atof(t.c_str());
  };
  for (size_t i = 0; i < calib_filenames.size(); ++i) {
    std::ifstream f(calib_filenames[i]);
    if (!f) throw std::invalid_argument{"cannot open " + calib_filenames[i]};
    float near_ = 0.3f, far_ = 7.0f;
    unsigned w = 0, h = 0, wc = 0, hc = 0;
    int crgb = 1;
    bool cdepth = false;
    std::string token;
    while (f >> token) {
      if (token == "rgb_size:") {
        advance(f, "[");
        wc = toUnsigned(komma(f));
        hc = toUnsigned(plain(f));
      } else if (token == "depth_size:") {
        advance(f, "[");
        w = toUnsigned(komma(f));
        h = toUnsigned(plain(f));
      } else if (token == "near_far:") {
        advance(f, "[");
        near_ = komma(f);
        far_ = plain(f);
      } else if (token == "compress_rgb:") {
        advance(f, "[");
        crgb = (int)toUnsigned(komma(f));
        plain(f);
      } else if (token == "compress_depth:") {
        advance(f, "[");
        cdepth = (bool)(toUnsigned(komma(f)));
        plain(f);
      }
    }
    cf.near_.push_back(near_);
    cf.far_.push_back(far_);
    if (i == 0) {  // element [0] decides sizes and compression of all sensors (calibration_files.cpp:26-33)
      cf.width = w;
      cf.height = h;
      cf.widthC = wc;
      cf.heightC = hc;
      cf.compressedRGB = crgb;
      cf.compressedDepth = cdepth;
    }
  }
  return cf;
}

// frame sizes of NetKinectArray::init (NetKinectArray.cpp:120-144)
inline size_t colorFrameBytes(CalibrationFiles const& cf)
{
  const size_t blocks = (size_t)((cf.widthC + 3) / 4) * ((cf.heightC + 3) / 4);
  if (cf.compressedRGB == 1) return blocks * 8;
  if (cf.compressedRGB == 5) return blocks * 16;
  return (size_t)cf.widthC * cf.heightC * 3;
}
inline size_t depthFrameBytes(CalibrationFiles const& cf)
{
  return (size_t)cf.width * cf.height * (cf.compressedDepth ? 1 : 4);
}

inline void check(rgbdr_ctx* ctx, int rc)
{
  if (rc == RGBDR_OK) return;
  const std::string msg = std::string(rgbdr_status_string(rc)) + ": " + rgbdr_last_error(ctx);
  switch (rc) {
    case RGBDR_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
    case RGBDR_ERR_OUT_OF_RANGE: throw std::out_of_range(msg);
    case RGBDR_ERR_NO_MEMORY: throw std::bad_alloc();  // what the reference's own containers would have thrown
    default: throw std::runtime_error(msg);
  }
}

// Owns the rgbdr_ctx the three mirrored classes share.
class Backend {
 public:
  Backend(const CalibrationFiles& cf, const BoundingBox& bbox, float limit = 0.01f, float voxel = 0.01f,
          float brick = 0.1f, int device = 0, int slab_rank = 0, int slab_count = 1)
  {
    if (cf.num() == 0) throw std::invalid_argument("no sensors");
    rgbdr_config c{};
    c.struct_size = sizeof(c);
    c.num_sensors = (int32_t)cf.num();
    c.depth_w = (int32_t)cf.getWidth();
    c.depth_h = (int32_t)cf.getHeight();
    c.color_w = (int32_t)cf.getWidthC();
    c.color_h = (int32_t)cf.getHeightC();
    for (int a = 0; a < 3; ++a) {
      c.bbox_min[a] = bbox.getPMin()[a];
      c.bbox_max[a] = bbox.getPMax()[a];
    }
    c.voxel_size = voxel;
    c.brick_size = brick;
    c.tsdf_limit = limit;
    c.min_voxels_per_brick = 10;  // recon_integration.cpp:62
    c.flags = RGBDR_FLAGS_DEFAULT;
    c.compress_depth = cf.isCompressedDepth() ? 1 : 0;
    c.compress_rgb = cf.isCompressedRGB();
    for (unsigned i = 0; i < RGBDR_MAX_SENSORS; ++i) {
      c.near_[i] = i < cf.near_.size() ? cf.near_[i] : 0.3f;  // KinectCalibrationFile defaults
      c.far_[i] = i < cf.far_.size() ? cf.far_[i] : 7.0f;
    }
    c.slab_rank = slab_rank;
    c.slab_count = slab_count;
    rgbdr_ctx* raw = nullptr;
    check(nullptr, rgbdr_create(&c, device, &raw));
    m_ctx.reset(raw, rgbdr_destroy);
    m_num = cf.num();
    m_cfg = c;
  }
  rgbdr_ctx* ctx() const { return m_ctx.get(); }
  unsigned num() const { return m_num; }
  rgbdr_config const& config() const { return m_cfg; }

 private:
  std::shared_ptr<rgbdr_ctx> m_ctx;
  unsigned m_num = 0;
  rgbdr_config m_cfg{};
};

// TimerDatabase (framework/rendering/timer_database.{hpp,cpp}): the application reads duration(name) every frame for its GUI
// (source/kinect_client.cpp:431-481) and writes the means, minima and maxima as three CSV files when it quits (:835-851).
// The library times its own launches under the reference's names (rgbdr_timer_ns); this class keeps the reference's
// statistics -- the fold of TimerDatabase::begin, timer_database.cpp:26-41, `else if` and all -- and its file format on
// top of them.  The reference folds a timer's previous interval when the timer is started again; there is no begin() on this
// side (the library brackets its launches itself), so the host calls sample() once per frame after the frame's calls.
// "3recon" (ReconIntegration::drawF as a whole, recon_integration.cpp:151-175) is the sum of the intervals it spans:
// "brickdraw" + "draw" + "holefill".  Durations are nanoseconds, like TimerGPU's GL_TIMESTAMP differences.
class TimerDatabase {
 public:
  explicit TimerDatabase(Backend& be) : m_be(be)
  {
    // the timers the reference registers (NetKinectArray.cpp:211-216, recon_integration.cpp:146-148, reconstruction.cpp:25-26)
    for (const char* n : {"morph", "bilateral", "boundary", "normal", "quality", "1preprocess", "holefill", "2integrate", "brickdraw",
                          "3recon", "draw"})
      addTimer(n);
    check(m_be.ctx(), rgbdr_enable_timers(m_be.ctx(), 1));
  }
  void addTimer(std::string const& name)
  {
    m_means.emplace(name, 0.0);
    m_nums.emplace(name, 0);
    m_extrema.emplace(name, std::make_pair(std::numeric_limits<double>::infinity(), 0.0));
  }
  // last completed interval in ns; 0 for a pass that has not run yet (the reference's query returns 0 - 0 there)
  double duration(std::string const& name) const
  {
    if (m_means.find(name) == m_means.end()) throw std::out_of_range("TimerDatabase: no timer " + name);   // std::map::at
    if (name == "3recon") return last("brickdraw") + last("draw") + last("holefill");
    return last(name);
  }
  void sample()
  {
    for (auto& kv : m_means) {
      const double d = duration(kv.first);
      if (!(d > 0.0)) continue;                     // never ran: nothing to fold
      std::size_t& num = m_nums.at(kv.first);
      kv.second = (kv.second * (double)num + d) / (double)(num + 1);
      auto& extremum = m_extrema.at(kv.first);
      if (extremum.first > d) {
        extremum.first = d;
      } else if (extremum.second < d) {
        extremum.second = d;
      }
      num += 1;
    }
  }
  double mean(std::string const& name) const { return m_means.at(name); }
  double getNum(std::string const& name) const { return (double)m_nums.at(name); }
  // "<path>mean_<file>", "<path>min_<file>", "<path>max_<file>": a header line `timer,"name",...` in map order and one
  // data line that starts with the part of <file> before its first comma (the configuration's name in the reference's
  // file names), values in ms (timer_database.cpp:59-121)
  void writeMean(std::string const& file_name) const
  {
    write(file_name, "mean_", [&](std::string const& n) { return m_means.at(n); });
  }
  void writeMin(std::string const& file_name) const
  {
    write(file_name, "min_", [&](std::string const& n) { return m_extrema.at(n).first; });
  }
  void writeMax(std::string const& file_name) const
  {
    write(file_name, "max_", [&](std::string const& n) { return m_extrema.at(n).second; });
  }

 private:
  double last(std::string const& name) const
  {
    uint64_t ns = 0;
    return rgbdr_timer_ns(m_be.ctx(), name.c_str(), &ns) == RGBDR_OK ? (double)ns : 0.0;
  }
  template <typename F>
  void write(std::string const& file_name, const char* prefix, F value) const
  {
    const std::size_t pos = file_name.find_last_of('/');
    const std::string filename = file_name.substr(pos + 1), path = file_name.substr(0, pos + 1);
    const std::string name = filename.substr(0, filename.find_first_of(','));
    std::ofstream file(path + prefix + filename);
    if (!file) throw std::runtime_error("cannot write " + path + prefix + filename);
    file << "timer";
    for (auto const& kv : m_means) file << ",\"" << kv.first << "\"";
    file << std::endl << name;
    for (auto const& kv : m_means) file << "," << value(kv.first) / 1000000.0f;
    file << std::endl;
  }
  Backend& m_be;
  std::map<std::string, std::size_t> m_nums;
  std::map<std::string, double> m_means;
  std::map<std::string, std::pair<double, double>> m_extrema;
};

// kinect::CalibVolumes
class CalibVolumes {
 public:
  // calib_volume_files are the yml paths; "<base>.cv_xyz" / "<base>.cv_uv" are read
  // next to them (CalibVolumes.cpp:34-39)
  CalibVolumes(Backend& be, std::vector<std::string> const& calib_volume_files) : m_be(be)
  {
    for (auto const& f : calib_volume_files) {
      if (f.size() < 3) throw std::invalid_argument("calibration file name too short");
      const std::string base = f.substr(0, f.size() - 3);
      m_cv_xyz_filenames.push_back(base + "cv_xyz");
      m_cv_uv_filenames.push_back(base + "cv_uv");
    }
    for (unsigned i = 0; i < m_cv_xyz_filenames.size(); ++i)
      check(m_be.ctx(), rgbdr_load_calibration_files(m_be.ctx(), (int)i, m_cv_xyz_filenames[i].c_str(),
                                                     m_cv_uv_filenames[i].c_str(), nullptr));
  }
  // "<path><basename>.cv_xyz_inv" (CalibVolumes.cpp:64-80)
  void loadInverseCalibs(std::string const& path)
  {
    for (unsigned i = 0; i < m_cv_xyz_filenames.size(); ++i) {
      const std::string& s = m_cv_xyz_filenames[i];
      const std::string name = s.substr(s.find_last_of("/\\") + 1);
      const std::string in = path + name + "_inv";
      check(m_be.ctx(), rgbdr_load_calibration_files(m_be.ctx(), (int)i, nullptr, nullptr, in.c_str()));
    }
  }
  // What the offline tool `calib_inverter` (CalibrationInverter::calculateInverseVolumes,
  // calibration_inverter.cpp:99-155) followed by loadInverseCalibs would provide, computed
  // on the device at the grid resolution from the cv_xyz volumes already loaded.
  void computeInverseCalibs(int window = 2)
  {
    for (unsigned i = 0; i < m_cv_xyz_filenames.size(); ++i)
      check(m_be.ctx(), rgbdr_compute_inverse_calibration(m_be.ctx(), (int)i, window));
  }
  // CalibrationInverter::calculateInverseVolumes + writeInverseVolumes (calibration_inverter.cpp:
  // 29-36, 99-155) on the device: a volume of `res` texels per sensor, written as
  // "<path><basename>.cv_xyz_inv" in the calibration-volume format (uvec3 res, the fixed
  // depth limits (0.5, 4.5) the tool stores, then RGBA32F records x-fastest).
  void writeInverseCalibs(std::string const& path, const uint32_t res[3], int window = 2) const
  {
    const size_t n = (size_t)res[0] * res[1] * res[2];
    std::vector<float> vol(n * 4);
    for (unsigned i = 0; i < m_cv_xyz_filenames.size(); ++i) {
      check(m_be.ctx(), rgbdr_generate_inverse_lut(m_be.ctx(), (int)i, res, window, vol.data()));
      const std::string& s = m_cv_xyz_filenames[i];
      const std::string out = path + s.substr(s.find_last_of("/\\") + 1) + "_inv";
      FILE* f = std::fopen(out.c_str(), "wb");
      if (!f) throw std::runtime_error("cannot write " + out);
      const float limits[2] = {0.5f, 4.5f};
      const bool ok = std::fwrite(res, sizeof(uint32_t), 3, f) == 3 && std::fwrite(limits, sizeof(float), 2, f) == 2 &&
                      std::fwrite(vol.data(), sizeof(float), vol.size(), f) == vol.size();
      std::fclose(f);
      if (!ok) throw std::runtime_error("short write to " + out);
    }
  }
  Backend& backend() const { return m_be; }
  // What the reference hands the other drawing modes as texture units and uniforms (CalibVolumes.cpp:90-96, 146-159;
  // recon_calibs.cpp:25-37): here the volumes where they live on the device, their resolutions and depth limits.
  rgbdr_calibration_device_view deviceVolumes(unsigned sensor) const  // getXYZVolumeUnits()[i] / getUVVolumeUnits()[i]
  {
    rgbdr_calibration_device_view v{};
    check(m_be.ctx(), rgbdr_device_calibration(m_be.ctx(), (int)sensor, &v));
    return v;
  }
  std::array<uint32_t, 3> getVolumeRes() const  // m_data_volumes_xyz_inv[0].res() (CalibVolumes.cpp:90-92): (0, 0, 0) before loadInverseCalibs
  {
    const rgbdr_calibration_device_view v = deviceVolumes(0);
    return {{v.inv_res[0], v.inv_res[1], v.inv_res[2]}};
  }
  std::array<float, 2> getDepthLimits(unsigned sensor) const
  {
    const rgbdr_calibration_device_view v = deviceVolumes(sensor);
    return {{v.depth_limits[0], v.depth_limits[1]}};
  }
  size_t numSensors() const { return m_cv_xyz_filenames.size(); }
  std::vector<std::array<float, 3>> getCameraPositions() const
  {
    std::vector<std::array<float, 3>> out(m_cv_xyz_filenames.size());
    for (unsigned i = 0; i < out.size(); ++i) check(m_be.ctx(), rgbdr_get_camera_position(m_be.ctx(), (int)i, out[i].data()));
    return out;
  }

 private:
  Backend& m_be;
  std::vector<std::string> m_cv_xyz_filenames, m_cv_uv_filenames;
};

// One frame of a sensor's ".stream" recording: [colorsize bytes][depthsize bytes] repeated
// (NetKinectArray.cpp:745-763 through sys::FileBuffer::read, io/FileBuffer.cpp:108-128, with
// looping off as the reference sets it: a read past the end is a short read).  Frame
// `index` is what the reference's reader delivers with its (index+1)-th pair of reads.
inline void readStreamFrame(std::string const& path, size_t colorsize, size_t depthsize, size_t index, unsigned char* color,
                            unsigned char* depth)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("error opening " + path);
  bool ok = std::fseek(f, (long)((colorsize + depthsize) * index), SEEK_SET) == 0 &&
            std::fread(color, 1, colorsize, f) == colorsize && std::fread(depth, 1, depthsize, f) == depthsize;
  std::fclose(f);
  if (!ok) throw std::runtime_error("short read from " + path);
}

// What the reference's other programs sample from the texture units NetKinectArray binds (setStartTextureUnit /
// bindToTextureUnits, NetKinectArray.cpp:430-465; kinect_client.cpp:245 passes start unit 1): the unit's NAME in
// m_texture_unit_offsets -> the image of the C ABI.  "depth" is m_textures_depth_b after processTextures (:379).
inline int imageOfTextureUnit(std::string const& name)
{
  if (name == "color") return RGBDR_IMG_COLOR;
  if (name == "depth") return RGBDR_IMG_DEPTH_B_RG;
  if (name == "quality") return RGBDR_IMG_QUALITY;
  if (name == "normal") return RGBDR_IMG_NORMAL;
  if (name == "silhouette") return RGBDR_IMG_SILHOUETTE;
  if (name == "morph_depth") return RGBDR_IMG_DEPTH_MORPH;
  if (name == "color_lab") return RGBDR_IMG_LAB;
  if (name == "raw_depth") return RGBDR_IMG_DEPTH_RAW;
  throw std::out_of_range("no texture unit named " + name);  // m_texture_unit_offsets.at(), NetKinectArray.cpp:248
}
// zero-copy view of one sensor's image layer (rgbdr_device_image): the counterpart of sampling the texture unit
inline rgbdr_image_device_view deviceImage(Backend const& be, int which, unsigned sensor)
{
  rgbdr_image_device_view v{};
  check(be.ctx(), rgbdr_device_image(be.ctx(), which, (int)sensor, &v));
  return v;
}
inline std::vector<float> readbackImage(Backend const& be, int which, unsigned sensor)
{
  rgbdr_image_device_view v = deviceImage(be, which, sensor);
  if (v.element_bytes != 4) throw std::invalid_argument("readbackImage: not a float image (use readbackColor)");
  std::vector<float> out((size_t)v.width * v.height * v.channels);
  check(be.ctx(), rgbdr_readback_image(be.ctx(), which, (int)sensor, out.data()));
  return out;
}
inline std::vector<unsigned char> readbackColor(Backend const& be, unsigned sensor)
{
  std::vector<unsigned char> out((size_t)be.config().color_w * be.config().color_h * 3);
  check(be.ctx(), rgbdr_readback_color(be.ctx(), (int)sensor, out.data()));
  return out;
}

// kinect::NetKinectArray (ingest + the five pre_* passes).  The ZeroMQ reader
// thread is transport and stays outside; frames arrive through update().
class NetKinectArray {
 public:
  explicit NetKinectArray(Backend& be) : m_be(be) {}
  // unsigned NetKinectArray::getTextureUnit(std::string const& name) (NetKinectArray.cpp:247-249) handed a consumer a
  // unit number to point its sampler at; here the same name yields the image itself: a zero-copy device view
  // (deviceImage) or a host copy (readbackImage / readbackColor).  `which` is an RGBDR_IMG_* value.
  rgbdr_image_device_view deviceImage(int which, unsigned sensor) const { return host::deviceImage(m_be, which, sensor); }
  rgbdr_image_device_view deviceImage(std::string const& unit_name, unsigned sensor) const
  {
    return host::deviceImage(m_be, imageOfTextureUnit(unit_name), sensor);
  }
  std::vector<float> readbackImage(int which, unsigned sensor) const { return host::readbackImage(m_be, which, sensor); }
  std::vector<unsigned char> readbackColor(unsigned sensor) const { return host::readbackColor(m_be, sensor); }
  // the reference's constructor arguments that concern the hot path (NetKinectArray.cpp:42: serverport and
  // slaveport are transport and stay outside); both objects share the backend of `vols`
  NetKinectArray(CalibrationFiles const* calibs, CalibVolumes const* vols) : m_be(vols->backend())
  {
    if (!calibs || calibs->num() != m_be.num()) throw std::invalid_argument("calibration files do not match the calibration volumes");
  }
  // bool NetKinectArray::update(): upload the newest frame set
  bool update(const void* depth_all_sensors, const void* color_all_sensors)
  {
    check(m_be.ctx(), rgbdr_upload_frame(m_be.ctx(), depth_all_sensors, color_all_sensors));
    return true;
  }
  void processTextures() { check(m_be.ctx(), rgbdr_process_textures(m_be.ctx())); }
  // The reference re-runs processTextures() inside these three setters without
  // clearing the brick counters; here the caller re-runs the
  // process_textures() sequence explicitly.
  void filterTextures(bool on) { check(m_be.ctx(), rgbdr_filter_textures(m_be.ctx(), on)); }
  void useProcessedDepths(bool on) { check(m_be.ctx(), rgbdr_use_processed_depths(m_be.ctx(), on)); }
  void refineBoundary(bool on) { check(m_be.ctx(), rgbdr_refine_boundary(m_be.ctx(), on)); }

  // the mapped back PBO the reader thread memcpys each message into (NetKinectArray.cpp:
  // 511-541) and the swap in update() (:226-238): page-locked, so the upload is a true DMA
  struct MappedFrame {
    unsigned char *depth = nullptr, *color = nullptr;
    size_t depth_bytes = 0, color_bytes = 0;
  };
  MappedFrame mapBackBuffer()
  {
    MappedFrame m;
    void *d = nullptr, *c = nullptr;
    check(m_be.ctx(), rgbdr_map_frame_buffer(m_be.ctx(), &d, &c, &m.depth_bytes, &m.color_bytes));
    m.depth = (unsigned char*)d;
    m.color = (unsigned char*)c;
    return m;
  }
  bool updateFromMapped()
  {
    check(m_be.ctx(), rgbdr_upload_mapped_frame(m_be.ctx()));
    return true;
  }
  // One message of the server as the reader thread takes it apart (NetKinectArray.cpp:511-541):
  // per sensor [colorsize bytes][depthsize bytes], K1 K2 ... KN; its first 8 bytes double as the
  // frame time (they overlap sensor 0's colour data -- the reference reads them as a double and
  // copies them as pixels all the same).  Scatters into the mapped back buffer and uploads.
  bool updateFromMessage(const void* message, size_t bytes, unsigned num_sensors)
  {
    if (num_sensors == 0 || num_sensors != (unsigned)m_be.config().num_sensors)
      throw std::invalid_argument{"message for another number of sensors than the context's"};
    MappedFrame m = mapBackBuffer();
    const size_t colorsize = m.color_bytes / num_sensors, depthsize = m.depth_bytes / num_sensors;
    if (bytes != (colorsize + depthsize) * num_sensors) throw std::invalid_argument{"message size does not match the sensor set"};
    const unsigned char* src = (const unsigned char*)message;
    std::memcpy(&m_curr_frametime, src, sizeof(double));
    for (unsigned i = 0; i < num_sensors; ++i) {
      std::memcpy(m.color + i * colorsize, src, colorsize);
      src += colorsize;
      std::memcpy(m.depth + i * depthsize, src, depthsize);
      src += depthsize;
    }
    return updateFromMapped();
  }
  double getCurrentFrameTime() const { return m_curr_frametime; }
  // glm::uvec2 NetKinectArray::getDepthResolution() / getColorResolution() (NetKinectArray.h:61-62)
  std::array<unsigned, 2> getDepthResolution() const { return {{(unsigned)m_be.config().depth_w, (unsigned)m_be.config().depth_h}}; }
  std::array<unsigned, 2> getColorResolution() const { return {{(unsigned)m_be.config().color_w, (unsigned)m_be.config().color_h}}; }

  // NetKinectArray::readFromFiles (NetKinectArray.cpp:724-764): one ".stream" file
  // per sensor, frames of [colorsize bytes][depthsize bytes]; reads frame `index`
  // of every file into contiguous per-sensor buffers and uploads them.
  bool readFromFiles(std::vector<std::string> const& stream_files, size_t colorsize, size_t depthsize, size_t index = 0)
  {
    // update() reads num_sensors frames of the context's own sizes: anything else would read past these buffers
    if (stream_files.size() != (size_t)m_be.config().num_sensors) throw std::invalid_argument{"one stream file per sensor expected"};
    {
      const rgbdr_config& c = m_be.config();
      const size_t blocks = (size_t)((c.color_w + 3) / 4) * ((c.color_h + 3) / 4);
      const size_t want_c = c.compress_rgb == 1 ? blocks * 8 : (c.compress_rgb == 5 ? blocks * 16 : (size_t)c.color_w * c.color_h * 3);
      const size_t want_d = (size_t)c.depth_w * c.depth_h * (c.compress_depth ? 1 : 4);
      if (colorsize != want_c || depthsize != want_d)
        throw std::invalid_argument{"frame sizes do not match the calibration files' image sizes and compression"};
    }
    std::vector<unsigned char> color(colorsize * stream_files.size()), depth(depthsize * stream_files.size());
    for (size_t i = 0; i < stream_files.size(); ++i)
      readStreamFrame(stream_files[i], colorsize, depthsize, index, color.data() + i * colorsize, depth.data() + i * depthsize);
    return update(depth.data(), color.data());
  }

 private:
  Backend& m_be;
  double m_curr_frametime = 0.0;
};

// kinect::Reconstruction (framework/reconstruction/reconstruction.hpp:11-37, reconstruction.cpp:14-62): the base the
// application holds its drawing modes by -- std::vector<std::shared_ptr<Reconstruction>> g_recons and
// g_recons.at(g_recon_mode)->drawF() (source/kinect_client.cpp:99, 251-255, 617).  Same virtuals, same
// defaults (reload / resize do nothing, setViewportOffset complains on stderr).  The reference's draw() reads
// its camera from OpenGL state (glGetFloatv(GL_MODELVIEW_MATRIX / GL_PROJECTION_MATRIX), glGetIntegerv(GL_VIEWPORT),
// recon_integration.cpp:182-203); there is no GL state behind this backend, so the host states it with
// setView() before drawF(): the one addition to the interface.
class Reconstruction {
 public:
  Reconstruction(CalibrationFiles const& cfs, CalibVolumes const* cv, BoundingBox const& bbox)
      : m_cv(cv), m_cf(&cfs), m_tex_width(cfs.getWidth()), m_tex_height(cfs.getHeight()), m_num_kinects(cfs.num()), m_bbox(bbox)
  {
  }
  virtual ~Reconstruction() = default;

  virtual void draw() = 0;
  // reconstruction.cpp:35-39: TimerDatabase "draw" around draw() (the library brackets its launches itself)
  virtual void drawF() { draw(); }
  virtual void reload() {}
  // "mustnt be implemented by children without fbos": the base ignores it
  virtual void resize(std::size_t /*width*/, std::size_t /*height*/) {}
  void setColorMaskMode(unsigned mode) { m_color_mask_mode = mode; }
  virtual void setViewportOffset(float /*x*/, float /*y*/)
  {
    std::fprintf(stderr, "Reconstruction::setViewportOffset(float x, float y) -> implement me in derived class!\n");
  }
  // what the GL matrix stack and viewport hold when the reference's draw() runs (rgbdr_view: modelview, projection,
  // NormalMatrix, img_to_eye ... exactly the uniforms ReconIntegration::draw derives from them, :177-241)
  void setView(rgbdr_view const& view) { m_view = view; m_have_view = true; }
  rgbdr_view const& view() const { return m_view; }

 protected:
  // Backend-only construction (no CalibrationFiles object at hand): sizes come from the backend's configuration
  explicit Reconstruction(Backend const& be)
      : m_cv(nullptr), m_cf(nullptr), m_tex_width((unsigned)be.config().depth_w), m_tex_height((unsigned)be.config().depth_h),
        m_num_kinects(be.num())
  {
    for (int a = 0; a < 3; ++a) {
      m_bbox.pmin[a] = be.config().bbox_min[a];
      m_bbox.pmax[a] = be.config().bbox_max[a];
    }
  }
  // The reference's drawing modes reach the frame through texture units 1-7, which NetKinectArray binds globally
  // (recon_trigrid.cpp:30-33, recon_points.cpp, tsdf_raymarch: kinect_colors 1, kinect_depths 2, kinect_qualities 3,
  // kinect_normals 4).  There is no global binding here: a mode constructed from the reference's triple
  // (cfs, cv, bbox) reaches the same images through the backend its calibration volumes live in.
  rgbdr_image_device_view frameImage(std::string const& unit_name, unsigned sensor) const
  {
    if (!m_cv) throw std::logic_error("this Reconstruction was not constructed from calibration volumes");
    return host::deviceImage(m_cv->backend(), imageOfTextureUnit(unit_name), sensor);
  }
  CalibVolumes const* m_cv;
  CalibrationFiles const* m_cf;
  unsigned m_tex_width, m_tex_height, m_num_kinects;
  BoundingBox m_bbox;
  unsigned m_color_mask_mode = 0;
  rgbdr_view m_view{};
  bool m_have_view = false;
};

// kinect::ReconIntegration : public Reconstruction (framework/reconstruction/recon_integration.hpp:35-103)
class ReconIntegration : public Reconstruction {
 public:
  explicit ReconIntegration(Backend& be) : Reconstruction(be), m_be(be) {}
  // ReconIntegration(CalibrationFiles const&, CalibVolumes const*, gloost::BoundingBox const&, float limit,
  // float size), recon_integration.cpp:30, as kinect_client.cpp:252 calls it.  The volume lives in the
  // backend `cv` was built on, whose box must be `bbox`; limit and voxel size are applied like the
  // constructor's setTsdfLimit / setVoxelSize (a changed voxel size reallocates the grid: load the
  // inverse calibration volumes afterwards).
  ReconIntegration(CalibrationFiles const& cfs, CalibVolumes const* cv, BoundingBox const& bbox, float limit, float size)
      : Reconstruction(cfs, cv, bbox), m_be(cv->backend())
  {
    if (cfs.num() != m_be.num()) throw std::invalid_argument("calibration files do not match the calibration volumes");
    for (int a = 0; a < 3; ++a)
      if (bbox.getPMin()[a] != m_be.config().bbox_min[a] || bbox.getPMax()[a] != m_be.config().bbox_max[a])
        throw std::invalid_argument("bounding box differs from the one the backend was created with");
    setTsdfLimit(limit);
    if (size != m_be.config().voxel_size) setVoxelSize(size);
  }
  void integrate() { check(m_be.ctx(), rgbdr_integrate(m_be.ctx())); }
  void clearOccupiedBricks() const { check(m_be.ctx(), rgbdr_clear_occupied_bricks(m_be.ctx())); }
  void updateOccupiedBricks() { check(m_be.ctx(), rgbdr_update_occupied_bricks(m_be.ctx())); }
  void setVoxelSize(float size) { check(m_be.ctx(), rgbdr_set_voxel_size(m_be.ctx(), size)); }
  void setTsdfLimit(float limit) { check(m_be.ctx(), rgbdr_set_tsdf_limit(m_be.ctx(), limit)); }
  void setBrickSize(float size) { check(m_be.ctx(), rgbdr_set_brick_size(m_be.ctx(), size)); }
  void setUseBricks(bool active) { check(m_be.ctx(), rgbdr_set_use_bricks(m_be.ctx(), active)); }
  // full sweeps only, same volume bit for bit (not in the reference: see RGBDR_FLAG_ELIDE_STORES / _SKIP_BACKGROUND)
  void setElideStores(bool active) { check(m_be.ctx(), rgbdr_set_elide_stores(m_be.ctx(), active)); }
  void setSkipBackground(bool active) { check(m_be.ctx(), rgbdr_set_skip_background(m_be.ctx(), active)); }
  void setMinVoxelsPerBrick(unsigned i) { check(m_be.ctx(), rgbdr_set_min_voxels_per_brick(m_be.ctx(), i)); }
  unsigned numBricks() const { return rgbdr_num_bricks(m_be.ctx()); }
  float occupiedRatio() const { return rgbdr_occupied_ratio(m_be.ctx()); }
  float getBrickSize() const { return rgbdr_get_brick_size(m_be.ctx()); }
  void setColorFilling(bool active) { m_fill_holes = active; }
  void setSpaceSkip(bool active) { m_skip_space = active; }

  // What the reference leaves in the framebuffer: RGBA32F colour, gl_FragDepth, the tex_num_samples image.
  struct Frame {
    int width = 0, height = 0;
    std::vector<float> color, depth, num_samples;
  };
  // ---- the Reconstruction interface (source/kinect_client.cpp:617: g_recons.at(g_recon_mode)->drawF()) ----
  // draw(): the ray-march of recon_integration.cpp:177-241 for the view set with setView(), into frame().
  // glColorMask of the anaglyph modes (:211-217, only without hole filling): mode 1 writes red only, mode 2
  // green and blue only -- the other channels of frame() keep what the previous draw left there.
  void draw() override
  {
    if (!m_have_view) throw std::invalid_argument("ReconIntegration::draw() before setView()");
    if (m_color_mask_mode > 0 && !m_fill_holes && m_frame.width == m_view.width && m_frame.height == m_view.height) {
      Frame now;
      draw(m_view, now);
      const size_t n = (size_t)now.width * now.height;
      for (size_t i = 0; i < n; ++i) {
        if (!(now.depth[i] < 1.0f)) continue;  // discarded fragment: nothing is written
        if (m_color_mask_mode == 1) {
          m_frame.color[4 * i] = now.color[4 * i];
        } else {
          m_frame.color[4 * i + 1] = now.color[4 * i + 1];
          m_frame.color[4 * i + 2] = now.color[4 * i + 2];
        }
        m_frame.depth[i] = now.depth[i];
      }
      m_frame.num_samples = now.num_samples;
      return;
    }
    draw(m_view, m_frame);
  }
  // drawF(), recon_integration.cpp:151-175: depth limits for space skipping when m_skip_space && m_use_bricks
  // (inside the library's ray-march, from view.skip_space), Reconstruction::drawF(), fillColors() when m_fill_holes
  // The last pass of fillColors draws into the WINDOW with GL_LESS against its cleared depth (:314, kinect_client.cpp:
  // 605-614,994): a fragment of depth exactly 1 -- a ray that hit nothing -- fails the test, the window keeps its clear
  // colour (g_clear_color, kinect_client.cpp:55: zeros unless `-c` is given).  frame() is that window.
  void drawF() override
  {
    Reconstruction::drawF();
    if (!m_fill_holes) return;
    check(m_be.ctx(), rgbdr_fill_colors(m_be.ctx(), m_frame.color.data(), m_frame.depth.data()));
    const size_t n = m_frame.depth.size();
    for (size_t i = 0; i < n; ++i)
      if (!(m_frame.depth[i] < 1.0f))
        for (int c = 0; c < 4; ++c) m_frame.color[4 * i + c] = m_clear_color[c];
  }
  // The same drawF() for a display loop: everything is enqueued behind the frame's passes and nothing is copied back --
  // the frame stays on the device like the reference's stays in the window (rgbdr_draw).  deviceFrame() says where
  // (filled = m_fill_holes); pixels of depth 1 keep the ray-march's / fill's values (the host applies its clear colour
  // when it presents the frame).
  void drawFOnDevice()
  {
    if (!m_have_view) throw std::invalid_argument("ReconIntegration::drawFOnDevice() before setView()");
    rgbdr_view view = m_view;
    view.skip_space = m_skip_space ? 1 : 0;
    check(m_be.ctx(), rgbdr_draw(m_be.ctx(), &view, m_fill_holes ? 1 : 0));
  }
  struct DeviceFrame {
    void* color = nullptr;  // [height][width][4] f32
    void* depth = nullptr;  // [height][width] f32
    int width = 0, height = 0;
    void* ready = nullptr;  // deviceFrameAsync: the hipEvent_t the presenter's queue waits for before it reads
  };
  DeviceFrame deviceFrame() const
  {
    DeviceFrame f;
    check(m_be.ctx(), rgbdr_device_view_frame(m_be.ctx(), m_fill_holes ? 1 : 0, &f.color, &f.depth, &f.width, &f.height));
    return f;
  }
  // for a presenter with a queue of its own (a pipelined context's hole filling runs beside the next frame): nothing waits
  DeviceFrame deviceFrameAsync() const
  {
    DeviceFrame f;
    check(m_be.ctx(), rgbdr_device_view_frame_async(m_be.ctx(), m_fill_holes ? 1 : 0, &f.color, &f.depth, &f.width, &f.height, &f.ready));
    return f;
  }
  void setClearColor(float r, float g, float b, float a)
  {
    m_clear_color[0] = r, m_clear_color[1] = g, m_clear_color[2] = b, m_clear_color[3] = a;
  }
  // resize(width, height), :494-512: the reference re-allocates its FBOs / LOD atlases for the new window; here
  // the next draw renders that many pixels (the library sizes its buffers per call)
  void resize(std::size_t width, std::size_t height) override
  {
    m_view.width = (int32_t)width;
    m_view.height = (int32_t)height;
  }
  // :538-543: the uniform shifts gl_FragCoord to viewport-local pixels (tsdf_raymarch.fs:70,396-397); frame() is
  // viewport-local already, so the offset is only kept
  void setViewportOffset(float x, float y) override
  {
    m_viewport_offset[0] = x;
    m_viewport_offset[1] = y;
  }
  Frame const& frame() const { return m_frame; }

  // ---- the same with the uniforms passed in (no state) ----
  void draw(rgbdr_view view, Frame& f) const
  {
    view.skip_space = m_skip_space ? 1 : 0;
    const size_t n = (size_t)view.width * view.height;
    f.width = view.width;
    f.height = view.height;
    f.color.resize(n * 4);
    f.depth.resize(n);
    f.num_samples.resize(n);
    check(m_be.ctx(), rgbdr_raymarch(m_be.ctx(), &view, f.color.data(), f.depth.data(), f.num_samples.data()));
  }
  void drawF(rgbdr_view const& view, Frame& f) const
  {
    draw(view, f);
    if (m_fill_holes) check(m_be.ctx(), rgbdr_fill_colors(m_be.ctx(), f.color.data(), f.depth.data()));
  }
  // the depth-peel image of drawDepthLimits (:409-429): RGBA32F per pixel
  std::vector<float> drawDepthLimits(rgbdr_view const& view) const
  {
    std::vector<float> peels((size_t)view.width * view.height * 4);
    check(m_be.ctx(), rgbdr_draw_depth_limits(m_be.ctx(), &view, peels.data()));
    return peels;
  }
  // no counterpart in the reference (consumers sample texture unit 29)
  std::vector<float> readbackTsdf(rgbdr_geometry* geo_out = nullptr) const
  {
    rgbdr_geometry g;
    check(m_be.ctx(), rgbdr_get_geometry(m_be.ctx(), &g));
    std::vector<float> v((size_t)g.res_volume[0] * g.res_volume[1] * (size_t)(g.slab_voxel_z1 - g.slab_voxel_z0));
    check(m_be.ctx(), rgbdr_readback_tsdf(m_be.ctx(), v.data()));
    if (geo_out) *geo_out = g;
    return v;
  }

 private:
  Backend& m_be;
  bool m_fill_holes = true, m_skip_space = true;  // defaults of recon_integration.cpp:60-63
  float m_clear_color[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // g_clear_color
  Frame m_frame;
  float m_viewport_offset[2] = {0.0f, 0.0f};
};

// The per-step halo exchange of a host that splits the volume into Z slabs, one context per GPU
// (the reference is single-GPU and has no counterpart).  `nccl_comm` is the host's
// ncclComm_t over the slab ranks (rank r owns slab r); the library binds RCCL at run time, so this
// header needs neither HIP nor RCCL headers.  Per frame:
//   halo.beginStep(); recon.integrate(); halo.exchangeAsync();      // transfer k overlaps frame k+1
// and halo.wait() before anything samples across slab faces (drawing).  With `loopback` both
// neighbours are `self_rank` (an inner slab exchanging with itself: tests on one GPU).
class HaloExchanger {
 public:
  HaloExchanger(Backend& be, void* nccl_comm, int slab_rank, int slab_count, bool loopback = false, int self_rank = 0)
      : m_be(be), m_comm(nccl_comm)
  {
    if (!nccl_comm) throw std::invalid_argument("null communicator");
    m_lo = loopback ? self_rank : (slab_rank > 0 ? slab_rank - 1 : -1);
    m_hi = loopback ? self_rank : (slab_rank < slab_count - 1 ? slab_rank + 1 : -1);
  }
  void beginStep() { check(m_be.ctx(), rgbdr_halo_begin_step(m_be.ctx())); }
  void exchangeAsync() { check(m_be.ctx(), rgbdr_halo_exchange_async(m_be.ctx(), m_comm, m_lo, m_hi)); }
  void wait() { check(m_be.ctx(), rgbdr_halo_wait(m_be.ctx())); }
  // TimerDatabase::duration("halo") in ms: the last transfer (timers must be enabled)
  double lastTransferMs() const
  {
    uint64_t ns = 0;
    check(m_be.ctx(), rgbdr_timer_ns(m_be.ctx(), "halo", &ns));
    return (double)ns * 1e-6;
  }

 private:
  Backend& m_be;
  void* m_comm;
  int m_lo = -1, m_hi = -1;
};

// The pre_* chain sharded by sensor over the ranks of a slab job (no counterpart in the single-GPU
// reference): rank r of `world` runs NetKinectArray::processTextures for sensors [r n / world, (r + 1) n / world) and
// gather() completes the frame on every rank -- ncclAllGather of the packed frame texels + ncclAllReduce of the brick
// counters, on the stream the chain ran on -- between processTextures() and updateOccupiedBricks().
class FrameGather {
 public:
  FrameGather(Backend& be, void* nccl_comm, int rank, int world) : m_be(be), m_comm(nccl_comm)
  {
    const int n = (int)be.num();
    if (world < 1 || n % world) throw std::invalid_argument("the sensors do not split evenly over the ranks");
    check(be.ctx(), rgbdr_set_sensor_shard(be.ctx(), rank * (n / world), n / world));
  }
  void gather() { check(m_be.ctx(), rgbdr_shard_allgather(m_be.ctx(), m_comm)); }

 private:
  Backend& m_be;
  void* m_comm;
};

// The sharded chain with its gather off the critical path: the sweep lags the chain by one frame.  A second, CHAIN-ONLY
// backend (same calibration files and bounding box, brick size = the sweeping backend's, voxel size = its brick size: one
// voxel per brick) runs NetKinectArray::processTextures for frame k+1 on the sweeping backend's stream -- hence before, not
// under, the sweep of frame k --, its gather (rgbdr_shard_allgather_async) travels on a stream of its own under that sweep,
// and the sweeping backend takes the completed frame with rgbdr_import_frame_from at the start of the next step:
//     stream   :  import(k) | chain(k+1) | sweep(k)        import(k+1) | chain(k+2) | sweep(k+1) ...
//     gather   :                 gather(k+1) ......               gather(k+2) ......
// (rgbd-recon_amd/dist.py LaggedChain is the same in Python; no counterpart in the single-GPU reference.)  After push() of
// frame k+1 the volume is frame k's; flush() sweeps the last frame.  Give it a communicator of its OWN.
class LaggedChain {
 public:
  LaggedChain(Backend& sweep, Backend& chain, void* nccl_comm, int rank, int world) : m_sweep(sweep), m_chain(chain), m_comm(nccl_comm)
  {
    const int n = (int)chain.num();
    if (world < 1 || n % world) throw std::invalid_argument("the sensors do not split evenly over the ranks");
    check(chain.ctx(), rgbdr_set_stream(chain.ctx(), rgbdr_stream(sweep.ctx())));
    check(chain.ctx(), rgbdr_set_sensor_shard(chain.ctx(), rank * (n / world), n / world));
    // the sweep in two launches: RCCL's kernel gets onto the device between them (next to ONE launch it sits until the end)
    check(sweep.ctx(), rgbdr_set_sweep_launches(sweep.ctx(), 2));
  }
  ~LaggedChain() { rgbdr_set_sweep_launches(m_sweep.ctx(), 1); }
  // `chain_nka` (a NetKinectArray of the chain backend) holds the frame just uploaded; `recon` sweeps on the other backend;
  // `halo` (may be null) is stepped around the sweep
  template <class NKA, class Recon, class Halo>
  void push(NKA& chain_nka, Recon& recon, Halo* halo)
  {
    const bool have = takeOver(recon);
    check(m_chain.ctx(), rgbdr_clear_occupied_bricks(m_chain.ctx()));
    chain_nka.processTextures();
    check(m_chain.ctx(), rgbdr_shard_allgather_async(m_chain.ctx(), m_comm));
    m_pending = true;
    if (have) sweep(recon, halo);
  }
  template <class Recon, class Halo>
  void flush(Recon& recon, Halo* halo)
  {
    if (takeOver(recon)) sweep(recon, halo);
    m_pending = false;
  }

 private:
  template <class Recon>
  bool takeOver(Recon& recon)
  {
    if (!m_pending) return false;
    recon.clearOccupiedBricks();
    check(m_sweep.ctx(), rgbdr_import_frame_from(m_sweep.ctx(), m_chain.ctx()));
    return true;
  }
  template <class Recon, class Halo>
  void sweep(Recon& recon, Halo* halo)
  {
    recon.updateOccupiedBricks();
    if (halo) halo->beginStep();
    recon.integrate();
    if (halo) halo->exchangeAsync();
  }
  Backend& m_sweep;
  Backend& m_chain;
  void* m_comm;
  bool m_pending = false;
};

// process_textures() of source/kinect_client.cpp:572-580
inline void process_textures(NetKinectArray& nka, ReconIntegration& recon, FrameGather* shard = nullptr)
{
  recon.clearOccupiedBricks();
  nka.processTextures();
  if (shard) shard->gather();  // multi-GPU hosts with a sharded chain: the other ranks' sensors arrive here
  recon.updateOccupiedBricks();
}

}  // namespace host
}  // namespace rgbdr
