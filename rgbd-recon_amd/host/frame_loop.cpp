// frame_loop.cpp -- the per-frame sequence of source/kinect_client.cpp:572-602
// (update -> process_textures -> integrate) written against the C++ host mirror,
// with calibration volumes and a recorded ".stream" frame in the reference's
// on-disk formats.  Used by tests/test_host_cpp.py as the C++-side drop-in check;
// also the smallest example of how an application calls the backend.
//
//   frame_loop <dir> <num_sensors> <W> <H> <G> <out.tsdf>
// expects <dir>/s<i>.yml names (only used to derive s<i>.cv_xyz / .cv_uv),
// <dir>/s<i>.cv_xyz_inv and <dir>/recordings/s<i>.stream.
//   frame_loop --ks <scene.ks> <voxel_size> <out.tsdf>
// reads the scene the way kinect_client does: `.ks` (kinect / bbx lines), the sensor
// ymls (sizes, near_far, compress_rgb / compress_depth), <yml base>.cv_xyz / .cv_uv,
// <ks dir>/<base>.cv_xyz_inv and recordings/<base>.stream relative to the working
// directory (NetKinectArray.cpp:727-731).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include <hip/hip_runtime_api.h>

#include "rgbdr_host.hpp"

using namespace rgbdr::host;

// A second drawing mode next to ReconIntegration, with the constructor triple of the reference's other modes --
// ReconTrigrid(CalibrationFiles const&, CalibVolumes const*, gloost::BoundingBox const&), recon_trigrid.hpp:16 -- that
// consumes what ReconTrigrid consumes: the frame's colour, depth_b, quality and normal images of every sensor
// (recon_trigrid.cpp:30-33: kinect_colors 1, kinect_depths 2, kinect_qualities 3, kinect_normals 4).  Its arithmetic
// (screen-space triangle meshes through the rasteriser) is outside this backend's scope; what it shows is the input
// side of such a mode over the C ABI: the images are read where they live on the device (rgbdr_device_image:
// zero-copy views), stream-ordered behind the passes that write them, with no host copy made by the library.
// draw() leaves a 64-bit FNV-1a digest of every image it consumed (the test compares them with the library's own
// readbacks) -- the stand-in for "the mode drew something that depends on every texel".
class ReconFrameConsumer : public Reconstruction {
 public:
  ReconFrameConsumer(CalibrationFiles const& cfs, CalibVolumes const* cv, BoundingBox const& bbox) : Reconstruction(cfs, cv, bbox) {}
  void draw() override
  {
    static const char* units[4] = {"color", "depth", "quality", "normal"};
    m_digests.assign((size_t)m_num_kinects * 4, 0);
    for (unsigned i = 0; i < m_num_kinects; ++i)
      for (int u = 0; u < 4; ++u) {
        rgbdr_image_device_view v = frameImage(units[u], i);
        const size_t bytes = (size_t)v.width * v.height * v.channels * v.element_bytes;
        std::vector<unsigned char> host(bytes);
        // the consumer's own device work: here a copy straight out of the context's image, ordered on the stream the
        // passes ran on (a renderer would launch its kernels on that stream, or wait for an event recorded on it)
        if (hipMemcpyAsync(host.data(), v.ptr, bytes, hipMemcpyDeviceToHost, (hipStream_t)v.stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)v.stream) != hipSuccess)
          throw std::runtime_error("ReconFrameConsumer: reading the device image failed");
        uint64_t h = 1469598103934665603ull;
        for (unsigned char b : host) h = (h ^ b) * 1099511628211ull;
        m_digests[(size_t)i * 4 + u] = h;
      }
  }
  std::vector<uint64_t> const& digests() const { return m_digests; }
  // The calibration volumes the reference's other modes sample on CalibVolumes' texture units (recon_calibs.cpp:35-36,
  // trigrid.vs), read where they live on the device: digests of the cv_xyz records (x, y, z: the 12-byte records of the
  // file, without the padding lane of the device layout) and of the cv_uv records of every sensor.
  std::vector<uint64_t> calibrationDigests() const
  {
    std::vector<uint64_t> out;
    for (unsigned i = 0; i < m_num_kinects; ++i) {
      const rgbdr_calibration_device_view v = m_cv->deviceVolumes(i);
      const size_t nx = (size_t)v.xyz_res[0] * v.xyz_res[1] * v.xyz_res[2], nu = (size_t)v.uv_res[0] * v.uv_res[1] * v.uv_res[2];
      std::vector<float> xyz(nx * 4), uv(nu * 2);
      if (hipMemcpyAsync(xyz.data(), v.cv_xyz, xyz.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)v.stream) != hipSuccess ||
          hipMemcpyAsync(uv.data(), v.cv_uv, uv.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)v.stream) != hipSuccess ||
          hipStreamSynchronize((hipStream_t)v.stream) != hipSuccess)
        throw std::runtime_error("ReconFrameConsumer: reading the calibration volumes failed");
      uint64_t h = 1469598103934665603ull;
      for (size_t c = 0; c < nx; ++c) {
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&xyz[c * 4]);
        for (int k = 0; k < 12; ++k) h = (h ^ b[k]) * 1099511628211ull;
      }
      out.push_back(h);
      h = 1469598103934665603ull;
      const unsigned char* b = reinterpret_cast<const unsigned char*>(uv.data());
      for (size_t k = 0; k < uv.size() * 4; ++k) h = (h ^ b[k]) * 1099511628211ull;
      out.push_back(h);
    }
    return out;
  }

 private:
  std::vector<uint64_t> m_digests;
};

static int run_ks(const char* ks_path, float voxel, const char* out_path)
{
  KsFile ks = parseKs(ks_path);
  CalibrationFiles cf = parseCalibrationFiles(ks.calib_filenames);
  Backend be(cf, ks.bbox, 0.01f, voxel, 0.1f);
  CalibVolumes cv(be, cf.filenames);
  cv.loadInverseCalibs(ks.resource_path);
  NetKinectArray nka(be);
  ReconIntegration recon(be);
  std::vector<std::string> streams;
  for (auto const& yml : cf.filenames) {
    std::string base = yml.substr(yml.find_last_of("/\\") + 1);
    base = base.substr(0, base.size() - 4);
    streams.push_back("recordings/" + base + ".stream");
  }
  nka.readFromFiles(streams, colorFrameBytes(cf), depthFrameBytes(cf), 0);
  process_textures(nka, recon);
  recon.integrate();
  rgbdr_geometry g;
  std::vector<float> tsdf = recon.readbackTsdf(&g);
  FILE* f = std::fopen(out_path, "wb");
  if (!f) return 3;
  std::fwrite(tsdf.data(), sizeof(float), tsdf.size(), f);
  std::fclose(f);
  std::printf("sensors %u depth %ux%u color %ux%u compress_rgb %d compress_depth %d res %d %d %d bricks %u occupied %.4f\n",
              cf.num(), cf.width, cf.height, cf.widthC, cf.heightC, cf.compressedRGB, (int)cf.compressedDepth,
              g.res_volume[0], g.res_volume[1], g.res_volume[2], recon.numBricks(), recon.occupiedRatio());
  return 0;
}

// frame_loop --parse <sensor.yml>...            prints what parseCalibrationFiles read
// frame_loop --stream <file> <colorsize> <depthsize> <index> <out>   writes frame `index` (colour, depth) to <out>
static int run_tools(int argc, char** argv)
{
  const std::string mode = argv[1];
  if (mode == "--parse") {
    std::vector<std::string> names(argv + 2, argv + argc);
    CalibrationFiles cf = parseCalibrationFiles(names);
    std::printf("%u %u %u %u %d %d", cf.getWidth(), cf.getHeight(), cf.getWidthC(), cf.getHeightC(), cf.isCompressedRGB(),
                cf.isCompressedDepth() ? 1 : 0);
    for (unsigned i = 0; i < cf.num(); ++i) std::printf(" %.9g %.9g", cf.near_[i], cf.far_[i]);
    std::printf("\n");
    return 0;
  }
  if (mode == "--stream" && argc == 7) {
    const size_t cs = (size_t)std::atoll(argv[3]), ds = (size_t)std::atoll(argv[4]);
    std::vector<unsigned char> buf(cs + ds);
    readStreamFrame(argv[2], cs, ds, (size_t)std::atoll(argv[5]), buf.data(), buf.data() + cs);
    FILE* f = std::fopen(argv[6], "wb");
    if (!f) return 3;
    std::fwrite(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return 0;
  }
  return 2;
}

int main(int argc, char** argv)
{
  if (argc >= 3 && (std::string(argv[1]) == "--parse" || std::string(argv[1]) == "--stream")) {
    try {
      return run_tools(argc, argv);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "frame_loop: %s\n", e.what());
      return 1;
    }
  }
  if (argc == 5 && std::string(argv[1]) == "--ks") {
    try {
      return run_ks(argv[2], (float)std::atof(argv[3]), argv[4]);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "frame_loop: %s\n", e.what());
      return 1;
    }
  }
  if (argc != 7 && argc != 8) {
    std::fprintf(stderr, "usage: %s <dir> <num_sensors> <W> <H> <G> <out.tsdf> [view.bin]\n", argv[0]);
    return 2;
  }
  try {
    const std::string dir = std::string(argv[1]) + "/";
    CalibrationFiles cf;
    const int n = std::atoi(argv[2]);
    cf.width = cf.widthC = (unsigned)std::atoi(argv[3]);
    cf.height = cf.heightC = (unsigned)std::atoi(argv[4]);
    const int G = std::atoi(argv[5]);
    std::vector<std::string> streams;
    for (int i = 0; i < n; ++i) {
      cf.filenames.push_back(dir + "s" + std::to_string(i) + ".yml");
      streams.push_back(dir + "recordings/s" + std::to_string(i) + ".stream");
    }
    cf.near_.assign(n, 0.5f);
    cf.far_.assign(n, 4.5f);
    BoundingBox bbox;
    bbox.pmax = {{1.0f, 2.0f, 1.0f}};
    const float voxel = 2.0f / (float)G;
    Backend be(cf, bbox, 0.01f, voxel, 8.0f * voxel);
    CalibVolumes cv(be, cf.filenames);
    cv.loadInverseCalibs(dir);
    // kinect_client.cpp:240-255: the application's globals -- g_nka, g_recon_integration and the list of drawing
    // modes it indexes with g_recon_mode, all through the reference's constructor signatures
    NetKinectArray nka(&cf, &cv);
    std::shared_ptr<ReconIntegration> recon_integration = std::make_shared<ReconIntegration>(cf, &cv, bbox, 0.01f, voxel);
    std::vector<std::shared_ptr<Reconstruction>> recons;
    recons.emplace_back(recon_integration);
    recons.emplace_back(std::make_shared<ReconFrameConsumer>(cf, &cv, bbox));   // like g_recons' ReconTrigrid, kinect_client.cpp:251-255
    const unsigned recon_mode = 0;
    ReconIntegration& recon = *recon_integration;
    const size_t colorsize = (size_t)cf.widthC * cf.heightC * 3, depthsize = (size_t)cf.width * cf.height * 4;
    if (const char* msg = std::getenv("RGBDR_MESSAGE_FILE")) {  // one server message instead of the recordings
      std::vector<unsigned char> buf((colorsize + depthsize) * (size_t)n);
      FILE* mf = std::fopen(msg, "rb");
      if (!mf || std::fread(buf.data(), 1, buf.size(), mf) != buf.size()) return 6;
      std::fclose(mf);
      nka.updateFromMessage(buf.data(), buf.size(), (unsigned)n);
      std::fprintf(stderr, "frame time %.17g\n", nka.getCurrentFrameTime());
    } else {
      nka.readFromFiles(streams, colorsize, depthsize, 0);
    }
    process_textures(nka, recon);
    recon.integrate();
    rgbdr_geometry g;
    std::vector<float> tsdf = recon.readbackTsdf(&g);
    FILE* f = std::fopen(argv[6], "wb");
    if (!f) return 3;
    std::fwrite(tsdf.data(), sizeof(float), tsdf.size(), f);
    std::fclose(f);
    std::printf("res %d %d %d bricks %u occupied %.4f\n", g.res_volume[0], g.res_volume[1], g.res_volume[2],
                recon.numBricks(), recon.occupiedRatio());
    {  // the other drawing mode, through the base pointer like g_recons.at(g_recon_mode)->drawF() (kinect_client.cpp:617):
       // it consumes the frame's images zero-copy; NetKinectArray's own accessors must hand out the same memory
      Reconstruction& other = *recons.at(1);
      other.drawF();
      auto const& dg = static_cast<ReconFrameConsumer&>(other).digests();
      std::printf("frame images");
      for (uint64_t h : dg) std::printf(" %016llx", (unsigned long long)h);
      std::printf("\n");
      std::printf("calibration volumes");
      for (uint64_t h : static_cast<ReconFrameConsumer&>(other).calibrationDigests()) std::printf(" %016llx", (unsigned long long)h);
      const std::array<uint32_t, 3> vr = cv.getVolumeRes();        // the inverse volume's, like the reference's
      const std::array<float, 2> dl = cv.getDepthLimits(0);
      std::printf(" inv_res %u %u %u limits %.9g %.9g\n", vr[0], vr[1], vr[2], dl[0], dl[1]);
      rgbdr_image_device_view a = nka.deviceImage("quality", 0), b = nka.deviceImage(RGBDR_IMG_QUALITY, 0);
      if (a.ptr != b.ptr || !a.ptr || a.width != (int)cf.width || a.height != (int)cf.height || a.channels != 1) return 8;
      if (nka.readbackImage(RGBDR_IMG_DEPTH_B_RG, 0).size() != (size_t)cf.width * cf.height * 2) return 8;
      try {
        nka.deviceImage("bg", 0);          // the reference's map throws std::out_of_range for an unknown unit name
        return 8;
      } catch (const std::out_of_range&) {
      }
    }
    if (argc == 8) {  // g_recons[mode]->drawF() of kinect_client.cpp: ray-march + hole filling for the given uniforms
      rgbdr_view view;
      FILE* vf = std::fopen(argv[7], "rb");
      if (!vf || std::fread(&view, sizeof(view), 1, vf) != 1) return 5;
      std::fclose(vf);
      // draw3d(), kinect_client.cpp:610-617: the camera is set, then g_recons.at(g_recon_mode)->drawF()
      Reconstruction& mode = *recons.at(recon_mode);
      mode.resize((std::size_t)view.width, (std::size_t)view.height);   // kinect_client.cpp:1003-1007
      mode.setView(view);
      mode.setViewportOffset(0.0f, 0.0f);
      mode.drawF();
      ReconIntegration::Frame const& frame = recon_integration->frame();
      FILE* ff = std::fopen((std::string(argv[6]) + ".frame").c_str(), "wb");
      if (!ff) return 3;
      std::fwrite(frame.color.data(), sizeof(float), frame.color.size(), ff);
      std::fwrite(frame.depth.data(), sizeof(float), frame.depth.size(), ff);
      std::fclose(ff);
      // the anaglyph path of draw3d() (kinect_client.cpp:620-640): without hole filling the left eye is drawn with
      // glColorMask(red) and the right eye with glColorMask(green, blue) into the same framebuffer
      recon_integration->setColorFilling(false);
      mode.setColorMaskMode(0);
      mode.drawF();                                   // what a plain draw leaves
      const std::vector<float> plain = recon_integration->frame().color;
      view.shade_mode = 2;                            // another image (normals) through the masks
      mode.setView(view);
      mode.setColorMaskMode(1);
      mode.drawF();
      const std::vector<float> red = recon_integration->frame().color;
      mode.setColorMaskMode(2);
      mode.drawF();
      const std::vector<float> both = recon_integration->frame().color;
      mode.setColorMaskMode(0);
      mode.drawF();
      const std::vector<float> normals = recon_integration->frame().color;
      size_t wrong = 0, hit = 0;
      const std::vector<float>& depth = recon_integration->frame().depth;
      for (size_t i = 0; i < depth.size(); ++i) {
        const bool h = depth[i] < 1.0f;
        hit += h;
        for (int c = 0; c < 4; ++c) {
          const float want_red = (h && c == 0) ? normals[4 * i + c] : plain[4 * i + c];       // only red of hit pixels changed
          const float want_both = (h && c < 3) ? normals[4 * i + c] : plain[4 * i + c];       // then green and blue; alpha never
          wrong += std::memcmp(&red[4 * i + c], &want_red, 4) != 0;
          wrong += std::memcmp(&both[4 * i + c], &want_both, 4) != 0;
        }
      }
      std::printf("color masks: %zu pixels hit, %zu channel values wrong\n", hit, wrong);
      if (wrong) return 7;
    }
    // error behaviour mirrors the reference's exception types
    try {
      recon.setVoxelSize(-1.0f);
      return 4;
    } catch (const std::invalid_argument&) {
    }
    if (const char* csv = std::getenv("RGBDR_TIMER_CSV")) {
      // The timing side of the application: TimerDatabase::instance().duration(name) every frame for the GUI
      // (kinect_client.cpp:431-481), writeMean / writeMin / writeMax when it quits (:835-851).  Four more frames of the same
      // loop with the timers on; with a view also the draw of every frame (depth limits, ray-march, hole filling).
      TimerDatabase timers(be);
      recon_integration->setColorFilling(true);
      recon_integration->setSpaceSkip(true);
      rgbdr_view view{};
      bool have_view = false;
      if (argc == 8) {
        FILE* vf = std::fopen(argv[7], "rb");
        have_view = vf && std::fread(&view, sizeof(view), 1, vf) == 1;
        if (vf) std::fclose(vf);
      }
      for (int frame_no = 0; frame_no < 4; ++frame_no) {
        nka.readFromFiles(streams, colorsize, depthsize, 0);
        process_textures(nka, recon);
        recon.integrate();
        if (have_view) {
          recons.at(recon_mode)->setView(view);
          recons.at(recon_mode)->drawF();
        }
        timers.sample();
      }
      std::printf("timers");
      for (const char* n : {"1preprocess", "2integrate", "3recon", "bilateral", "boundary", "brickdraw", "draw", "holefill", "morph", "normal", "quality"})
        std::printf(" %s %.0f %.4f %.0f", n, timers.duration(n), timers.mean(n), timers.getNum(n));
      std::printf("\n");
      timers.writeMean(csv);
      timers.writeMin(csv);
      timers.writeMax(csv);
      try {
        timers.duration("no such timer");
        return 9;
      } catch (const std::out_of_range&) {
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "frame_loop: %s\n", e.what());
    return 1;
  }
  return 0;
}
