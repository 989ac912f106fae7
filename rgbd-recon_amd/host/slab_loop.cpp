// slab_loop.cpp -- the frame loop of a multi-GPU C++ host: one process per GPU, each owning one
// Z slab of the volume, the per-step halo exchange over RCCL through the C ABI
// (rgbdr_halo_begin_step / rgbdr_halo_exchange_async / rgbdr_halo_wait via host::HaloExchanger).
// The frame sequence is source/kinect_client.cpp:572-602 per rank; the exchange of step k
// overlaps step k+1 and no host synchronisation happens between frames.
//
//   slab_loop <dir> <num_sensors> <W> <H> <G> <frames> <out.bin> --loopback
//       one GPU: this process is slab 1 of 4 (an inner slab) and both of its neighbours (RCCL
//       accepts a send / recv pair to the own rank) -- tests/test_host_cpp.py
//   ... --lag   (either form) the sharded chain runs one frame AHEAD of the sweep on a chain-only backend, its gather under
//       the sweep (host::LaggedChain: rgbdr_shard_allgather_async + rgbdr_import_frame_from)
//   slab_loop <dir> <num_sensors> <W> <H> <G> <frames> <out.bin> --rank r --world n --id <file>
//       n processes (GPUs 0..n-1 of one node): rank 0 writes the ncclUniqueId to <file>
// <dir> as for frame_loop (s<i>.yml / .cv_xyz / .cv_uv / .cv_xyz_inv, recordings/s<i>.stream with
// at least <frames> frames).  <out.bin>: halo_tile_layers h, layer floats, then the tile layers
// [lower halo | first h owned | last h owned | upper halo] after the last frame, then the "halo"
// timer in ms.
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <memory>
#include <string>
#include <thread>

#include "rgbdr_host.hpp"

using namespace rgbdr::host;

#define NCCLCHK(expr)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

int main(int argc, char** argv)
{
  if (argc < 9) {
    std::fprintf(stderr, "usage: %s <dir> <num_sensors> <W> <H> <G> <frames> <out.bin> --loopback | --rank r --world n --id <file>\n", argv[0]);
    return 2;
  }
  try {
    const std::string dir = std::string(argv[1]) + "/";
    const int n = std::atoi(argv[2]), G = std::atoi(argv[5]), frames = std::atoi(argv[6]);
    bool loopback = false, lag = false;
    int rank = 0, world = 1;
    std::string id_file;
    for (int i = 8; i < argc; ++i) {
      const std::string a = argv[i];
      if (a == "--loopback") loopback = true;
      else if (a == "--lag") lag = true;
      else if (a == "--rank" && i + 1 < argc) rank = std::atoi(argv[++i]);
      else if (a == "--world" && i + 1 < argc) world = std::atoi(argv[++i]);
      else if (a == "--id" && i + 1 < argc) id_file = argv[++i];
    }
    const int device = loopback ? 0 : rank;
    if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    // Two communicators over the slab ranks: one for the halo exchange, one for the frame gather.  Operations on ONE
    // communicator execute in issue order whatever their streams (include/rgbdr.h, rgbdr_shard_allgather), so on a shared one
    // the gather of frame k+1 would queue behind the face transfer of frame k and the overlap would be lost.
    ncclUniqueId ids[2];
    if (loopback || rank == 0) {
      NCCLCHK(ncclGetUniqueId(&ids[0]));
      NCCLCHK(ncclGetUniqueId(&ids[1]));
      if (!loopback) {
        FILE* f = std::fopen((id_file + ".tmp").c_str(), "wb");
        if (!f || std::fwrite(ids, sizeof(ids[0]), 2, f) != 2) throw std::runtime_error("cannot write " + id_file);
        std::fclose(f);
        std::rename((id_file + ".tmp").c_str(), id_file.c_str());
      }
    } else {
      FILE* f = nullptr;
      for (int tries = 0; tries < 600 && !(f = std::fopen(id_file.c_str(), "rb")); ++tries)
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      if (!f || std::fread(ids, sizeof(ids[0]), 2, f) != 2) throw std::runtime_error("cannot read " + id_file);
      std::fclose(f);
    }
    ncclComm_t comm, comm_gather = nullptr;
    NCCLCHK(ncclCommInitRank(&comm, loopback ? 1 : world, ids[0], loopback ? 0 : rank));
    // The pre_* chain is sharded by sensor only when the sensors split evenly over the ranks; otherwise (one sensor, two
    // sensors on four ranks ...) every rank runs the whole chain, as a single-GPU host does.
    const int gather_world = loopback ? 1 : world;
    const bool sharded = n > 1 && n % gather_world == 0;
    if (sharded) NCCLCHK(ncclCommInitRank(&comm_gather, gather_world, ids[1], loopback ? 0 : rank));

    CalibrationFiles cf;
    cf.width = cf.widthC = (unsigned)std::atoi(argv[3]);
    cf.height = cf.heightC = (unsigned)std::atoi(argv[4]);
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
    const int slab_rank = loopback ? 1 : rank, slab_count = loopback ? 4 : world;
    Backend be(cf, bbox, 0.01f, voxel, 8.0f * voxel, device, slab_rank, slab_count);
    CalibVolumes cv(be, cf.filenames);
    cv.loadInverseCalibs(dir);
    NetKinectArray nka(&cf, &cv);
    ReconIntegration recon(cf, &cv, bbox, 0.01f, voxel);
    recon.setUseBricks(false);  // full sweep: the kernel stores its boundary layers into the staging set itself
    HaloExchanger halo(be, comm, slab_rank, slab_count, loopback, 0);
    // the pre_* chain sharded by sensor over the ranks of the communicator (every rank runs n / world sensors and the
    // packed frames are all-gathered); with --loopback the communicator has one rank, which holds every sensor: the
    // collectives still run (all-gather and all-reduce of one rank) and must leave the frame as it is
    std::unique_ptr<FrameGather> shard;
    // --lag: a chain-only backend with the same brick grid (one voxel per brick) and its own NetKinectArray
    std::unique_ptr<Backend> be_chain;
    std::unique_ptr<CalibVolumes> cv_chain;
    std::unique_ptr<NetKinectArray> nka_chain;
    std::unique_ptr<LaggedChain> lagged;
    if (lag && sharded) {
      be_chain.reset(new Backend(cf, bbox, 0.01f, 8.0f * voxel, 8.0f * voxel, device));
      cv_chain.reset(new CalibVolumes(*be_chain, cf.filenames));
      nka_chain.reset(new NetKinectArray(&cf, cv_chain.get()));
      lagged.reset(new LaggedChain(be, *be_chain, comm_gather, loopback ? 0 : rank, gather_world));
    } else if (sharded) {
      shard.reset(new FrameGather(be, comm_gather, loopback ? 0 : rank, gather_world));
    }
    check(be.ctx(), rgbdr_enable_timers(be.ctx(), 1));
    const size_t colorsize = (size_t)cf.widthC * cf.heightC * 3, depthsize = (size_t)cf.width * cf.height * 4;
    for (int k = 0; k < frames; ++k) {  // no host synchronisation between frames
      if (lagged) {
        nka_chain->readFromFiles(streams, colorsize, depthsize, (size_t)k);
        lagged->push(*nka_chain, recon, &halo);  // the chain of frame k, the sweep of frame k - 1
        continue;
      }
      nka.readFromFiles(streams, colorsize, depthsize, (size_t)k);
      process_textures(nka, recon, shard.get());
      halo.beginStep();
      recon.integrate();
      halo.exchangeAsync();
    }
    if (lagged) lagged->flush(recon, &halo);
    halo.wait();
    check(be.ctx(), rgbdr_sync(be.ctx()));
    rgbdr_geometry g;
    check(be.ctx(), rgbdr_get_geometry(be.ctx(), &g));
    rgbdr_tsdf_device_view view;
    check(be.ctx(), rgbdr_device_tsdf(be.ctx(), &view));
    const int h = view.halo_layers, owned = view.owned_layers;
    const size_t layer = view.layer_bytes / sizeof(float);
    std::vector<float> out(layer * (size_t)h * 4);
    const int first[4] = {0, h, h + owned - h, h + owned};
    for (int i = 0; i < 4; ++i)
      check(be.ctx(), rgbdr_readback_tile_layers(be.ctx(), first[i], h, out.data() + layer * (size_t)h * i));
    const double ms = halo.lastTransferMs();
    FILE* f = std::fopen(argv[7], "wb");
    if (!f) return 3;
    const int32_t hdr[2] = {h, (int32_t)layer};
    std::fwrite(hdr, sizeof(int32_t), 2, f);
    std::fwrite(out.data(), sizeof(float), out.size(), f);
    std::fwrite(&ms, sizeof(double), 1, f);
    std::fclose(f);
    std::printf("slab %d of %d: tile layers [%d, %d), halo %d layers, last transfer %.4f ms\n", slab_rank, slab_count,
                g.slab_tile_z0, g.slab_tile_z1, h, ms);
    if (comm_gather) NCCLCHK(ncclCommDestroy(comm_gather));
    NCCLCHK(ncclCommDestroy(comm));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "slab_loop: %s\n", e.what());
    return 1;
  }
  return 0;
}
