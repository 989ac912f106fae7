"""Z-slab sharding across ranks: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) or gloo on CPU.

The TSDF volume is split along Z into slabs of whole storage-tile layers
(rgbdr_slab_range).  Integration needs no communication: every voxel is a pure
function of its own position (glsl/tsdf_integration.vs:23-59).  The one exchange
per step is the boundary tile layer each slab sends to its Z neighbours, so that
a consumer sampling the volume with a LINEAR filter / central differences
(glsl/tsdf_raymarch.fs:144-157) sees no seam.  In the tile-linear layout a tile
layer is one contiguous range of memory, so the exchange is two point-to-point
messages per neighbour with no packing -- over xGMI each uses one direct link.
"""
import torch
import torch.distributed as dist


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can
    wrap library-owned HBM without a copy."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2}


def wrap_device_floats(ptr, nfloats, device):
    return torch.as_tensor(_DevicePtr(ptr, nfloats), device=device)


def halo_views(view, device):
    """(send_lo, send_hi, recv_lo, recv_hi) float tensors aliasing the h = halo_layers
    boundary tile layers of a rgbdr_tsdf_device_view: the h lowest / highest owned
    layers and the h-layer halos below / above them (each one contiguous range)."""
    h = view.halo_layers
    n = h * view.layer_bytes // 4
    base, owned = int(view.base), int(view.owned)
    last = owned + (view.owned_layers - h) * view.layer_bytes
    hi_halo = owned + view.owned_layers * view.layer_bytes
    return (wrap_device_floats(owned, n, device), wrap_device_floats(last, n, device),
            wrap_device_floats(base, n, device), wrap_device_floats(hi_halo, n, device))


def exchange_halo(send_lo, send_hi, recv_lo, recv_hi, rank=None, world=None, group=None, loopback=False):
    """Neighbour exchange of one tile layer per slab face.  Slab r's lowest layer
    goes to r-1's upper halo, its highest layer to r+1's lower halo.  The same
    function runs on CUDA tensors (RCCL) and CPU tensors (gloo).
    `loopback` (tests on a single GPU): both neighbours are this process itself -- the
    lower face lands in the own upper halo and the upper face in the own lower halo,
    through the same send/recv pairs."""
    if loopback:
        me = dist.get_rank(group)
        if rank is not None and world is not None and rank == 0:            # stands in for the lowest slab: one face
            ops = [dist.P2POp(dist.isend, send_hi, me, group), dist.P2POp(dist.irecv, recv_hi, me, group)]
        elif rank is not None and world is not None and rank == world - 1:  # ... the highest slab
            ops = [dist.P2POp(dist.isend, send_lo, me, group), dist.P2POp(dist.irecv, recv_lo, me, group)]
        else:
            ops = [dist.P2POp(dist.isend, send_lo, me, group), dist.P2POp(dist.irecv, recv_hi, me, group),
                   dist.P2POp(dist.isend, send_hi, me, group), dist.P2POp(dist.irecv, recv_lo, me, group)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        return
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, send_lo, rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_lo, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, send_hi, rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_hi, rank + 1, group))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()


class _DeviceWords:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def wrap_device_words(ptr, nwords, device):
    """int32 tensor over library-owned HBM (packed frame texels, u32 brick counters: summing u32 as i32 is the same bits)"""
    return torch.as_tensor(_DeviceWords(ptr, nwords), device=device)


class FrameGather:
    """The pre_* chain sharded by sensor over the ranks of a slab job (rgbdr_set_sensor_shard): rank r runs the five
    passes for sensors [r n / k, (r + 1) n / k) and this completes the frame on every rank -- one all-gather of the
    packed 8-byte frame texels (what the sweep and the slab ray-march read of a sensor) and one all-reduce(sum) of the
    u32 brick counters (each rank counted its own sensors' pixels), on the stream the chain ran on: with
    RGBDR_FLAG_PIPELINE that is the library's second stream, i.e. next to the sweep of the frame before.

        ctx.update_device(...); ctx.clear_occupied_bricks(); ctx.process_textures()
        gather()                      # instead of nothing
        ctx.update_occupied_bricks(); ctx.integrate()

    nccl: torch.distributed collectives on the library's own buffers (zero copy), enqueued on the chain's stream.
    gloo (`via_host`, several ranks on one GPU for tests): host-staged with synchronisation.
    `loopback` (one GPU standing in for rank r of k): the other ranks' sensors are not recomputed; the traffic of the
    gather is reproduced by sending the foreign layers to this process itself into a scratch buffer (same bytes, same
    RCCL kernels, same stream ordering), the counters are all-reduced in a one-rank group."""

    def __init__(self, ctx, device, rank=None, world=None, group=None, via_host=False, loopback=False):
        self.ctx, self.device, self.group, self.via_host, self.loopback = ctx, device, group, via_host, loopback
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        n = ctx.cfg.num_sensors
        if n % self.world:
            raise ValueError("%d sensors do not split evenly over %d ranks" % (n, self.world))
        self.count = n // self.world
        self.first = self.rank * self.count
        self.all_counts = None
        if loopback:
            # the frame the context processed last (unsharded, every sensor): its brick counts stand for "the sum over
            # the ranks" in every later frame of the same static scene
            v = ctx.shard_view()
            torch.cuda.synchronize()
            self.all_counts = wrap_device_words(v.counters, v.num_bricks, device).clone()
        ctx.set_sensor_shard(self.first, self.count)
        self.scratch = None
        self.foreign_counts = None
        self.timing = None           # (start, end) events of the last gather on the chain's stream

    def __call__(self, stream=None):
        """`stream` (a torch stream): run the collectives there instead of on the chain's stream (LaggedChain)"""
        v = self.ctx.shard_view()
        words = v.sensor_bytes // 4
        n = v.num_sensors
        frames = wrap_device_words(v.frames, words * n, self.device)
        counters = wrap_device_words(v.counters, v.num_bricks, self.device)
        mine = frames[self.first * words:(self.first + self.count) * words]
        if self.via_host:                                   # gloo: through host memory, stream order by synchronisation
            self.ctx.sync()
            torch.cuda.synchronize()
            h = mine.cpu()
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            c = counters.cpu()
            dist.all_reduce(c, group=self.group)
            frames.copy_(torch.cat(parts).to(self.device))
            counters.copy_(c.to(self.device))
            torch.cuda.synchronize()
            self.ctx.shard_gather_done()
            return
        st = stream if stream is not None else torch.cuda.ExternalStream(int(v.stream), device=self.device)
        with torch.cuda.stream(st):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(st)
            self._collectives(frames, counters, mine, words, n)
            t1.record(st)
            self.timing = (t0, t1)
        self.ctx.shard_gather_done()

    def last_ms(self):
        """duration of the last gather on the chain's stream (synchronises on its end)"""
        if self.timing is None:
            return None
        self.timing[1].synchronize()
        return float(self.timing[0].elapsed_time(self.timing[1]))

    def _collectives(self, frames, counters, mine, words, n):
        if not self.loopback:
            dist.all_gather_into_tensor(frames, mine, group=self.group)
            dist.all_reduce(counters, group=self.group)
            return
        # one GPU standing in for a rank: the other ranks' layers travel to this process itself (same bytes, same RCCL
        # kernels, same stream ordering); their brick counts -- what the unsharded frame counted beyond this shard of the
        # same static scene -- are added back after the one-rank all-reduce
        lo, hi = frames[:self.first * words], frames[(self.first + self.count) * words:]
        if self.scratch is None:
            self.scratch = torch.empty(words * (n - self.count), dtype=torch.int32, device=self.device)
        me = dist.get_rank(self.group)
        ops = []
        if lo.numel():
            ops += [dist.P2POp(dist.isend, lo, me, self.group), dist.P2POp(dist.irecv, self.scratch[:lo.numel()], me, self.group)]
        if hi.numel():
            ops += [dist.P2POp(dist.isend, hi, me, self.group), dist.P2POp(dist.irecv, self.scratch[lo.numel():], me, self.group)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if self.foreign_counts is None:
            self.foreign_counts = self.all_counts - counters
        dist.all_reduce(counters, group=self.group)
        counters.add_(self.foreign_counts)


class RcclComm:
    """A raw RCCL communicator through ctypes, for hosts that let the LIBRARY enqueue the collectives on its own streams
    (rgbdr_halo_exchange_async, rgbdr_shard_allgather: the C ABI a C++ host uses, host/slab_loop.cpp) instead of going
    through torch.distributed's process group, whose collectives run on a stream of its own.  The unique id travels
    through the torch.distributed group that exists anyway (any backend)."""
    _lib = None
    _how = None

    @classmethod
    def lib(cls):
        """ONE RCCL per process: the copy that is mapped already (torch links its own librccl.so, soname librccl.so.1) is
        bound with RTLD_NOLOAD; only a process without any loads the system's.  A second copy next to torch's would have
        its own bootstrap state, and a rendezvous between two different RCCL builds blocks instead of failing.  The C
        library binds the same way (csrc/api_halo.cpp load_rccl)."""
        if cls._lib is None:
            import ctypes as C
            import os
            RTLD_NOLOAD = 4
            names = [n for n in (os.environ.get("RGBDR_RCCL_LIB"), "librccl.so.1", "librccl.so") if n]
            for name in names:
                try:
                    cls._lib, cls._how = C.CDLL(name, mode=C.RTLD_GLOBAL | RTLD_NOLOAD), "already mapped (RTLD_NOLOAD %s)" % name
                    break
                except OSError:
                    pass
            if cls._lib is None:
                for name in names:
                    try:
                        # RTLD_GLOBAL: the C library then finds this copy with its own RTLD_NOLOAD
                        cls._lib, cls._how = C.CDLL(name, mode=C.RTLD_GLOBAL), "loaded here (%s)" % name
                        break
                    except OSError:
                        pass
            if cls._lib is None:
                raise OSError("no librccl.so.1 in this process or on the library path (set RGBDR_RCCL_LIB)")
        return cls._lib

    @classmethod
    def library_info(cls):
        """{"path": file of the bound RCCL, "version": ncclGetVersion, "bound": how, "copies_mapped": RCCL files in this process}"""
        import ctypes as C
        L = cls.lib()
        v = C.c_int(0)
        version = int(v.value) if L.ncclGetVersion(C.byref(v)) else int(v.value)

        class DlInfo(C.Structure):
            _fields_ = [("dli_fname", C.c_char_p), ("dli_fbase", C.c_void_p), ("dli_sname", C.c_char_p), ("dli_saddr", C.c_void_p)]
        info, path = DlInfo(), None
        libc = C.CDLL(None)
        libc.dladdr.argtypes = [C.c_void_p, C.POINTER(DlInfo)]
        if libc.dladdr(C.cast(L.ncclGetVersion, C.c_void_p), C.byref(info)) and info.dli_fname:
            path = info.dli_fname.decode()
        copies = set()
        try:
            for ln in open("/proc/self/maps"):
                f = ln.split()[-1]
                if "librccl" in f or "libnccl" in f:
                    copies.add(f)
        except OSError:
            pass
        return {"path": path, "version": version, "bound": cls._how, "copies_mapped": sorted(copies)}

    def describe(self):
        """library_info() + the number of ranks and this rank's index as the COMMUNICATOR reports them"""
        import ctypes as C
        L = self.lib()
        n, r = C.c_int(-1), C.c_int(-1)
        L.ncclCommCount(self.handle, C.byref(n))
        L.ncclCommUserRank(self.handle, C.byref(r))
        d = self.library_info()
        d.update(ranks=int(n.value), rank=int(r.value), communicator="raw ncclComm_t created by rgbd_recon_amd.dist.RcclComm")
        return d

    def __init__(self, rank, world, group=None, device=None):
        import ctypes as C

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        L = self.lib()
        uid = UniqueId()
        if rank == 0:
            rc = L.ncclGetUniqueId(C.byref(uid))
            if rc:
                raise RuntimeError("ncclGetUniqueId: %d" % rc)
        raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
        if world > 1:
            # the id travels through the existing process group; an nccl group only moves device tensors
            on_device = device is not None and dist.get_backend(group) == "nccl"
            t = raw.to(device) if on_device else raw
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            raw = t.cpu()
        C.memmove(C.byref(uid), bytes(raw.numpy().tobytes()), 128)
        self.handle = C.c_void_p()
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rc = L.ncclCommInitRank(C.byref(self.handle), world, uid, rank)
        if rc:
            raise RuntimeError("ncclCommInitRank: %d" % rc)
        self.rank, self.world = rank, world
        for name, args in (("ncclSend", [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                           ("ncclRecv", [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                           ("ncclAllReduce", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p])):
            getattr(L, name).argtypes = args

    def close(self):
        if self.handle:
            self.lib().ncclCommDestroy(self.handle)
            self.handle = None


class RawLoopbackGather:
    """One GPU standing in for rank `rank` of `world` of a sensor-sharded chain, through a raw one-rank communicator: the
    traffic of the frame gather is reproduced by RCCL send / recv of the OTHER ranks' frame layers to this process itself
    (same bytes; the layers hold an unsharded frame of the same static scene, which `ctx` must have processed before this
    is created), the brick counters are all-reduced in the one-rank communicator and the other ranks' counts added back.
    Called with a torch stream it enqueues there (dist.LaggedChain's side stream), else on the chain's stream."""

    def __init__(self, ctx, device, rank, world, comm):
        self.ctx, self.device, self.comm = ctx, device, comm
        n = ctx.cfg.num_sensors
        self.count, self.first = n // world, rank * (n // world)
        v = ctx.shard_view()
        torch.cuda.synchronize()
        self.all_counts = wrap_device_words(v.counters, v.num_bricks, device).clone()
        self.foreign = None
        self.scratch = torch.empty((v.sensor_bytes // 4) * (n - self.count), dtype=torch.int32, device=device)
        ctx.set_sensor_shard(self.first, self.count)

    def __call__(self, stream=None):
        import ctypes as C
        L, comm = RcclComm.lib(), self.comm.handle
        v = self.ctx.shard_view()
        words = v.sensor_bytes // 4
        sp = int(stream.cuda_stream) if stream is not None else int(v.stream)
        st = C.c_void_p(sp)
        lo_words, hi_words = self.first * words, (v.num_sensors - self.first - self.count) * words
        base = int(v.frames)

        def chk(rc, what):
            if rc:
                raise RuntimeError("%s failed with RCCL status %d" % (what, rc))
        chk(L.ncclGroupStart(), "ncclGroupStart")
        if lo_words:
            chk(L.ncclSend(C.c_void_p(base), lo_words, 3, 0, comm, st), "ncclSend")
            chk(L.ncclRecv(C.c_void_p(self.scratch.data_ptr()), lo_words, 3, 0, comm, st), "ncclRecv")
        if hi_words:
            chk(L.ncclSend(C.c_void_p(base + 4 * (self.first + self.count) * words), hi_words, 3, 0, comm, st), "ncclSend")
            chk(L.ncclRecv(C.c_void_p(self.scratch.data_ptr() + 4 * lo_words), hi_words, 3, 0, comm, st), "ncclRecv")
        chk(L.ncclAllReduce(C.c_void_p(v.counters), C.c_void_p(v.counters), v.num_bricks, 3, 0, comm, st), "ncclAllReduce")
        chk(L.ncclGroupEnd(), "ncclGroupEnd")
        with torch.cuda.stream(torch.cuda.ExternalStream(sp, device=self.device)):
            counters = wrap_device_words(v.counters, v.num_bricks, self.device)
            if self.foreign is None:
                self.foreign = self.all_counts - counters
            counters.add_(self.foreign)
        self.ctx.shard_gather_done()


class ManagedSlabExchange:
    """Per-step communication of one slab rank with everything enqueued by the LIBRARY on its own streams (the C ABI's
    managed forms): the halo exchange on the context's side stream behind events (rgbdr_halo_begin_step /
    rgbdr_halo_exchange_async / rgbdr_halo_wait) and, with `shard`, the sensor-sharded pre_* chain completed by
    rgbdr_shard_allgather on the chain's stream.  `loopback`: one GPU stands in for rank `rank` of `world` -- the
    communicator has one rank, the halo faces are exchanged with this process itself, and the gather's traffic is
    reproduced by RCCL send / recv of the other ranks' frame layers to this process itself on the chain's stream (the
    layers hold an unsharded frame of the same static scene; their brick counts are added back)."""

    def __init__(self, ctx, device, rank, world, group=None, shard=False, loopback=False):
        self.ctx, self.device, self.rank, self.world, self.loopback = ctx, device, rank, world, loopback
        self.comm = RcclComm(0 if loopback else rank, 1 if loopback else world, group, device)
        # the gather gets a communicator of its own: operations on ONE communicator execute in issue order whatever their
        # streams, so on the halo's communicator the gather of frame k+1 would wait for the face transfer of frame k
        # (0.45 ms for the 64 MiB faces of 1024^3 / 8), which is meant to overlap the whole next frame
        self.comm_gather = None
        me = 0 if loopback else rank
        self.peer_lo = (me if loopback else rank - 1) if rank > 0 else -1
        self.peer_hi = (me if loopback else rank + 1) if rank < world - 1 else -1
        self.shard = None
        self.loop_gather = None
        n = ctx.cfg.num_sensors
        if shard and n % world == 0 and world > 1:
            self.count, self.first = n // world, rank * (n // world)
            self.comm_gather = RcclComm(0 if loopback else rank, 1 if loopback else world, group, device)
            if loopback:
                self.loop_gather = RawLoopbackGather(ctx, device, rank, world, self.comm_gather)
            else:
                ctx.set_sensor_shard(self.first, self.count)
            self.shard = True

    def begin_step(self):
        self.ctx.halo_begin_step()

    def exchange_async(self):
        self.ctx.halo_exchange_async(self.comm.handle, self.peer_lo, self.peer_hi)

    def wait(self):
        self.ctx.halo_wait()

    def gather(self):
        if not self.shard:
            return
        if self.loop_gather is not None:
            self.loop_gather()
        else:
            self.ctx.shard_allgather(self.comm_gather.handle)

    def last_transfer_ms(self):
        try:
            return self.ctx.timer_ns("halo") * 1e-6
        except Exception:  # noqa: BLE001 -- timers off
            return None

    def close(self):
        self.comm.close()
        if self.comm_gather is not None:
            self.comm_gather.close()


class PeerCopySlabExchange:
    """The halo exchange of one slab rank by the COPY ENGINE (the C ABI's rgbdr_halo_export / rgbdr_halo_set_peer /
    rgbdr_halo_pull_async): every rank pulls its neighbours' staged faces with device-to-device copies from their staging
    sets, mapped once through HIP IPC -- no send / recv kernels next to the sweep, no collective enqueued per frame.  The
    exports (plain bytes) travel once through torch.distributed (`group`: a group that moves Python objects, e.g. gloo).
    Same interface as ManagedSlabExchange.  `loopback`: one context stands in for rank `rank` of `world` with itself as its
    neighbours (a neighbour in the caller's own process is used without IPC)."""

    def __init__(self, ctx, device, rank, world, group=None, loopback=False):
        self.ctx, self.device, self.rank, self.world, self.loopback = ctx, device, rank, world, loopback
        self.shard = None
        mine = ctx.halo_export()
        if loopback:
            exports = {rank - 1: mine, rank + 1: mine}
        else:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine, group=group)
            exports = dict(enumerate(gathered))
        self.sides = []
        if rank > 0:
            ctx.halo_set_peer(0, exports[rank - 1])
            self.sides.append(0)
        if rank < world - 1:
            ctx.halo_set_peer(1, exports[rank + 1])
            self.sides.append(1)
        if not loopback:
            dist.barrier(group=group)          # every rank has mapped its neighbours before anyone steps

    def begin_step(self):
        self.ctx.halo_begin_step()

    def exchange_async(self):
        self.ctx.halo_pull_async()

    def wait(self):
        self.ctx.halo_wait()

    def gather(self):
        return None

    def last_transfer_ms(self):
        try:
            return self.ctx.timer_ns("halo") * 1e-6
        except Exception:  # noqa: BLE001 -- timers off
            return None

    def close(self):
        self.ctx.sync()
        for side in self.sides:
            self.ctx.halo_set_peer(side, None)


class HaloExchanger:
    """The per-step halo exchange, off the critical path.

    At 1024^3 on 8 GPUs a slab face is 2 tile layers = 64 MiB per neighbour, about 1 ms on
    one xGMI link -- as long as a whole step.  Sending straight from the volume would make
    the next integrate (which overwrites those layers) wait for the transfer, so the
    boundary layers go through one of two staging sets and the transfer runs from there on a
    side stream while the next frames are processed.  With `ctx` the staging sets are the
    library's (rgbdr_halo_staging): the integrate sweep stores its boundary tiles there
    itself, no copy at all.  Without, they are torch buffers filled by a device-to-device
    copy after the sweep.  Per step k (b = k mod 2):
        begin_step()      compute stream: [wait: transfer k-2 done]; the sweep will fill set b
        ... ctx.integrate() ...
        exchange_async()  compute stream: record staged_k
                          side stream   : wait staged_k, send set b / receive into the halo layers,
                                          record done_k
    A consumer that samples across slab faces calls wait(stream) first.

    `ctx` must enqueue on `compute_stream` (ctx.set_stream(compute_stream.cuda_stream)).
    `via_host` is for a backend without stream-ordered device transport (gloo reads and
    writes device pointers from the host with no regard for streams): the staged layers
    travel through host tensors, with host synchronisation."""

    def __init__(self, tsdf_view, device, compute_stream, rank=None, world=None, group=None, via_host=False, ctx=None,
                 loopback=False):
        self.via_host = via_host
        self.loopback = loopback        # tests: see exchange_halo
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.group = group
        self.send_lo, self.send_hi, self.recv_lo, self.recv_hi = halo_views(tsdf_view, device)
        self.compute = compute_stream
        self.side = torch.cuda.Stream(device)
        self.ctx = ctx
        if ctx is not None:
            self.stage = []
            for b in range(2):
                lo, hi, nbytes = ctx.halo_staging(b)
                assert nbytes == self.send_lo.numel() * 4
                self.stage.append((wrap_device_floats(lo, nbytes // 4, device), wrap_device_floats(hi, nbytes // 4, device)))
        else:
            self.stage = [(torch.empty_like(self.send_lo), torch.empty_like(self.send_hi)) for _ in range(2)]
        self.done = [None, None]
        self.timing = [None, None]       # (start, end) events of the transfer that used stage[b]
        self.k = 0
        self.begun = False

    def begin_step(self):
        """before integrate(): claims the staging set of this step"""
        b = self.k & 1
        if self.done[b] is not None and not self.done[b].query():
            self.compute.wait_event(self.done[b])        # the transfer that last read stage[b] (skipped once it has completed)
        if self.ctx is not None:
            self.ctx.set_halo_staging(b)
        self.begun = True

    def exchange_async(self):
        """after integrate() has been enqueued on the compute stream"""
        if not self.begun:
            if self.ctx is not None:
                raise RuntimeError("HaloExchanger(ctx=...): call begin_step() before integrate()")
            self.begin_step()
        self.begun = False
        b = self.k & 1
        self.k += 1
        lo, hi = self.stage[b]
        with torch.cuda.stream(self.compute):
            if self.ctx is None:
                if self.rank > 0:
                    lo.copy_(self.send_lo, non_blocking=True)
                if self.rank < self.world - 1:
                    hi.copy_(self.send_hi, non_blocking=True)
            staged = torch.cuda.Event()
            staged.record(self.compute)
        with torch.cuda.stream(self.side):
            self.side.wait_event(staged)
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(self.side)
            if self.via_host:
                host = [lo.cpu(), hi.cpu(), torch.empty(lo.shape), torch.empty(hi.shape)]   # .cpu() waits for the side stream
                exchange_halo(*host, rank=self.rank, world=self.world, group=self.group)
                if self.rank > 0:
                    self.recv_lo.copy_(host[2])
                if self.rank < self.world - 1:
                    self.recv_hi.copy_(host[3])
            else:
                exchange_halo(lo, hi, self.recv_lo, self.recv_hi, rank=self.rank, world=self.world, group=self.group,
                              loopback=self.loopback)
            done = torch.cuda.Event(enable_timing=True)
            done.record(self.side)
        self.done[b] = done
        self.timing[b] = (t0, done)
        self.last = done

    def last_transfer_ms(self):
        """duration of the most recent transfer that has completed on the side stream (the
        "halo" timer of SURVEY 8b); None before any has"""
        for b in ((self.k - 1) & 1, self.k & 1):
            t = self.timing[b]
            if t is not None and t[1].query():
                return t[0].elapsed_time(t[1])
        return None

    def wait(self, stream=None):
        """make `stream` (default: the compute stream) wait for the newest halos"""
        if self.k:
            (stream or self.compute).wait_event(self.last)


class LaggedChain:
    """The sensor-sharded pre_* chain with its gather OFF the critical path: the sweep lags the chain by one frame.

    In the plain sharded schedule a rank's frame is  chain(k) -> gather(k) -> sweep(k): the all-gather of the packed frames
    (3.5 MB per rank at BASELINE configs[3]; ~20 us to the same GPU, an estimated 45-140 us over xGMI) sits between the
    chain and the sweep.  Here a second, chain-only context (`chain_ctx`: the same sensors, calibration, bounding box and
    brick size; its own token volume) runs frame k+1 on the SAME stream as -- so before, not under -- the sweep of frame k,
    its gather runs on a side stream under that sweep, and the sweeping context takes the completed frame with
    rgbdr_import_frame at the start of the next step:

        main stream :  import(k) | chain(k+1) | sweep(k)          import(k+1) | chain(k+2) | sweep(k+1) ...
        side stream :                 gather(k+1) ........              gather(k+2) ........

    Only the gather (RCCL's copy kernels) shares the GPU with the sweep; the chain's kernels do not (next to a sweep that
    refills every wave slot they take 8 x as long: profiles/r04_notes).  RCCL's kernel does not get onto the device either
    while ONE sweep launch refills every slot as it frees -- it ran when the sweep ended, the whole gather exposed again
    (profiles/r05_lag_timeline_lagged.txt) --, so the sweep is issued as `sweep_launches` = 2 launches
    (rgbdr_set_sweep_launches): the queue drains between them, the collective starts there and ends under the second half
    (r05_lag_timeline_lagged_split.txt; the drain costs the sweep 8-19 us).  The volume after push(frame k+1) is frame
    k's; flush() sweeps the last frame; close() puts the sweep back to one launch.  Results are those of the plain
    schedule, one frame later (tests/test_dist_gpu.py)."""

    def __init__(self, ctx, chain_ctx, device, gather, before_sweep=None, after_sweep=None, nccl_comm=None, sweep_launches=2):
        """gather: a FrameGather of `chain_ctx` (torch.distributed collectives, or the one-GPU loopback), or None with
        `nccl_comm`: a raw ncclComm_t -- then the LIBRARY enqueues the gather on the chain context's own stream and takes the
        frame over itself (rgbdr_shard_allgather_async / rgbdr_import_frame_from: what host::LaggedChain does in C++)"""
        self.ctx, self.chain, self.device, self.gather, self.comm = ctx, chain_ctx, device, gather, nccl_comm
        self.before_sweep, self.after_sweep = before_sweep, after_sweep      # halo hooks: begin_step / exchange_async
        chain_ctx.set_stream(ctx.stream())
        self.sweep_launches = sweep_launches
        ctx.set_sweep_launches(sweep_launches)
        self.main = torch.cuda.ExternalStream(int(ctx.stream()), device=device)
        self.side = torch.cuda.Stream(device)
        self.ev_chain, self.ev_gather, self.ev_imported = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        self.pending = None              # (frames pointer, counters pointer) of the frame whose gather is under way

    def _sweep_pending(self):
        if self.pending is None:
            return False
        frames, counters = self.pending
        self.ctx.clear_occupied_bricks()
        if self.comm is not None:
            self.ctx.import_frame_from(self.chain)       # waits for the chain and for its asynchronous gather
        else:
            self.ctx.import_frame(frames, counters, wait_event=self.ev_gather.cuda_event)
            # the pointer form orders read-after-write only (include/rgbdr.h): a sweeping context on its two-stream
            # schedule copies on its SECOND stream, and the chain of the next frame -- on the main stream -- must not
            # overwrite the source before that copy has run
            ps = int(self.ctx.shard_view().stream or 0)
            if ps and ps != int(self.ctx.stream()):
                self.ev_imported.record(torch.cuda.ExternalStream(ps, device=self.device))
                self.main.wait_event(self.ev_imported)
        return True

    def push(self, depth_ptr, color_ptr):
        """one step: take over the frame pushed before (its gather has had a whole sweep to finish), run the chain of this
        one, start its gather on the side stream, sweep the frame taken over"""
        have = self._sweep_pending()                 # main stream: [wait gather(k)] copy frame k out of the chain context
        a = self.chain
        a.update_device(depth_ptr, color_ptr)
        a.clear_occupied_bricks()
        a.process_textures()                         # main stream, after the copy above: frame k+1 overwrites frame k there
        v = a.shard_view()
        if self.comm is not None:
            a.shard_allgather_async(self.comm)       # the library's gather stream, behind the chain
        else:
            self.ev_chain.record(self.main)
            self.side.wait_event(self.ev_chain)
            if self.gather is not None:
                self.gather(stream=self.side)        # (marks the chain context's frame complete: shard_gather_done)
            self.ev_gather.record(self.side)
        self.pending = (int(v.frames), int(v.counters))
        if have:
            self._sweep()

    def _sweep(self):
        self.ctx.update_occupied_bricks()
        if self.before_sweep:
            self.before_sweep()
        self.ctx.integrate()
        if self.after_sweep:
            self.after_sweep()

    def flush(self):
        """sweep the frame pushed last"""
        if self._sweep_pending():
            self._sweep()
            self.pending = None

    def close(self):
        """flush, and the sweeping context back to one launch per sweep"""
        self.flush()
        self.ctx.set_sweep_launches(1)


def torch_rccl_info():
    """what carried the exchange when torch.distributed did: RCCL as torch reports it, the ranks of its process group"""
    try:
        version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        version = None
    copies = set()
    try:
        for ln in open("/proc/self/maps"):
            f = ln.split()[-1]
            if "librccl" in f or "libnccl" in f:
                copies.add(f)
    except OSError:
        pass
    return {"path": sorted(copies)[0] if len(copies) == 1 else None, "version": version, "bound": "torch.distributed (ProcessGroupNCCL)",
            "copies_mapped": sorted(copies), "ranks": dist.get_world_size(), "rank": dist.get_rank(),
            "communicator": "torch.distributed default process group"}


NO_HIT = 0x7FFFFFFF


def wrap_device_int32(ptr, n, device):
    class _P32:
        def __init__(self):
            self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}

    return torch.as_tensor(_P32(), device=device)


def exchange_halo_via_host(views, rank=None, world=None, group=None):
    """exchange_halo for a backend without device transport (gloo when several ranks
    share one GPU for debugging): the four halo_views are staged through the host."""
    host = [t.cpu() for t in views]
    exchange_halo(host[0], host[1], host[2], host[3], rank=rank, world=world, group=group)
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    if rank > 0:
        views[2].copy_(host[2])
    if rank < world - 1:
        views[3].copy_(host[3])


def composite_slab_frames(color, depth, mine, group=None):
    """Frames of the slabs -> one frame, by selection: `mine` marks the pixels this rank
    shaded (at most one rank per pixel), everything else holds the cleared values.
    The reduction runs on the int32 bit patterns: adding zeros to an integer is exact,
    so the owner's value arrives bit for bit (-0.0 and NaN payloads included)."""
    col = torch.where(mine[..., None], color, torch.zeros_like(color)).contiguous().view(torch.int32)
    dep = torch.where(mine, depth, torch.zeros_like(depth)).contiguous().view(torch.int32)
    own = mine.to(torch.int32)
    dist.all_reduce(col, group=group)
    dist.all_reduce(dep, group=group)
    dist.all_reduce(own, group=group)
    col, dep = col.view(torch.float32), dep.view(torch.float32)
    none = own == 0                                                  # no slab hit: the cleared frame
    col[none] = torch.tensor([0.0, 1.0, 0.0, 0.0], device=col.device)
    dep[none] = 1.0
    return col, dep


def raymarch_slabs(ctx, view, device, group=None, via_host=False):
    """Ray-march a volume split into Z slabs across the ranks of `group` (the TSDF halos
    must be current): every rank finds its first owned hit sample per pixel, one
    all-reduce MIN picks the global first hit, the owning rank shades it, and a second
    reduction composites the frames.  Returns (color [H,W,4], depth [H,W],
    num_samples [H,W]) torch tensors, identical on every rank and bit-identical to
    rgbdr_raymarch on a single context (tests/test_raymarch_gpu.py, tests/test_dist_gpu.py)."""
    h, w = view.height, view.width
    k = wrap_device_int32(ctx.raymarch_find(view), h * w, device)
    if via_host:
        kh = k.cpu()
        dist.all_reduce(kh, op=dist.ReduceOp.MIN, group=group)
        k.copy_(kh)
        torch.cuda.synchronize(device)
    else:
        dist.all_reduce(k, op=dist.ReduceOp.MIN, group=group)
        torch.cuda.synchronize(device)
    color, depth, ns = ctx.raymarch_shade(view)
    out_dev = torch.device("cpu") if via_host else device
    mine = (k != NO_HIT).reshape(h, w).to(out_dev)
    col, dep = composite_slab_frames(torch.from_numpy(color).to(out_dev), torch.from_numpy(depth).to(out_dev), mine, group)
    return col, dep, torch.from_numpy(ns).to(out_dev)
