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
    """(send_lo, send_hi, recv_lo, recv_hi) float tensors aliasing the layers of a
    rgbdr_tsdf_device_view with halo_layers == 1."""
    n = view.layer_bytes // 4
    base, owned = int(view.base), int(view.owned)
    last = owned + (view.owned_layers - 1) * view.layer_bytes
    hi_halo = owned + view.owned_layers * view.layer_bytes
    return (wrap_device_floats(owned, n, device), wrap_device_floats(last, n, device),
            wrap_device_floats(base, n, device), wrap_device_floats(hi_halo, n, device))


def exchange_halo(send_lo, send_hi, recv_lo, recv_hi, rank=None, world=None, group=None):
    """Neighbour exchange of one tile layer per slab face.  Slab r's lowest layer
    goes to r-1's upper halo, its highest layer to r+1's lower halo.  The same
    function runs on CUDA tensors (RCCL) and CPU tensors (gloo)."""
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
