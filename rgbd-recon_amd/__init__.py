"""rgbd-recon_amd -- MI355X-native TSDF fusion + depth preprocessing backend.

The product is the C-ABI library ``librgbdr_hip.so`` (sources in ``csrc/``,
header ``include/rgbdr.h``) plus the C++ host mirror in ``host/``.  This Python
package is only the test / benchmark harness around that ABI:

* ``capi``  -- ctypes bindings of include/rgbdr.h (fails loudly if the library
  is missing: there is no Python or CPU fallback for the hot path),
* ``synth`` -- the deterministic synthetic scene of SURVEY.md section 8(d),
* ``dist``  -- Z-slab halo exchange over torch.distributed (RCCL / gloo).

The directory name contains a hyphen, so import it through
``__graft_entry__.load_package()`` (registers it as ``rgbd_recon_amd``).
"""
