"""idm_vton_amd -- MI355X-native (gfx950) implementation of the IDM-VTON denoising hot path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed only) calling the hand-written HIP kernels
of `libidmvton_hip.so` through the C ABI declared in `include/idmvton_hip.h`.  There is no CPU / eager fallback: every
op raises if the HIP library is missing (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C idm-vton_amd/csrc`).
"""
from . import ffi  # noqa: F401
