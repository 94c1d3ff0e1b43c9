"""`ip_adapter.resampler.Resampler` (reference :129-176) on the HIP kernels."""
import idm_vton_amd  # noqa: F401
from idm_vton_amd.boundary.resampler import Resampler  # noqa: F401
