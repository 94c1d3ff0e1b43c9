"""`ip_adapter.attention_processor` (reference :189-278, :1879-2010): the processors TryonNet installs, on the HIP kernels."""
import idm_vton_amd  # noqa: F401
from idm_vton_amd.boundary.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0  # noqa: F401
