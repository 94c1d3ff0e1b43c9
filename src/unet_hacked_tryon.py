"""`from src.unet_hacked_tryon import UNet2DConditionModel` (inference.py:41) -> TryonNet on the HIP kernels."""
import idm_vton_amd  # noqa: F401
from idm_vton_amd.boundary.unet import TryonUNet2DConditionModel as UNet2DConditionModel, UNet2DConditionOutput  # noqa: F401
