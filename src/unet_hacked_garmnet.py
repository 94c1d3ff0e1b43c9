"""`from src.unet_hacked_garmnet import UNet2DConditionModel` (inference.py:40) -> GarmentNet on the HIP kernels."""
import idm_vton_amd  # noqa: F401
from idm_vton_amd.boundary.unet import GarmentUNet2DConditionModel as UNet2DConditionModel, UNet2DConditionOutput  # noqa: F401
