"""`from src.tryon_pipeline import StableDiffusionXLInpaintPipeline` (inference.py:42, gradio_demo/app.py:13) -> MI355X engine."""
import idm_vton_amd  # noqa: F401  (registers the package under its importable name)
from idm_vton_amd.boundary.tryon_pipeline import StableDiffusionXLInpaintPipeline  # noqa: F401
