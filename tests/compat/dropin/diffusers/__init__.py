"""Name shim that lets the reference's UNMODIFIED scripts (`inference.py:33-38`, `gradio_demo/app.py`) import `diffusers` on a box
that has only this repository: `AutoencoderKL` and `DDPMScheduler` resolve to the MI355X boundary classes, everything the
scripts import but never execute is a placeholder.  Test / demo environment only (tests/compat/dropin on PYTHONPATH); with the
real diffusers installed the scripts keep importing that and only `src.*` / `ip_adapter.*` come from this repository."""
from idm_vton_amd.boundary.scheduler import DDIMScheduler, DDPMScheduler  # noqa: F401
from idm_vton_amd.boundary.vae import AutoencoderKL  # noqa: F401

from . import utils  # noqa: F401

__version__ = "0.25.0"


class _NotOnThePath:
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is imported by the reference scripts but never used on the try-on path")


class StableDiffusionPipeline(_NotOnThePath):
    pass


class StableDiffusionXLControlNetInpaintPipeline(_NotOnThePath):
    pass
