"""Host-side mirror of the reference's Python call surface and plugin API for the hot path.

  attention_processor.py  ip_adapter/attention_processor.py  AttnProcessor2_0 :189-278, IPAttnProcessor2_0 :1879-2010
  resampler.py            ip_adapter/resampler.py            Resampler :129-176
  modules.py              diffusers-Attention-like module + parameter-tree builder (state-dict keys of SURVEY App. C)
  unet.py                 src/unet_hacked_tryon.py / src/unet_hacked_garmnet.py  UNet2DConditionModel
  vae.py, scheduler.py    the AutoencoderKL / DDPMScheduler objects inference.py:232-233 hands to the pipeline
  tryon_pipeline.py       src/tryon_pipeline.py              StableDiffusionXLInpaintPipeline (:309, __call__ :1254-1894)

Same class names, constructor / call signatures, attribute names, state-dict keys and error behaviour as the
reference; every arithmetic op is a C-ABI call into libidmvton_hip.so (ops.py).  No class here has a CPU path: calling
one with CPU tensors, or without the built library, raises.  The top-level `src/` and `ip_adapter/` packages of this
repository re-export these classes under the import paths inference.py uses (inference.py:15,40-42).
"""
