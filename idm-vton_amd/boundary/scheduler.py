"""`DDPMScheduler` as the reference instantiates it (inference.py:232 `DDPMScheduler.from_pretrained(path, subfolder="scheduler")`)
and as `StableDiffusionXLInpaintPipeline.__call__` drives it (src/tryon_pipeline.py:1561-1567 set_timesteps, :1772
scale_model_input, :1823 step).  Third-party semantics restated from SURVEY.md Appendix B.8 (diffusers==0.25.0): epsilon
prediction, fixed_small variance, leading spacing with steps_offset, scaled_linear betas.

Inside the pipeline the update is fused with the CFG combine into one HIP kernel (idmvton_cfg_step); this class supplies
the coefficient schedule (idm_vton_amd.scheduler.StepScheduler) and keeps `.step()` for API users.  `DDIMScheduler`
(eta = 0) is the deterministic variant the benchmark metric names.
"""
import json
import os
from types import SimpleNamespace

import torch

from ..scheduler import StepScheduler

_DEFAULTS = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, timestep_spacing="leading", prediction_type="epsilon", variance_type="fixed_small",
                 clip_sample=False, thresholding=False)


class _SchedulerBase:
    _kind = "ddpm"

    def __init__(self, **kw):
        cfg = dict(_DEFAULTS)
        unknown = set(kw) - set(cfg) - {"trained_betas", "dynamic_thresholding_ratio", "clip_sample_range", "sample_max_value",
                                         "set_alpha_to_one", "skip_prk_steps", "interpolation_type", "use_karras_sigmas",
                                         "rescale_betas_zero_snr"}
        if unknown:
            raise TypeError(f"unknown scheduler config keys {sorted(unknown)}")
        cfg.update({k: v for k, v in kw.items() if k in cfg})
        for k, want in (("beta_schedule", "scaled_linear"), ("timestep_spacing", "leading"), ("prediction_type", "epsilon"),
                        ("clip_sample", False), ("thresholding", False)):
            if cfg[k] != want:
                raise NotImplementedError(f"scheduler config {k}={cfg[k]!r}: the try-on path uses {want!r} (SURVEY.md A.1)")
        if self._kind == "ddpm" and cfg["variance_type"] != "fixed_small":
            raise NotImplementedError("DDPM variance_type other than 'fixed_small'")
        self.config = SimpleNamespace(**cfg)
        self._impl = StepScheduler(self._kind, cfg["num_train_timesteps"], cfg["beta_start"], cfg["beta_end"], cfg["steps_offset"],
                                   set_alpha_to_one=bool(kw.get("set_alpha_to_one", False)))
        self.config.set_alpha_to_one = bool(kw.get("set_alpha_to_one", False))
        self.init_noise_sigma = 1.0
        self.order = 1
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kw):
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        cj = os.path.join(d, "scheduler_config.json")
        if not os.path.isfile(cj):
            raise EnvironmentError(f"{cj} not found: from_pretrained needs a local diffusers-layout directory")
        raw = {k: v for k, v in json.load(open(cj)).items() if not k.startswith("_")}
        raw.update(kw)
        return cls(**raw)

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        c = dict(vars(self.config), _class_name=type(self).__name__)
        json.dump(c, open(os.path.join(save_directory, "scheduler_config.json"), "w"), indent=1)

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        self.num_inference_steps = num_inference_steps
        ts = self._impl.set_timesteps(num_inference_steps)
        self.timesteps = torch.from_numpy(ts.copy()).to(device) if device is not None else torch.from_numpy(ts.copy())

    def scale_model_input(self, sample, timestep=None):
        return sample                                  # identity for DDPM / DDIM

    def step(self, model_output, timestep, sample, generator=None, eta=0.0, return_dict=True, **kw):
        """x_{t-1} from eps (Appendix B.8).  Elementwise torch on the tensors' device; the pipeline never calls this (its
        update is fused into idmvton_cfg_step) -- it exists so code written against the diffusers scheduler API keeps working."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        c_x, c_eps, sigma = self._impl.coeffs(int(timestep))
        prev = c_x * sample.float() + c_eps * model_output.float()
        if sigma > 0.0:
            noise = torch.randn(model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            prev = prev + sigma * noise.float()
        prev = prev.to(sample.dtype)
        return SimpleNamespace(prev_sample=prev) if return_dict else (prev,)


class DDPMScheduler(_SchedulerBase):
    _kind = "ddpm"


class DDIMScheduler(_SchedulerBase):
    _kind = "ddim"
