"""Oracle DDPM / DDIM schedulers.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference instantiates diffusers' DDPMScheduler (inference.py:232) and calls `set_timesteps`/`step`
(src/tryon_pipeline.py:1561,1823).  Source not under /root/reference; restated from SURVEY.md B.8 with the SDXL
scheduler_config values recorded there (A.1).  Noise is supplied by the caller so runs are reproducible across devices.
"""
import numpy as np
import torch


class Scheduler:
    def __init__(self, kind="ddpm", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False):
        assert kind in ("ddpm", "ddim")
        self.kind = kind
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        # diffusers DDIMScheduler.final_alpha_cumprod: alphas_cumprod[0] unless set_alpha_to_one (SDXL scheduler_config: false);
        # used for the step whose previous timestep is < 0 (leading spacing + steps_offset=1: the LAST step, t = 1).
        # DDPMScheduler has no such switch: its previous alpha-bar is 1 there.
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.n = None

    def set_timesteps(self, n):
        """timestep_spacing="leading": (arange(n) * (T // n)).round()[::-1] + steps_offset."""
        self.n = n
        ratio = self.T // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)) + self.steps_offset
        return self.timesteps

    def coeffs(self, t):
        """Per-step scalars: x_prev = c_x0 * x0 + c_xt * x_t (+ sigma * noise), x0 = (x_t - sqrt(1-ab_t) eps)/sqrt(ab_t).

        Returned as (c_eps, c_x, sigma) so that x_prev = c_x * x_t + c_eps * eps + sigma * noise.
        """
        t = int(t)
        prev_t = t - self.T // self.n
        ab_t = float(self.alphas_cumprod[t])
        ab_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else (1.0 if self.kind == "ddpm" else self.final_alpha_cumprod)
        bb_t, bb_p = 1.0 - ab_t, 1.0 - ab_p
        if self.kind == "ddpm":
            a_t = ab_t / ab_p
            b_t = 1.0 - a_t
            c_x0 = (ab_p ** 0.5) * b_t / bb_t
            c_xt = (a_t ** 0.5) * bb_p / bb_t
            var = max(bb_p / bb_t * b_t, 1e-20)
            sigma = var ** 0.5 if t > 0 else 0.0
            c_x = c_xt + c_x0 / ab_t ** 0.5
            c_eps = -c_x0 * (bb_t ** 0.5) / ab_t ** 0.5
        else:  # DDIM eta=0: x_prev = sqrt(ab_p) x0 + sqrt(1-ab_p) eps
            c_x = (ab_p ** 0.5) / ab_t ** 0.5
            c_eps = bb_p ** 0.5 - (ab_p ** 0.5) * (bb_t ** 0.5) / ab_t ** 0.5
            sigma = 0.0
        return c_eps, c_x, sigma

    def add_noise(self, x0, noise, t):
        """diffusers SchedulerMixin.add_noise: sqrt(ab_t) x0 + sqrt(1 - ab_t) noise (used when strength < 1,
        src/tryon_pipeline.py:889-893)."""
        ab = float(self.alphas_cumprod[int(t)])
        return ab ** 0.5 * x0 + (1.0 - ab) ** 0.5 * noise

    def step(self, eps, t, x, noise=None):
        c_eps, c_x, sigma = self.coeffs(t)
        out = c_x * x + c_eps * eps
        if sigma > 0.0:
            out = out + sigma * noise
        return out
