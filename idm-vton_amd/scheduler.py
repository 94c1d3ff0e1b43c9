"""DDPM (reference default: inference.py:232) and DDIM eta=0 (benchmark metric wording) step coefficients.

Host-side mirror of diffusers' DDPMScheduler.set_timesteps/step as the reference drives it (src/tryon_pipeline.py:1561,
1823) with the SDXL scheduler_config values (A.1): 1000 train steps, scaled_linear betas 0.00085..0.012,
timestep_spacing="leading", steps_offset=1, epsilon prediction, fixed_small variance, no clipping.  The update itself
runs on the GPU (idmvton_cfg_step): x_prev = c_x*x + c_eps*eps + sigma*noise; this file only produces the scalars.
"""
import numpy as np


class StepScheduler:
    def __init__(self, kind="ddpm", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False):
        if kind not in ("ddpm", "ddim"):
            raise ValueError(f"unknown scheduler kind {kind!r}")
        self.kind, self.T, self.steps_offset = kind, num_train_timesteps, steps_offset
        import torch                      # same float32 linspace / cumprod ops as diffusers' DDPMScheduler.__init__
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double().numpy()
        # diffusers DDIMScheduler.final_alpha_cumprod: alphas_cumprod[0] unless set_alpha_to_one (SDXL scheduler_config: false);
        # used for the step whose previous timestep is < 0 (leading spacing + steps_offset=1: the LAST step, t = 1).
        # DDPMScheduler has no such switch: its previous alpha-bar is 1 there.
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.init_noise_sigma = 1.0
        self.order = 1

    def set_timesteps(self, n):
        self.n = n
        ratio = self.T // n
        self.timesteps = (np.arange(0, n) * ratio).round()[::-1].astype(np.int64) + self.steps_offset
        return self.timesteps

    def coeffs(self, t):
        """(c_x, c_eps, sigma) of x_prev = c_x*x + c_eps*eps + sigma*noise for timestep t."""
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
            sigma = max(bb_p / bb_t * b_t, 1e-20) ** 0.5 if t > 0 else 0.0
            return c_xt + c_x0 / ab_t ** 0.5, -c_x0 * (bb_t ** 0.5) / ab_t ** 0.5, sigma
        return (ab_p ** 0.5) / ab_t ** 0.5, bb_p ** 0.5 - (ab_p ** 0.5) * (bb_t ** 0.5) / ab_t ** 0.5, 0.0
