"""`AutoencoderKL` as the reference loads it (inference.py:233 `AutoencoderKL.from_pretrained(path, subfolder="vae",
torch_dtype=)`) and calls it (src/tryon_pipeline.py:924 encode -> latent_dist.sample, :1876 decode, :1076-1093
upcast_vae), executing idm_vton_amd.vae.HipVAE (NHWC implicit-GEMM convs, GroupNorm, single-head attention as two GEMMs +
row softmax).  Same state-dict keys as diffusers' AutoencoderKL; third-party semantics restated
from SURVEY.md A.3 / B.7.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ffi, ops
from ..config import VAEConfig, vae_param_shapes
from ..vae import HipVAE
from .modules import build_param_tree, params_version


class DiagonalGaussianDistribution:
    """`latent_dist` of encode(): holds the HIP-computed moments; sample() draws with torch's generator on the device."""

    def __init__(self, moments_nhwc, B, h, w, vae):
        self._mom, self._B, self._h, self._w, self._vae = moments_nhwc, B, h, w, vae

    def sample(self, generator=None):
        lc = self._vae.cfg.latent_channels
        noise = torch.randn((self._B, lc, self._h, self._w), generator=generator, device=self._mom.device, dtype=torch.float32)
        return ops.vae_sample(self._mom, noise, 1.0)

    def mode(self):
        lc = self._vae.cfg.latent_channels
        return ops.vae_sample(self._mom, torch.zeros((self._B, lc, self._h, self._w), device=self._mom.device), 1.0)


class AutoencoderKL(nn.Module):
    def __init__(self, config: VAEConfig = None, torch_dtype=torch.float32, device="cpu", **overrides):
        super().__init__()
        cfg = config or VAEConfig()
        for k, v in overrides.items():
            if k in VAEConfig.__dataclass_fields__:
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        self.cfg = cfg
        self.config = SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                      latent_channels=cfg.latent_channels, block_out_channels=cfg.block_out_channels,
                                      layers_per_block=cfg.layers_per_block, norm_num_groups=cfg.norm_num_groups,
                                      scaling_factor=cfg.scaling_factor, force_upcast=cfg.force_upcast)
        build_param_tree(self, vae_param_shapes(cfg), torch_dtype, device)
        self._hip, self._hip_key = None, None
        self.use_slicing = self.use_tiling = False          # diffusers AutoencoderKL.__init__ defaults

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=torch.float32, **kw):
        from safetensors.torch import load_file
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        cj = os.path.join(d, "config.json")
        if not os.path.isfile(cj):
            raise EnvironmentError(f"{cj} not found: from_pretrained needs a local diffusers-layout directory")
        raw = {k: v for k, v in json.load(open(cj)).items() if not k.startswith("_")}
        m = cls(torch_dtype=torch_dtype, device="meta", **raw)
        sd = load_file(os.path.join(d, "diffusion_pytorch_model.safetensors"))
        m.load_state_dict({k: v.to(torch_dtype) for k, v in sd.items()}, strict=True, assign=True)
        return m

    def save_pretrained(self, save_directory):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        c = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()}
        c["_class_name"] = "AutoencoderKL"
        json.dump(c, open(os.path.join(save_directory, "config.json"), "w"), indent=1)
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    # diffusers AutoencoderKL.enable_slicing / enable_tiling (called through the pipeline's enable_vae_slicing / enable_vae_tiling,
    # /root/reference/src/tryon_pipeline.py:427-457).  Slicing is what the HIP VAE does anyway (batch elements are independent; the batch is
    # walked in chunks that fit a 32-bit buffer descriptor): identical results.  Tiling in diffusers is an APPROXIMATION (overlapping tiles
    # blended) that exists to fit small GPUs; here the request is recorded and the exact untiled result is computed -- 288 GB of HBM hold a
    # 1024x768 decode (4.3 GB of activations) many times over.
    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def enable_tiling(self, use_tiling=True):
        self.use_tiling = bool(use_tiling)

    def disable_tiling(self):
        self.use_tiling = False

    def hip_engine(self):
        ffi.lib()
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("AutoencoderKL runs on the GPU only (HIP kernels); call .to('cuda') first")
        # The reference runs the VAE in fp32 whenever the module is fp16 and config.force_upcast is set (upcast_vae,
        # tryon_pipeline.py:911-930, 1868-1880): the SDXL VAE's activations overflow fp16.  The HIP VAE has fp32 accumulation,
        # fp32 GroupNorm statistics and an fp32 softmax; what it stores between layers is 16-bit, so the storage type must have
        # fp32's EXPONENT range: an fp16 (or fp32) module is executed with bf16 storage (same MFMA rate), never with fp16
        # storage, unless force_upcast is explicitly False (the fp16-fix VAE checkpoints set that).
        dt = p0.dtype
        if dt not in (torch.float16, torch.bfloat16) or (dt == torch.float16 and getattr(self.config, "force_upcast", True)):
            dt = torch.bfloat16
        key = (params_version(self), dt)
        if key != self._hip_key:
            self._hip = HipVAE(self.cfg, self.state_dict(), dt, p0.device)
            self._hip_key = key
        return self._hip

    def encode(self, x, return_dict=True):
        eng = self.hip_engine()
        mom, h, w = eng.encode_moments(x.to(eng.device, torch.float32))
        dist = DiagonalGaussianDistribution(mom, x.shape[0], h, w, self)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z, return_dict=True):
        eng = self.hip_engine()
        img = eng.decode(z.to(eng.device, torch.float32))
        return SimpleNamespace(sample=img) if return_dict else (img,)
