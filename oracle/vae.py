"""Oracle SDXL AutoencoderKL (pure torch, fp32, CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

The reference calls diffusers' AutoencoderKL (src/tryon_pipeline.py:924,1646,1876); its source is not under
/root/reference.  Block structure follows the verbatim diffusers-0.25 copies the reference carries in
src/unet_block_hacked_tryon.py (`DownEncoderBlock2D :1292-1349`, `UpDecoderBlock2D :2511-2568`,
`UNetMidBlock2D :505-627`) and SURVEY.md A.3 / B.7.  The three block types are pinned to those reference-held classes
(tests/test_oracle.py::test_oracle_vae_blocks_match_reference_code_golden); the Encoder / Decoder wiring around them is
"parity unpinned" (see oracle/__init__.py).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import Downsample2D, ResnetBlock2D, Upsample2D


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.13025
    force_upcast: bool = True


class VAEAttention(nn.Module):
    """Single-head spatial attention of UNetMidBlock2D (unet_block_hacked_tryon.py:585-597): GroupNorm, q/k/v/out
    Linear WITH bias, softmax(QK^T/sqrt(C)), residual_connection=True, rescale_output_factor=1."""

    def __init__(self, ch, groups, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        a = torch.softmax((q @ k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = self.to_out[0](a @ v)
        return o.transpose(1, 2).reshape(b, c, h, w) + res


class MidBlock(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, groups, 1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([VAEAttention(ch, groups)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)                                           # called on the 4-D map (:624)
        return self.resnets[1](x)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cin, cout, n, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(n)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, c in enumerate(boc):
            cin, out = out, c
            self.down_blocks.append(DownEncoderBlock2D(cin, out, cfg.layers_per_block, g, i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_num_groups
        rboc = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlock(boc[-1], g)
        self.up_blocks = nn.ModuleList()
        out = rboc[0]
        for i, c in enumerate(rboc):
            cin, out = out, c
            self.up_blocks.append(UpDecoderBlock2D(cin, out, cfg.layers_per_block + 1, g, i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    def encode_moments(self, x):
        """-> (mean, std) of the diagonal Gaussian posterior (logvar clamped to [-30, 20])."""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, torch.exp(0.5 * logvar.clamp(-30.0, 20.0))

    def encode_sample(self, x, noise):
        """`vae.encode(x).latent_dist.sample()` with the N(0,1) draw supplied by the caller (tryon_pipeline.py:255)."""
        mean, std = self.encode_moments(x)
        return mean + std * noise

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))
