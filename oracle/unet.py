"""Oracle TryonNet / GarmentNet (pure torch, fp32, CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

One class, two modes:
  mode="tryon"   restates src/unet_hacked_tryon.py:204 (+ unet_block_hacked_tryon.py, transformerhacked_tryon.py,
                 attentionhacked_tryon.py): consumes `garment_features`, IP-Adapter cross-attention.
  mode="garmnet" restates src/unet_hacked_garmnet.py:80 (+ *_garmnet.py twins): exports norm1 outputs, skips the
                 non-attention up block and conv_out.
Module/parameter names follow diffusers so state-dict keys match SURVEY.md Appendix C.
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import (Attention, AttnProcessor2_0, Downsample2D, FeedForward, IPAttnProcessor2_0, ResnetBlock2D,
                     TimestepEmbedding, Timesteps, Upsample2D)
from .resampler import Resampler


@dataclass
class UNetConfig:
    """SDXL values from SURVEY.md A.1 (class defaults at src/unet_hacked_tryon.py:301-356 are overridden by the
    checkpoint's config.json, which is not in the repo; train_xl.py:323-373 documents the surgery)."""
    mode: str = "tryon"
    in_channels: int = 13
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    num_attention_heads: Tuple[int, ...] = (5, 10, 20)      # diffusers calls this `attention_head_dim` (:366-372)
    cross_attention_dim: int = 2048
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = "text_time"
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    encoder_hid_dim_type: Optional[str] = "ip_image_proj"
    encoder_hid_dim: int = 1280
    ip_num_tokens: int = 16
    # Resampler hyper-parameters are hard-coded at src/unet_hacked_tryon.py:474-485; overridable for tiny tests.
    resampler: dict = field(default_factory=lambda: dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16,
                                                         ff_mult=4))

    @staticmethod
    def sdxl_tryon():
        return UNetConfig()

    @staticmethod
    def sdxl_garmnet():
        return UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None)


class BasicTransformerBlock(nn.Module):
    """tryon: src/attentionhacked_tryon.py:284-415; garmnet: src/attentionhacked_garmnet.py:284-406."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim, mode, ip_num_tokens):
        super().__init__()
        self.mode = mode
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)                                        # :199 (eps :147)
        self.attn1 = Attention(dim, heads, head_dim)                                    # :201-210
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)                                        # :229
        proc = (IPAttnProcessor2_0(dim, cross_attention_dim, scale=1.0, num_tokens=ip_num_tokens)
                if mode == "tryon" else None)                                           # unet_hacked_tryon.py:773-791
        self.attn2 = Attention(dim, heads, head_dim, cross_attention_dim=cross_attention_dim, processor=proc)  # :231-240
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)                                        # :256
        self.ff = FeedForward(dim)                                                      # :258-265

    def forward(self, x, encoder_hidden_states, garment_features=None, idx=0):
        n = self.norm1(x)                                                               # :310
        if self.mode == "tryon":
            m = torch.cat([n, garment_features[idx]], dim=1)                            # :334
            idx += 1                                                                    # :335
            a = self.attn1(m)                                                           # :336-342
            x = a[:, : x.shape[-2], :] + x                                              # :348
            exported = None
        else:
            exported = n                                                                # garmnet :321-322
            x = self.attn1(n) + x                                                       # garmnet :331-342
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x  # :365-384
        x = self.ff(self.norm3(x)) + x                                                  # :390-412
        return x, idx, exported


class Transformer2DModel(nn.Module):
    """src/transformerhacked_tryon.py:246-467 (use_linear_projection=True path)."""

    def __init__(self, heads, head_dim, in_channels, num_layers, cross_attention_dim, groups, mode, ip_num_tokens):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)                         # :148
        self.proj_in = nn.Linear(in_channels, inner)                                    # :150
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim, mode, ip_num_tokens)
             for _ in range(num_layers)])                                               # :184-205
        self.proj_out = nn.Linear(inner, in_channels)                                   # :213

    def forward(self, x, encoder_hidden_states, garment_features=None, idx=0):
        b, c, h, w = x.shape
        res = x                                                                         # :328
        x = self.norm(x)                                                                # :329
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)                                  # :341
        x = self.proj_in(x)                                                             # :342-346
        feats = []
        for blk in self.transformer_blocks:                                             # :370-411
            x, idx, f = blk(x, encoder_hidden_states, garment_features, idx)
            if f is not None:
                feats.append(f)
        x = self.proj_out(x)                                                            # :423
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()                     # :425
        return x + res, idx, feats                                                      # :427


class _CrossAttnMixin:
    def _attn(self, heads, ch, n_layers, cfg):
        return Transformer2DModel(heads, ch // heads, ch, n_layers, cfg.cross_attention_dim, cfg.norm_num_groups,
                                  cfg.mode, cfg.ip_num_tokens)


class DownBlock2D(nn.Module):
    """src/unet_block_hacked_tryon.py:1204-1289."""
    has_cross_attention = False

    def __init__(self, cin, cout, temb, n_layers, add_down, cfg):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, cfg.norm_num_groups,
                                                    cfg.norm_eps) for i in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb):
        outs = ()
        for r in self.resnets:
            x = r(x, temb)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class CrossAttnDownBlock2D(nn.Module, _CrossAttnMixin):
    """src/unet_block_hacked_tryon.py:1031-1201 (garmnet twin returns the feature list, :1136-1189 there)."""
    has_cross_attention = True

    def __init__(self, cin, cout, temb, n_layers, n_tf, heads, add_down, cfg):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, cfg.norm_num_groups,
                                                    cfg.norm_eps) for i in range(n_layers)])
        self.attentions = nn.ModuleList([self._attn(heads, cout, n_tf, cfg) for _ in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, enc, gf, idx):
        outs, feats = (), []
        for r, a in zip(self.resnets, self.attentions):
            x = r(x, temb)
            x, idx, f = a(x, enc, gf, idx)
            feats += f
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs, idx, feats


class UNetMidBlock2DCrossAttn(nn.Module, _CrossAttnMixin):
    """src/unet_block_hacked_tryon.py:630-781."""
    has_cross_attention = True

    def __init__(self, ch, temb, n_tf, heads, cfg):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([self._attn(heads, ch, n_tf, cfg)])

    def forward(self, x, temb, enc, gf, idx):
        x = self.resnets[0](x, temb)
        feats = []
        for a, r in zip(self.attentions, self.resnets[1:]):
            x, idx, f = a(x, enc, gf, idx)
            feats += f
            x = r(x, temb)
        return x, idx, feats


class CrossAttnUpBlock2D(nn.Module, _CrossAttnMixin):
    """src/unet_block_hacked_tryon.py:2217-2397."""
    has_cross_attention = True

    def __init__(self, cin, cout, prev, temb, n_layers, n_tf, heads, add_up, cfg):
        super().__init__()
        rs = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout                                   # :2253
            rin = prev if i == 0 else cout                                              # :2254
            rs.append(ResnetBlock2D(rin + skip, cout, temb, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([self._attn(heads, cout, n_tf, cfg) for _ in range(n_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, res_tuple, temb, enc, gf, idx, upsample_size=None):
        feats = []
        for r, a in zip(self.resnets, self.attentions):
            res, res_tuple = res_tuple[-1], res_tuple[:-1]                              # :2327-2328
            x = torch.cat([x, res], dim=1)                                              # :2346
            x = r(x, temb)
            x, idx, f = a(x, enc, gf, idx)
            feats += f
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)                                    # :2393-2395
        return x, idx, feats


class UpBlock2D(nn.Module):
    """src/unet_block_hacked_tryon.py:2400-2507."""
    has_cross_attention = False

    def __init__(self, cin, cout, prev, temb, n_layers, add_up, cfg):
        super().__init__()
        rs = []
        for i in range(n_layers):
            skip = cin if i == n_layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, cfg.norm_num_groups, cfg.norm_eps))
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, res_tuple, temb, upsample_size=None):
        for r in self.resnets:
            res, res_tuple = res_tuple[-1], res_tuple[:-1]
            x = torch.cat([x, res], dim=1)                                              # :2482
            x = r(x, temb)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        boc = cfg.block_out_channels
        temb = boc[0] * 4                                                               # unet_hacked_tryon.py:432
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)                 # :415-418
        self.time_proj = Timesteps(boc[0], True, 0)                                     # :434
        self.time_embedding = TimestepEmbedding(boc[0], temb)                           # :441-447
        if cfg.encoder_hid_dim_type == "ip_image_proj":                                 # :474-485
            self.encoder_hid_proj = Resampler(embedding_dim=cfg.encoder_hid_dim, output_dim=cfg.cross_attention_dim,
                                              **cfg.resampler)
        else:
            self.encoder_hid_proj = None
        if cfg.addition_embed_type == "text_time":                                      # :540-542
            self.add_time_proj = Timesteps(cfg.addition_time_embed_dim, True, 0)
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)

        nb = len(boc)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(cfg.down_block_types):                                    # :589-623
            in_ch, out_ch = out_ch, boc[i]
            final = i == nb - 1
            if t == "DownBlock2D":
                self.down_blocks.append(DownBlock2D(in_ch, out_ch, temb, cfg.layers_per_block, not final, cfg))
            else:
                self.down_blocks.append(CrossAttnDownBlock2D(in_ch, out_ch, temb, cfg.layers_per_block,
                                                             cfg.transformer_layers_per_block[i],
                                                             cfg.num_attention_heads[i], not final, cfg))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb, cfg.transformer_layers_per_block[-1],
                                                 cfg.num_attention_heads[-1], cfg)      # :626-643
        rboc = list(reversed(boc))
        rheads = list(reversed(cfg.num_attention_heads))
        rtf = list(reversed(cfg.transformer_layers_per_block))
        self.up_blocks = nn.ModuleList()
        out_ch = rboc[0]
        for i, t in enumerate(cfg.up_block_types):                                      # :693-744
            final = i == nb - 1
            prev, out_ch = out_ch, rboc[i]
            in_ch = rboc[min(i + 1, nb - 1)]
            if t == "UpBlock2D":
                self.up_blocks.append(UpBlock2D(in_ch, out_ch, prev, temb, cfg.layers_per_block + 1, not final, cfg))
            else:
                self.up_blocks.append(CrossAttnUpBlock2D(in_ch, out_ch, prev, temb, cfg.layers_per_block + 1, rtf[i],
                                                         rheads[i], not final, cfg))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)  # :747-750
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)               # :757-760
        self.num_upsamplers = nb - 1

    def time_embed(self, sample, timestep, added_cond_kwargs):
        """src/unet_hacked_tryon.py:1118-1213."""
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], dtype=torch.int64)
        if t.ndim == 0:
            t = t[None]
        t = t.to(sample.device).expand(sample.shape[0])                                 # :1132 (tests may run this module on the GPU)
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))                   # :1134-1141
        if self.cfg.addition_embed_type == "text_time":                                 # :1174-1190
            text_embeds = added_cond_kwargs["text_embeds"]
            time_ids = added_cond_kwargs["time_ids"]
            time_embeds = self.add_time_proj(time_ids.flatten()).reshape(text_embeds.shape[0], -1)
            add = torch.cat([text_embeds, time_embeds], dim=-1).to(emb.dtype)
            emb = emb + self.add_embedding(add)                                         # :1210
        return emb

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, garment_features=None):
        cfg = self.cfg
        tryon = cfg.mode == "tryon"
        factor = 2 ** self.num_upsamplers                                               # :1090-1099
        forward_upsample_size = any(s % factor != 0 for s in sample.shape[-2:])
        emb = self.time_embed(sample, timestep, added_cond_kwargs)
        if cfg.encoder_hid_dim_type == "ip_image_proj":                                 # :1234-1242 (already projected)
            encoder_hidden_states = torch.cat([encoder_hidden_states, added_cond_kwargs["image_embeds"]], dim=1)
        x = self.conv_in(sample)                                                        # :1245
        idx, feats = 0, []
        res = (x,)
        for blk in self.down_blocks:                                                    # :1282-1305
            if blk.has_cross_attention:
                x, outs, idx, f = blk(x, emb, encoder_hidden_states, garment_features, idx)
                feats += f
            else:
                x, outs = blk(x, emb)
            res += outs
        x, idx, f = self.mid_block(x, emb, encoder_hidden_states, garment_features, idx)  # :1320-1331
        feats += f
        for i, blk in enumerate(self.up_blocks):                                        # :1349-1381
            final = i == len(self.up_blocks) - 1
            n = len(blk.resnets)
            rs, res = res[-n:], res[:-n]
            up_size = res[-1].shape[2:] if (not final and forward_upsample_size) else None
            if blk.has_cross_attention:
                x, idx, f = blk(x, rs, emb, encoder_hidden_states, garment_features, idx, up_size)
                feats += f
            elif tryon:
                x = blk(x, rs, emb, up_size)
            # garmnet: no else-branch at unet_hacked_garmnet.py:1267-1279 -> non-attention up blocks are skipped
        if not tryon:
            return (x,), feats                                                          # garmnet :1281-1282
        x = self.conv_out(F.silu(self.conv_norm_out(x)))                                # :1383-1386
        return (x,)
