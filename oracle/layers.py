"""Oracle building blocks (pure torch, fp32, CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates the diffusers==0.25.0 layers the reference constructs (source not under /root/reference;
semantics per SURVEY.md Appendix B) plus the reference's own attention processors.  Parameter names
follow diffusers so state-dict keys match SURVEY.md Appendix C.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# Embeddings -- used at src/unet_hacked_tryon.py:430-445,540-542 (ctor) and :1134-1141,1174-1190 (fwd)
# ------------------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    """diffusers `get_timestep_embedding`: sin|cos of t*exp(-ln(P)*i/(half-shift)), flipped to cos|sin."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    """Linear -> SiLU -> Linear."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(F.silu(self.linear_1(sample)))


# ------------------------------------------------------------------------------------------------
# ResNet / sampling blocks -- constructed at src/unet_block_hacked_tryon.py:1068-1079,1113-1115,2258-2269,2301
# ------------------------------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D, time_embedding_norm="default", output_scale_factor=1."""

    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    """use_conv=True.  padding=1: conv s2 p1 (UNet); padding=0: F.pad(0,1,0,1) then conv s2 p0 (VAE)."""

    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    """nearest x2 (or to `output_size`) then conv3x3 (forwarded size: unet_hacked_tryon.py:1357-1358)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


# ------------------------------------------------------------------------------------------------
# Feed-forward -- src/attentionhacked_tryon.py:621-679 + diffusers GEGLU
# ------------------------------------------------------------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)  # erf form


class FeedForward(nn.Module):
    """net = [GEGLU(dim, 4*dim), Dropout(0), Linear(4*dim, dim)]  (attentionhacked_tryon.py:656-667)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


# ------------------------------------------------------------------------------------------------
# Attention module + the two processors the reference instantiates
# ------------------------------------------------------------------------------------------------
def sdpa(q, k, v):
    """F.scaled_dot_product_attention(q,k,v) with no mask/dropout, written out (scale = d^-0.5)."""
    w = torch.softmax((q @ k.transpose(-2, -1)) * (q.shape[-1] ** -0.5), dim=-1)
    return w @ v


class AttnProcessor2_0(nn.Module):
    """Restates ip_adapter/attention_processor.py:189-278 for the path taken (3-D input, no mask/norms)."""

    def forward(self, attn, hidden_states, encoder_hidden_states=None):
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                                   # :238
        enc = hidden_states if encoder_hidden_states is None else encoder_hidden_states  # :240-243
        k, v = attn.to_k(enc), attn.to_v(enc)                                          # :245-246
        hd = k.shape[-1] // attn.heads
        q = q.view(b, -1, attn.heads, hd).transpose(1, 2)                              # :251-254
        k = k.view(b, -1, attn.heads, hd).transpose(1, 2)
        v = v.view(b, -1, attn.heads, hd).transpose(1, 2)
        o = sdpa(q, k, v)                                                              # :258-260
        o = o.transpose(1, 2).reshape(b, -1, attn.heads * hd)                          # :262
        return attn.to_out[1](attn.to_out[0](o))                                       # :266-268


class IPAttnProcessor2_0(nn.Module):
    """Restates ip_adapter/attention_processor.py:1879-2010 (text SDPA + scale * image-token SDPA).

    The write-only `attn_map` side effect (:1989-1990) is not restated: nothing reads it.
    """

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.scale, self.num_tokens = scale, num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)  # :1904
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)  # :1905

    def forward(self, attn, hidden_states, encoder_hidden_states=None):
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                                   # :1943
        end = encoder_hidden_states.shape[1] - self.num_tokens                         # :1949-1953
        text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        k, v = attn.to_k(text), attn.to_v(text)                                        # :1957-1958
        hd = k.shape[-1] // attn.heads
        sp = lambda t: t.view(b, -1, attn.heads, hd).transpose(1, 2)
        q = sp(q)
        o = sdpa(q, sp(k), sp(v)).transpose(1, 2).reshape(b, -1, attn.heads * hd)      # :1970-1975
        ipk, ipv = self.to_k_ip(ip), self.to_v_ip(ip)                                  # :1978-1979
        oi = sdpa(q, sp(ipk), sp(ipv)).transpose(1, 2).reshape(b, -1, attn.heads * hd)  # :1986-1992
        o = o + self.scale * oi                                                        # :1995
        return attn.to_out[1](attn.to_out[0](o))                                       # :1998-2000


class Attention(nn.Module):
    """diffusers Attention as built at src/attentionhacked_tryon.py:201-210,231-240:
    to_q/k/v without bias, to_out[0] with bias, scale d^-0.5, no residual/rescale/norms."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None, processor=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])
        self.processor = processor if processor is not None else AttnProcessor2_0()

    def forward(self, hidden_states, encoder_hidden_states=None):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states)
