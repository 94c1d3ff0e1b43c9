"""Oracle Resampler (Perceiver).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates /root/reference/ip_adapter/resampler.py (all of it that `Resampler.forward` :164-176 reaches).
PINNED: checked bit-for-bit against that file imported verbatim (oracle/make_golden.py,
tests/test_oracle.py::test_resampler_matches_reference_file).
"""
import math

import torch
import torch.nn as nn


class PerceiverAttention(nn.Module):
    """ip_adapter/resampler.py:34-78."""

    def __init__(self, dim, dim_head, heads):
        super().__init__()
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, x, latents):
        x = self.norm1(x)                                                   # :57
        latents = self.norm2(latents)                                       # :58
        b, l, _ = latents.shape
        q = self.to_q(latents)                                              # :62
        k, v = self.to_kv(torch.cat((x, latents), dim=-2)).chunk(2, dim=-1)  # :63-64
        sp = lambda t: t.view(b, t.shape[1], self.heads, -1).transpose(1, 2)  # reshape_tensor :23-31
        q, k, v = sp(q), sp(k), sp(v)
        scale = 1 / math.sqrt(math.sqrt(self.dim_head))                     # :71
        w = (q * scale) @ (k * scale).transpose(-2, -1)                     # :72
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)                  # :73
        out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)                 # :74-76
        return self.to_out(out)                                             # :78


def FeedForward(dim, mult=4):
    """ip_adapter/resampler.py:13-20."""
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(),
                         nn.Linear(inner, dim, bias=False))


class Resampler(nn.Module):
    """ip_adapter/resampler.py:129-176 (max_seq_len/apply_pos_emb/num_latents_mean_pooled are accepted and, as in
    the reference's forward, unused)."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len=257, apply_pos_emb=False, num_latents_mean_pooled=0):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)  # :145
        self.proj_in = nn.Linear(embedding_dim, dim)                                # :147
        self.proj_out = nn.Linear(dim, output_dim)                                  # :149
        self.norm_out = nn.LayerNorm(output_dim)                                    # :150
        self.layers = nn.ModuleList([
            nn.ModuleList([PerceiverAttention(dim, dim_head, heads), FeedForward(dim, ff_mult)])
            for _ in range(depth)])                                                 # :152-162

    def forward(self, x):
        latents = self.latents.repeat(x.size(0), 1, 1)                              # :166
        x = self.proj_in(x)                                                         # :168
        for attn, ff in self.layers:                                                # :171-173
            latents = attn(x, latents) + latents
            latents = ff(latents) + latents
        return self.norm_out(self.proj_out(latents))                                # :175-176
