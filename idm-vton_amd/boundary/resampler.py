"""`ip_adapter.resampler.Resampler` (reference ip_adapter/resampler.py:129-176) with its forward on the HIP kernels.

Same constructor signature and defaults, same parameter names (`latents`, `proj_in`, `layers.{i}.0.{norm1,norm2,to_q,
to_kv,to_out}`, `layers.{i}.1.{0,1,3}`, `proj_out`, `norm_out`) so a reference state dict loads unchanged; `max_seq_len`,
`apply_pos_emb` and `num_latents_mean_pooled` are accepted and, as in the reference's forward (:164-176), unused.
forward(x) executes idm_vton_amd.resampler.HipResampler (LayerNorm / GEMM / two-segment flash-attention kernels).
"""
import torch
import torch.nn as nn

from .. import ffi
from ..resampler import HipResampler
from .modules import params_version


class _PerceiverAttention(nn.Module):
    """Parameter container of reference PerceiverAttention (:34-47); arithmetic runs in HipResampler."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


def _feed_forward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


class Resampler(nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len: int = 257, apply_pos_emb: bool = False, num_latents_mean_pooled: int = 0):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("HIP attention kernels are specialised for head_dim 64")
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([_PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    _feed_forward(dim=dim, mult=ff_mult)]) for _ in range(depth)])
        self._kw = dict(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_queries=num_queries, ff_mult=ff_mult)
        self._hip, self._hip_key = None, None

    def _engine(self, dtype, device):
        key = (params_version(self), dtype, device)
        if key != self._hip_key:
            self._hip = HipResampler(self.state_dict(), dtype=dtype, device=device, **self._kw)
            self._hip_key = key
        return self._hip

    def forward(self, x):
        ffi.lib()
        if not x.is_cuda:
            raise RuntimeError("Resampler runs on the GPU only (HIP kernels); there is no CPU fallback")
        if x.dtype not in (torch.float16, torch.bfloat16):
            raise TypeError(f"Resampler takes float16/bfloat16 input, got {x.dtype}")
        return self._engine(x.dtype, x.device)(x)
