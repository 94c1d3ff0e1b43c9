"""Weight preparation for the HIP kernels: layouts the C ABI expects (see include/idmvton_hip.h, idmvton_gemm_conv).

Pure tensor re-arrangements done once at load time; no arithmetic on the hot path lives here.
"""
import torch


def conv_weight_nhwc(w):
    """nn.Conv2d weight [Cout][Cin][kh][kw] -> GEMM weight [Cout][kh*kw*Cin] (tap-major, channels contiguous)."""
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).contiguous()


def pad_k(w2d, k_to):
    """Zero-pad the K (last) dim of a [N][K] weight to `k_to` (inputs with fewer than 64 channels are zero padded)."""
    n, k = w2d.shape
    if k == k_to:
        return w2d.contiguous()
    out = torch.zeros(n, k_to, dtype=w2d.dtype, device=w2d.device)
    out[:, :k] = w2d
    return out


def conv_weight_nhwc_padded(w, cin_pad):
    """Conv weight for an input stored with `cin_pad` (>= Cin) channels per pixel: each tap's K block is zero padded."""
    co, ci, kh, kw = w.shape
    out = torch.zeros(co, kh, kw, cin_pad, dtype=w.dtype, device=w.device)
    out[..., :ci] = w.permute(0, 2, 3, 1)
    return out.reshape(co, kh * kw * cin_pad).contiguous()


def interleave_geglu(w, b):
    """diffusers GEGLU.proj weight [2*inner][C] = [h rows ; gate rows] -> 64-row blocks [32 h | 32 gate] so that the
    GEMM epilogue finds h_j and gate_j in the same lane (csrc/gemm_conv.hip, mode GEGLU).  inner % 32 == 0."""
    two_inner, c = w.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    wi = torch.stack([w[:inner].reshape(inner // 32, 32, c), w[inner:].reshape(inner // 32, 32, c)], dim=1)
    wi = wi.reshape(two_inner, c).contiguous()
    bi = None
    if b is not None:
        bi = torch.stack([b[:inner].reshape(inner // 32, 32), b[inner:].reshape(inner // 32, 32)], dim=1)
        bi = bi.reshape(two_inner).contiguous()
    return wi, bi
