"""The two attention processors the reference installs on TryonNet (src/unet_hacked_tryon.py:773-791), on the HIP kernels.

Mirror of /root/reference/ip_adapter/attention_processor.py:
  AttnProcessor2_0    :189-278    q/k/v projection -> SDPA -> to_out            (self-attention, and GarmentNet's attn2)
  IPAttnProcessor2_0  :1879-2010  text SDPA + scale * image-token SDPA -> to_out (TryonNet attn2)
Same constructor arguments, attributes, parameters (`to_k_ip.weight`, `to_v_ip.weight`) and `__call__` signature
`(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0)`.  `attn` is a
diffusers-Attention-like object (boundary/modules.py:Attention); only its weights are read -- the projections run as
one MFMA GEMM with K and V^T epilogues (idmvton_gemm_conv) and the SDPA as idmvton_attn_fwd.

Differences from the reference, all loud: attention masks, spatial_norm / group_norm / norm_cross and head_dim != 64
raise NotImplementedError (the try-on path uses none of them); fp32 tensors raise TypeError (the kernels compute in
fp16/bf16 storage with fp32 accumulation, as the reference does under autocast: inference.py:223,339); the write-only
`attn_map` debug tensor (:1989-1990, never read anywhere) is not produced.
"""
import torch
import torch.nn as nn

from .. import ffi, ops


def _check_attn(attn, attention_mask):
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is not supported by the HIP attention kernels")
    if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None:
        raise NotImplementedError("spatial_norm / group_norm attention variants are not on the try-on path")
    if getattr(attn, "norm_cross", None):
        raise NotImplementedError("norm_cross is not on the try-on path")
    inner = attn.to_q.weight.shape[0]
    if inner % attn.heads != 0 or inner // attn.heads != 64:
        raise NotImplementedError(f"HIP attention kernels are specialised for head_dim 64 (got {inner}/{attn.heads})")
    return inner


def _tokens(hidden_states):
    """[B,C,H,W] -> [B,HW,C] exactly as the reference does (:218-222); returns (tokens, restore-info)."""
    if hidden_states.ndim == 4:
        b, c, h, w = hidden_states.shape
        return hidden_states.view(b, c, h * w).transpose(1, 2).contiguous(), (b, c, h, w)
    if hidden_states.ndim != 3:
        raise ValueError(f"hidden_states must be 3-D or 4-D, got {hidden_states.ndim}-D")
    return hidden_states.contiguous(), None


class _WeightCat:
    """cat([w_a, w_b, ...]) cached until any source weight is re-assigned or modified in place."""

    def __init__(self):
        self.key, self.w, self.b = None, None, None

    def get(self, linears):
        key = tuple((l.weight.data_ptr(), l.weight._version, None if l.bias is None else l.bias._version) for l in linears)
        if key != self.key:
            self.w = torch.cat([l.weight.detach() for l in linears]).contiguous()
            bs = [l.bias for l in linears]
            if any(b is not None for b in bs):
                self.b = torch.cat([(b.detach() if b is not None else torch.zeros_like(l.weight[:, 0]))
                                    for b, l in zip(bs, linears)]).contiguous()
            else:
                self.b = None
            self.key = key
        return self.w, self.b


def _project_kv(enc, w_kv, b_kv, inner):
    """enc [B][L][Ck] -> (K [B*L8][inner], V^T [B][inner][L8], L8): one GEMM, K rows / V^T epilogues; token rows padded
    with zeros to a multiple of 16 (the V^T epilogue writes the attention kernel's key order; padded keys are masked by nk)."""
    B, L, Ck = enc.shape
    L8 = ops.round16(L)
    if L8 != L:
        pad = torch.zeros(B, L8, Ck, dtype=enc.dtype, device=enc.device)
        pad[:, :L] = enc
        enc = pad
    k = torch.empty(B * L8, inner, dtype=enc.dtype, device=enc.device)
    vt = torch.empty(B, inner, L8, dtype=enc.dtype, device=enc.device)
    ops.linear(enc.reshape(B * L8, Ck), w_kv, bias=b_kv, out=k, vt=vt, vt_n0=inner, vt_tokens=L8)
    return k, vt, L8


def _finish(attn, att, residual, shape4, B, L):
    """to_out[0] (+bias, + fused residual) -> to_out[1] (dropout) -> layout restore -> rescale (:266-276)."""
    lin = attn.to_out[0]
    fuse_res = attn.residual_connection and shape4 is None
    res2d = residual.reshape(B * L, -1) if fuse_res else None
    out = ops.linear(att, lin.weight.detach(), bias=None if lin.bias is None else lin.bias.detach(), res=res2d)
    out = attn.to_out[1](out.view(B, L, -1))
    if shape4 is not None:
        b, c, h, w = shape4
        out = out.transpose(-1, -2).reshape(b, c, h, w)
        if attn.residual_connection:
            out = out + residual
    if attn.rescale_output_factor != 1.0:
        out = out / attn.rescale_output_factor
    return out


class AttnProcessor2_0(nn.Module):
    """Scaled-dot-product attention processor (reference :189-278) on the gfx950 flash-attention kernel."""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()
        ffi.lib()                                    # reference raises ImportError without torch 2.0 SDPA (:199-200);
        self._qkv, self._kv = _WeightCat(), _WeightCat()   # here the requirement is the built HIP library

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        residual = hidden_states
        inner = _check_attn(attn, attention_mask)
        x, shape4 = _tokens(hidden_states)
        B, L, C = x.shape
        dt, dev = x.dtype, x.device
        att = torch.empty(B * L, inner, dtype=dt, device=dev)
        if encoder_hidden_states is None:
            # self-attention: one fused QKV GEMM (q|k row-major, V^T epilogue); keys == queries (:240-246)
            w, b = self._qkv.get([attn.to_q, attn.to_k, attn.to_v])
            if L % 16 == 0:
                qk = torch.empty(B * L, 2 * inner, dtype=dt, device=dev)
                vt = torch.empty(B, inner, L, dtype=dt, device=dev)
                ops.linear(x.reshape(B * L, C), w, bias=b, out=qk, vt=vt, vt_n0=2 * inner, vt_tokens=L, colscale_n=inner, colscale=ops.QSCALE)
                seg = dict(k=qk[:, inner:], vt=vt, nk=L, ldk=2 * inner, ldvt=L)
                ops.attention(qk, att, [seg], attn.heads, B=B, Nq=L, ldq=2 * inner, ldo=inner, q_prescaled=True)
                return _finish(attn, att, residual, shape4, B, L)
            q = ops.linear(x.reshape(B * L, C), w[:inner], bias=None if b is None else b[:inner])
            k, vt, L8 = _project_kv(x, w[inner:], None if b is None else b[inner:], inner)
        else:
            enc = encoder_hidden_states.contiguous()
            q = ops.linear(x.reshape(B * L, C), attn.to_q.weight.detach(),
                           bias=None if attn.to_q.bias is None else attn.to_q.bias.detach())
            w, b = self._kv.get([attn.to_k, attn.to_v])
            k, vt, L8 = _project_kv(enc, w, b, inner)
        nk = x.shape[1] if encoder_hidden_states is None else encoder_hidden_states.shape[1]
        seg = dict(k=k, vt=vt, nk=nk, ldk=inner, ldvt=L8, k_rows=L8)
        ops.attention(q, att, [seg], attn.heads, B=B, Nq=L, ldq=inner, ldo=inner)
        return _finish(attn, att, residual, shape4, B, L)


class IPAttnProcessor2_0(nn.Module):
    r"""IP-Adapter attention processor (reference :1879-2010): the last `num_tokens` rows of `encoder_hidden_states`
    are image tokens with their own K/V projections; out = SDPA(q, K_text, V_text) + scale * SDPA(q, K_ip, V_ip).

    Args (same as the reference): hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4 (16 for IDM-VTON).
    """

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        ffi.lib()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self._kv, self._kv_ip = _WeightCat(), _WeightCat()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        residual = hidden_states
        inner = _check_attn(attn, attention_mask)
        if encoder_hidden_states is None:
            # the reference leaves `ip_hidden_states` undefined on this branch and dies with a NameError (:1945-1978)
            raise ValueError("IPAttnProcessor2_0 needs encoder_hidden_states = cat([text tokens, image tokens], dim=1)")
        x, shape4 = _tokens(hidden_states)
        B, L, C = x.shape
        end_pos = encoder_hidden_states.shape[1] - self.num_tokens                       # :1949
        if end_pos <= 0:
            raise ValueError(f"encoder_hidden_states has {encoder_hidden_states.shape[1]} tokens, need more than "
                             f"num_tokens={self.num_tokens}")
        text = encoder_hidden_states[:, :end_pos, :].contiguous()
        ip = encoder_hidden_states[:, end_pos:, :].contiguous()
        q = ops.linear(x.reshape(B * L, C), attn.to_q.weight.detach(),
                       bias=None if attn.to_q.bias is None else attn.to_q.bias.detach())
        w, b = self._kv.get([attn.to_k, attn.to_v])
        kt, vtt, Lt8 = _project_kv(text, w, b, inner)                                    # :1957-1958
        wi, bi = self._kv_ip.get([self.to_k_ip, self.to_v_ip])
        ki, vti, Li8 = _project_kv(ip, wi, bi, inner)                                    # :1978-1979
        att = torch.empty(B * L, inner, dtype=x.dtype, device=x.device)
        segs = [dict(k=kt, vt=vtt, nk=end_pos, ldk=inner, ldvt=Lt8, k_rows=Lt8),
                dict(k=ki, vt=vti, nk=self.num_tokens, ldk=inner, ldvt=Li8, k_rows=Li8)]
        ops.attention(q, att, segs, attn.heads, mode=ffi.ATTN_CROSS, ip_scale=float(self.scale), B=B, Nq=L,
                      ldq=inner, ldo=inner)                                              # :1970-1995
        return _finish(attn, att, residual, shape4, B, L)
