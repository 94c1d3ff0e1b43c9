"""Kernel-level parity checks: each C-ABI op vs a plain PyTorch fp32 reference of the same op on the same
(dtype-rounded) inputs.  Used by tests/test_kernels_gpu.py (pytest -m gpu) and tools/gpu_diag.py (prints every result
instead of stopping at the first failure).

Error metric everywhere: err = max|x - ref| / max|ref|  (SURVEY.md 7.3 H3).  Tolerances: fp16 storage / fp32 accumulate
-> 2e-3; bf16 -> 1.6e-2 (one rounding of the output dominates: 2^-11 / 2^-8 relative to the value, plus accumulation).
"""
import math
import os

import torch
import torch.nn.functional as F

TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


def relerr(x, ref):
    x, ref = x.float(), ref.float()
    if not torch.isfinite(x).all():
        return float("inf")
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()


def _r(*shape, dtype, dev, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape) * 7919 % 100003)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


# ------------------------------------------------------------------------------------------------ MFMA layout probe
def check_probe_mfma(dtype, dev):
    """A[32][16] . B[16][32] through one MFMA; operand/result lane layouts as documented in csrc/common.cuh."""
    from idm_vton_amd import ops
    A = _r(32, 16, dtype=dtype, dev="cpu", seed=1)
    Bm = _r(16, 32, dtype=dtype, dev="cpu", seed=2)       # deliberately not symmetric
    lane = torch.arange(64)
    ka = (8 * (lane // 32))[:, None] + torch.arange(8)[None, :]          # k index of element j in lane
    a_frag = A[(lane % 32)[:, None], ka]                                 # A[i = lane&31][k]
    b_frag = Bm[ka, (lane % 32)[:, None]]                                # B[k][j = lane&31]
    c = ops.probe_mfma(0 if dtype == torch.bfloat16 else 1, a_frag.contiguous().to(dev), b_frag.contiguous().to(dev)).cpu()
    ref = A.float() @ Bm.float()
    r = torch.arange(16)
    rows = (r % 4)[None, :] + 8 * (r // 4)[None, :] + 4 * (lane // 32)[:, None]   # row of reg r in lane
    got = torch.empty(32, 32)
    got[rows, (lane % 32)[:, None].expand(64, 16)] = c
    return relerr(got, ref)


def _hint(variant, bn, bm):
    return (variant << 28) | (bn << 16) | bm


# ------------------------------------------------------------------------------------------------ GEMM / conv
def check_linear(M, N, K, dtype, dev, bias=True, res=True, rowbias=False, tile_hint=0, seed=0):
    from idm_vton_amd import ops
    x = _r(M, K, dtype=dtype, dev=dev, seed=seed)
    w = _r(N, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=seed + 1)
    b = _r(N, dtype=dtype, dev=dev, seed=seed + 2) if bias else None
    rs = _r(M, N, dtype=dtype, dev=dev, seed=seed + 3) if res else None
    ref = x.float() @ w.float().t()
    if bias:
        ref = ref + b.float()
    kw = {}
    if rowbias:
        G = 4 if M % 4 == 0 else 1
        rb = _r(G, N, dtype=dtype, dev=dev, seed=seed + 4)
        ref = ref + rb.float().repeat_interleave(M // G, dim=0)
        kw = dict(rowbias=rb, rowbias_ld=N, rows_per_group=M // G)
    if res:
        ref = ref + rs.float()
    out = ops.linear(x, w, bias=b, res=rs, tile_hint=tile_hint, **kw)
    return relerr(out, ref)


def check_geglu(M, C, dtype, dev, seed=0, tile_hint=0):
    """GEGLU(x) = h * gelu(gate), [h | gate] = x W^T + b  (weights interleaved in 64-row blocks for the kernel)."""
    from idm_vton_amd import ops
    from idm_vton_amd.weights import interleave_geglu
    inner = 4 * C
    x = _r(M, C, dtype=dtype, dev=dev, seed=seed)
    w = _r(2 * inner, C, dtype=dtype, dev=dev, scale=C ** -0.5, seed=seed + 1)
    b = _r(2 * inner, dtype=dtype, dev=dev, seed=seed + 2)
    y = x.float() @ w.float().t() + b.float()
    h, g = y.chunk(2, dim=-1)
    ref = h * F.gelu(g)
    wi, bi = interleave_geglu(w, b)
    out = ops.linear(x, wi, bias=bi, geglu=True, tile_hint=tile_hint)
    return relerr(out, ref)


def check_vt(B, Ntok, C, dtype, dev, seed=0, tile_hint=0):
    """Fused QKV-style projection: columns [0,2C) normal, columns [2C,3C) written transposed as V^T[b][c][tok]."""
    from idm_vton_amd import ops
    M = B * Ntok
    x = _r(M, C, dtype=dtype, dev=dev, seed=seed)
    w = _r(3 * C, C, dtype=dtype, dev=dev, scale=C ** -0.5, seed=seed + 1)
    ref = x.float() @ w.float().t()
    out = torch.zeros(M, 2 * C, dtype=dtype, device=dev)
    vt = torch.zeros(B, C, Ntok, dtype=dtype, device=dev)
    vt_ref = ref[:, 2 * C:].reshape(B, Ntok, C).transpose(1, 2)
    ops.linear(x, w, out=out, vt=vt, vt_n0=2 * C, vt_tokens=Ntok, vt_perm=False, tile_hint=tile_hint)     # plain transpose
    e1 = relerr(out, ref[:, : 2 * C])
    e2 = relerr(vt, vt_ref)
    e3 = 0.0
    if Ntok % 16 == 0:                                   # attention key order (bits 2/3 of the token index swapped per 16)
        vt2 = torch.zeros(B, C, Ntok, dtype=dtype, device=dev)
        ops.linear(x, w, out=out, vt=vt2, vt_n0=2 * C, vt_tokens=Ntok, tile_hint=tile_hint)
        e3 = relerr(vt2, ops.key_order(vt_ref))
        if not torch.equal(ops.key_order(vt2), vt):      # same values, only the position differs
            e3 = float("inf")
    return max(e1, e2, e3)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def check_conv_ups_odd(B, Cin, Cout, H, W, Ho, Wo, dtype, dev, seed=0):
    """Upsample2D with `upsample_size` (F.interpolate(size=(Ho, Wo)) nearest, Ho in {2H - 1, 2H}) + 3x3 conv, fused in the gather."""
    from idm_vton_amd import ops
    from idm_vton_amd.weights import conv_weight_nhwc
    x = _r(B, H, W, Cin, dtype=dtype, dev=dev, seed=seed)
    w = _r(Cout, Cin, 3, 3, dtype=dtype, dev=dev, scale=(9 * Cin) ** -0.5, seed=seed + 1)
    b = _r(Cout, dtype=dtype, dev=dev, seed=seed + 2)
    up = F.interpolate(x.float().permute(0, 3, 1, 2), size=(Ho, Wo), mode="nearest")
    ref = F.conv2d(up, w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout)
    out = ops.gemm_conv(ops.conv_segs(x, 3, 1), conv_weight_nhwc(w), B * Ho * Wo, Ho=Ho, Wo=Wo, Hi=H, Wi=W, ups=True, bias=b)
    return relerr(out, ref)


def check_conv(B, Cin, Cout, H, W, dtype, dev, k=3, stride=1, ups=False, split=0, shortcut=0, temb=False, res=False,
               seed=0, tile_hint=0):
    """3x3 / 1x1 conv over NHWC with the fused extras of ResnetBlock2D:
    split>0  : input is cat([x1 (split ch), x2]) along C, never materialised (two pointers);
    shortcut : extra 1x1 conv of a second tensor (`shortcut` channels) fused as centre-tap K segments;
    temb     : + per-batch row vector; res: + residual; ups: nearest-2x upsample fused into the gather."""
    from idm_vton_amd import ops
    from idm_vton_amd.weights import conv_weight_nhwc
    pad = (k - 1) // 2
    x = _r(B, Cin, H, W, dtype=dtype, dev=dev, seed=seed)
    w = _r(Cout, Cin, k, k, dtype=dtype, dev=dev, scale=(Cin * k * k) ** -0.5, seed=seed + 1)
    bias = _r(Cout, dtype=dtype, dev=dev, seed=seed + 2)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), bias.float(), stride=stride, padding=pad)
    Ho, Wo = ref.shape[-2:]
    xn = _nhwc(x)
    if split:
        x1, x2 = xn[..., :split].contiguous(), xn[..., split:].contiguous()
        segs = []
        for ky in range(k):
            for kx in range(k):
                segs.append(ops.SegSpec(x1, 0, split, ky - pad, kx - pad))
                segs.append(ops.SegSpec(x2, 0, Cin - split, ky - pad, kx - pad))
    else:
        segs = ops.conv_segs(xn, k, pad)
    wk = conv_weight_nhwc(w)
    kw = {}
    if shortcut:
        xs = _r(B, shortcut, Ho, Wo, dtype=dtype, dev=dev, seed=seed + 5)
        ws = _r(Cout, shortcut, 1, 1, dtype=dtype, dev=dev, scale=shortcut ** -0.5, seed=seed + 6)
        ref = ref + F.conv2d(xs.float(), ws.float())
        xsn = _nhwc(xs)
        segs.append(ops.SegSpec(xsn, 0, shortcut, 0, 0))
        wk = torch.cat([wk, ws.reshape(Cout, shortcut)], dim=1).contiguous()
        assert stride == 1 and not ups
    M = B * Ho * Wo
    if temb:
        tb = _r(B, Cout, dtype=dtype, dev=dev, seed=seed + 3)
        ref = ref + tb.float()[:, :, None, None]
        kw.update(rowbias=tb, rowbias_ld=Cout, rows_per_group=Ho * Wo)
    if res:
        rs = _r(B, Cout, Ho, Wo, dtype=dtype, dev=dev, seed=seed + 4)
        ref = ref + rs.float()
        kw.update(res=_nhwc(rs).reshape(M, Cout))
    if len(segs) > 12:
        raise ValueError("too many segments for one launch")
    out = ops.gemm_conv(segs, wk, M, Ho=Ho, Wo=Wo, Hi=H, Wi=W, stride=stride, ups=ups, bias=bias, tile_hint=tile_hint, **kw)
    return relerr(out.reshape(B, Ho, Wo, Cout), _nhwc(ref))


# ------------------------------------------------------------------------------------------------ attention
def _key_order_padded(v, nk):
    """v [B][nk][C] -> (V^T [B][C][round16(nk)] in idmvton_attn_fwd's key order, padded keys zero; row length)."""
    from idm_vton_amd import ops
    ld = ops.round16(nk)
    vt = torch.zeros(v.shape[0], v.shape[2], ld, dtype=v.dtype, device=v.device)
    vt[:, :, :nk] = v.transpose(1, 2)
    return ops.key_order(vt), ld


def _run_attn(q, out, segs, heads, tune, prescaled, sp, kk, vv, ref):
    """Launch SELF attention on q (or, prescaled: on q * softmax_scale * log2e rounded to the storage dtype, against the reference
    of that rounded q) and return the max-rel error."""
    from idm_vton_amd import ops
    if prescaled:
        qs = (q.float() * ops.QSCALE).to(q.dtype)
        B, N, Cc = q.shape
        ref = F.scaled_dot_product_attention(sp(qs) / ops.QSCALE, kk, vv).transpose(1, 2).reshape(B, N, Cc)
        ops.attention(qs, out, segs, heads, tune=tune, q_prescaled=True)
    else:
        ops.attention(q, out, segs, heads, tune=tune)
    return relerr(out, ref)


def check_attn_self(B, heads, N, dtype, dev, n_garm=0, b0=0, scale=1.0, seed=0, tune=0, prescaled=False):
    """TryonNet attn1 semantics: keys = [own N tokens ; n_garm garment tokens]; batches < b0 see all-zero garment K/V."""
    from idm_vton_amd import ops
    Cc = heads * 64
    q = _r(B, N, Cc, dtype=dtype, dev=dev, scale=scale, seed=seed)
    k1 = _r(B, N, Cc, dtype=dtype, dev=dev, scale=scale, seed=seed + 1)
    v1 = _r(B, N, Cc, dtype=dtype, dev=dev, seed=seed + 2)
    sp = lambda t: t.float().view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    kk, vv = sp(k1), sp(v1)
    ko = lambda v, nk: _key_order_padded(v, nk)        # [B][nk][C] -> V^T [B][C][round16(nk)] in the kernel's key order
    vt1, ld1 = ko(v1, N)
    segs = [dict(k=k1, vt=vt1, nk=N, ldk=Cc, ldvt=ld1)]
    if n_garm:
        Bg = B - b0
        k2 = _r(Bg, n_garm, Cc, dtype=dtype, dev=dev, scale=scale, seed=seed + 3)
        v2 = _r(Bg, n_garm, Cc, dtype=dtype, dev=dev, seed=seed + 4)
        vt2, ld = ko(v2, n_garm)
        segs.append(dict(k=k2, vt=vt2, nk=n_garm, ldk=Cc, ldvt=ld, b0=b0))
        z = torch.zeros(b0, heads, n_garm, 64, device=dev)
        kk = torch.cat([kk, torch.cat([z, sp(k2)], dim=0)], dim=2)
        vv = torch.cat([vv, torch.cat([z, sp(v2)], dim=0)], dim=2)
    ref = F.scaled_dot_product_attention(sp(q), kk, vv).transpose(1, 2).reshape(B, N, Cc)
    out = torch.empty(B, N, Cc, dtype=dtype, device=dev)
    return _run_attn(q, out, segs, heads, tune, prescaled, sp, kk, vv, ref)


def check_f8_out_320_refused(dtype, dev):
    """The 12-wave 320x192 tile carries only the plain 16-byte epilogue (csrc/gemm_conv.hip): asked to write e4m3 q | k | V^T it must fail
    loudly or run the launch on another tile bit-exactly -- never write something else.  0.0 = one of the two happened."""
    try:
        return check_gemm_f8_out(dtype, dev, B=2, N=192, C=256, K=320, hint=_hint(6, 320, 192))
    except RuntimeError as e:
        assert "gemm_conv" in str(e), e
        return 0.0


def _e4m3(t):
    """fp32 -> OCP e4m3 bytes (uint8), saturating, by torch's own conversion (round to nearest even)."""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def _e4m3_to_f32(u8):
    return u8.view(torch.float8_e4m3fn).float()


def check_probe_mfma_f8(dev, scaled=False):
    """One v_mfma_scale_f32_32x32x64_f8f6f4: lane l supplies k-slots 32*(l>>5) .. +31 of row l & 31 (A) / column l & 31 (B), 32 e4m3
    bytes; result layout = the 32x32 f32 layout of every other MFMA.  scaled: E8M0 scale bytes 124 (A: 2^-3) and 128 (B: 2^1)."""
    from idm_vton_amd import ops
    g = torch.Generator().manual_seed(5)
    A = _e4m3_to_f32(_e4m3(torch.randn(32, 64, generator=g) * 2))           # values exactly representable in e4m3
    Bm = _e4m3_to_f32(_e4m3(torch.randn(64, 32, generator=g) * 2))          # deliberately not symmetric
    lane = torch.arange(64)
    ks = (32 * (lane // 32))[:, None] + torch.arange(32)[None, :]
    a_frag = _e4m3(A[(lane % 32)[:, None], ks]).contiguous()               # [64 lanes][32 bytes]
    b_frag = _e4m3(Bm[ks, (lane % 32)[:, None]]).contiguous()
    c = ops.probe_mfma(3 if scaled else 2, a_frag.to(dev), b_frag.to(dev)).cpu()
    ref = (A @ Bm) * (0.25 if scaled else 1.0)
    r = torch.arange(16)
    rows = (r % 4)[None, :] + 8 * (r // 4)[None, :] + 4 * (lane // 32)[:, None]
    got = torch.empty(32, 32)
    got[rows, (lane % 32)[:, None].expand(64, 16)] = c
    return relerr(got, ref)


def check_quant_f8(dtype, dev, seed=0):
    """idmvton_quant_f8 against torch's float8_e4m3fn conversion (bit-equal bytes): rows mode with a strided source, saturation at
    +-448, and the V^T re-order (16-bit key order -> fp8 slot order, zero fill of the last partial 64-key tile)."""
    from idm_vton_amd import ops
    x = _r(200, 256, dtype=dtype, dev=dev, scale=3.0, seed=seed)
    x[0, :4] = torch.tensor([1000.0, -1000.0, 448.0, 0.0], dtype=dtype, device=dev)
    src = x[:, 64:192]                                                       # strided view: lds = 256, cols = 128
    got = ops.quant_f8(src, 4.0)
    ok = torch.equal(got.cpu(), _e4m3(src.float().cpu() * 4.0))
    N = 208                                                                  # 3 full tiles + 16 keys
    v = _r(5, N, dtype=dtype, dev=dev, scale=1.5, seed=seed + 1)            # plain V^T rows [row][key]
    vt16 = ops.key_order(v)                                                  # what the 16-bit path stores
    got2 = ops.quant_f8(vt16, 2.0, mode=1).cpu()
    pos = torch.arange(256)
    key = 64 * (pos // 64) + 32 * ((pos // 16) % 2) + 8 * ((pos // 4) % 4) + 4 * ((pos // 32) % 2) + pos % 4
    vp = torch.zeros(5, 256)
    vp[:, key < N] = v.float().cpu()[:, key[key < N]]
    ok = ok and got2.shape == (5, 256) and torch.equal(got2, _e4m3(vp * 2.0))
    return 0.0 if ok else float("inf")


def check_gemm_f8_out(dtype, dev, B=2, N=192, C=128, K=320, hint=0, seed=0, fused_attn=False):
    """The QKV projection writing e4m3 operands itself (IDMVTON_IO_OUT_F8): q (softmax-scaled) | k into `out` bytes, V^T into the fp8
    kernel's slot order -- against the fp32 product of the same 16-bit operands, quantised by torch.  One e4m3 rounding of the kernel's fp32
    accumulator vs one of the reference's: they may land on neighbouring codes when the accumulators differ in their last bits, so the
    comparison is in value: |got - ref*scale| <= half an e4m3 ulp of the larger (2^-4 relative, 2^-10 in the subnormal range) + the
    accumulation-order slack.  fused_attn: also feed the bytes to idmvton_attn_f8 and return its error against the two-launch route
    (16-bit projection -> idmvton_quant_f8 -> idmvton_attn_f8) on the same inputs: both are e4m3 roundings of the same values."""
    from idm_vton_amd import ops
    M = B * N
    x = _r(M, K, dtype=dtype, dev=dev, scale=1.0, seed=seed)
    w = _r(3 * C, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=seed + 1)
    so, sv = 4.0, 2.0
    qk8 = torch.empty(M, 2 * C, dtype=torch.uint8, device=dev)
    vt8 = torch.empty(B, C, N, dtype=torch.uint8, device=dev)
    ops.linear(x, w, out=qk8, vt=vt8, vt_n0=2 * C, vt_tokens=N, colscale_n=C, colscale=ops.QSCALE, f8=(so, sv), tile_hint=hint)
    y = x.float() @ w.float().t()
    ref_o = torch.cat([y[:, :C] * ops.QSCALE, y[:, C:2 * C]], 1) * so
    got_o = qk8.view(torch.float8_e4m3fn).float()
    pos = torch.arange(N, device=dev)
    key = 64 * (pos // 64) + 32 * ((pos // 16) % 2) + 8 * ((pos // 4) % 4) + 4 * ((pos // 32) % 2) + pos % 4
    ref_v = (y[:, 2 * C:].view(B, N, C).transpose(1, 2) * sv)[:, :, key]            # [B][C][position]
    got_v = vt8.view(torch.float8_e4m3fn).float()
    worst = 0.0
    for got, ref in ((got_o, ref_o), (got_v, ref_v)):
        ref = ref.clamp(-448.0, 448.0)
        tol = ref.abs() * (2.0 ** -4 + 4e-3) + 2.0 ** -10 + 1e-3
        worst = max(worst, float(((got - ref).abs() / tol).max()))
    err = 0.0 if worst <= 1.0 else worst
    if fused_attn and err == 0.0:
        heads = C // 64
        a1 = torch.empty(M, C, dtype=dtype, device=dev)
        ops.attention_f8(qk8, a1, [dict(k8=qk8[:, C:], vt8=vt8, nk=N, ldk=2 * C, ldvt=N, k_rows=N)], heads, qk_scale_exp=-4, v_scale_exp=-1, B=B, Nq=N, ldq=2 * C, ldo=C)
        qk16 = torch.empty(M, 2 * C, dtype=dtype, device=dev)
        vt16 = torch.empty(B, C, N, dtype=dtype, device=dev)
        ops.linear(x, w, out=qk16, vt=vt16, vt_n0=2 * C, vt_tokens=N, colscale_n=C, colscale=ops.QSCALE)
        q8 = ops.quant_f8(qk16[:, :C], so)
        k8 = ops.quant_f8(qk16[:, C:], so)
        v8 = ops.quant_f8(vt16.view(B * C, N), sv, mode=1)
        a2 = torch.empty(M, C, dtype=dtype, device=dev)
        ops.attention_f8(q8, a2, [dict(k8=k8, vt8=v8, nk=N, ldk=C, ldvt=v8.shape[1], k_rows=N)], heads, qk_scale_exp=-4, v_scale_exp=-1, B=B, Nq=N, ldq=C, ldo=C)
        return relerr(a1, a2)
    return err


def check_gemm_f8_kv(dtype, dev, Bg=3, N=128, C=128, K=192, seed=3):
    """The garment K / V^T projection of the fp8 path (vt_n0 = C, no colscale) + the argument contract of IDMVTON_IO_OUT_F8."""
    from idm_vton_amd import ops
    g = _r(Bg * N, K, dtype=dtype, dev=dev, scale=1.0, seed=seed)
    w = _r(2 * C, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=seed + 1)
    k8 = torch.empty(Bg * N, C, dtype=torch.uint8, device=dev)
    vt8 = torch.empty(Bg, C, N, dtype=torch.uint8, device=dev)
    ops.linear(g, w, out=k8, vt=vt8, vt_n0=C, vt_tokens=N, f8=(4.0, 4.0))
    y = g.float() @ w.float().t() * 4.0
    pos = torch.arange(N, device=dev)
    key = 64 * (pos // 64) + 32 * ((pos // 16) % 2) + 8 * ((pos // 4) % 4) + 4 * ((pos // 32) % 2) + pos % 4
    worst = 0.0
    for got, ref in ((k8.view(torch.float8_e4m3fn).float(), y[:, :C]), (vt8.view(torch.float8_e4m3fn).float(), y[:, C:].view(Bg, N, C).transpose(1, 2)[:, :, key])):
        ref = ref.clamp(-448.0, 448.0)
        worst = max(worst, float(((got - ref).abs() / (ref.abs() * (2.0 ** -4 + 4e-3) + 2.0 ** -10 + 1e-3)).max()))
    bad = 0
    for kw in (dict(vt_tokens=96), dict(res=torch.zeros(Bg * N, C, dtype=dtype, device=dev)), dict(gelu=True)):   # 96 % 64 != 0; residual; activation
        try:
            vt_tokens = kw.pop("vt_tokens", N)
            ops.linear(g[:Bg * vt_tokens] if vt_tokens != N else g, w, out=k8, vt=vt8, vt_n0=C, vt_tokens=vt_tokens, f8=(4.0, 4.0), **kw)
            bad += 1
        except Exception:
            pass
    return 0.0 if worst <= 1.0 and bad == 0 else max(worst, float(bad))


def check_attn_f8(B, heads, N, dtype, dev, n_garm=0, b0=0, seed=0, eq=2, ek=2, ev=2):
    """fp8 self-attention (idmvton_attn_f8) with the TryonNet attn1 semantics of check_attn_self.  Returns (error against fp32 SDPA on
    the UNQUANTISED operands, error against fp32 SDPA on the dequantised e4m3 operands): the second isolates the kernel's own
    arithmetic (P in e4m3, fp32 accumulation) from the quantisation of q, k, v."""
    from idm_vton_amd import ops
    Cc = heads * 64
    q = _r(B, N, Cc, dtype=dtype, dev=dev, seed=seed)
    k1 = _r(B, N, Cc, dtype=dtype, dev=dev, seed=seed + 1)
    v1 = _r(B, N, Cc, dtype=dtype, dev=dev, seed=seed + 2)
    sp = lambda t: t.float().view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    qs = (q.float() * ops.QSCALE).to(dtype)                                  # what the QKV projection's colscale hands over
    q8 = ops.quant_f8(qs.view(B * N, Cc), 2.0 ** eq)
    deq = lambda u8, e: _e4m3_to_f32(u8.cpu()).to(dev) * 2.0 ** -e

    def seg(kx, vx, nk, b0_=0):
        Bx = kx.shape[0]
        k8 = ops.quant_f8(kx.view(Bx * nk, Cc), 2.0 ** ek)
        vt16, ld16 = _key_order_padded(vx, nk)                               # [Bx][C][round16(nk)] in the 16-bit key order
        vt8 = ops.quant_f8(vt16.view(Bx * Cc, ld16)[:, :ops.round16(nk)], 2.0 ** ev, mode=1)
        # dequantised operands in plain layout for the second reference
        kq = deq(k8, ek).view(Bx, nk, Cc)
        vq = _e4m3_to_f32(_e4m3(vx.float().cpu() * 2.0 ** ev)).to(dev) * 2.0 ** -ev
        return dict(k8=k8, vt8=vt8, nk=nk, ldk=Cc, ldvt=vt8.shape[1], b0=b0_), kq, vq
    s1, kq1, vq1 = seg(k1, v1, N)
    segs = [s1]
    kk, vv, kkq, vvq = sp(k1), sp(v1), sp(kq1), sp(vq1)
    if n_garm:
        Bg = B - b0
        k2 = _r(Bg, n_garm, Cc, dtype=dtype, dev=dev, seed=seed + 3)
        v2 = _r(Bg, n_garm, Cc, dtype=dtype, dev=dev, seed=seed + 4)
        s2, kq2, vq2 = seg(k2, v2, n_garm, b0)
        segs.append(s2)
        z = torch.zeros(b0, heads, n_garm, 64, device=dev)
        kk = torch.cat([kk, torch.cat([z, sp(k2)], dim=0)], dim=2)
        vv = torch.cat([vv, torch.cat([z, sp(v2)], dim=0)], dim=2)
        kkq = torch.cat([kkq, torch.cat([z, sp(kq2)], dim=0)], dim=2)
        vvq = torch.cat([vvq, torch.cat([z, sp(vq2)], dim=0)], dim=2)
    ref = F.scaled_dot_product_attention(sp(q), kk, vv).transpose(1, 2).reshape(B, N, Cc)
    qq = sp(deq(q8, eq).view(B, N, Cc)) / ops.QSCALE                         # dequantised q, back in SDPA's units
    refq = F.scaled_dot_product_attention(qq, kkq, vvq).transpose(1, 2).reshape(B, N, Cc)
    out = torch.empty(B, N, Cc, dtype=dtype, device=dev)
    ops.attention_f8(q8, out, segs, heads, qk_scale_exp=-(eq + ek), v_scale_exp=-ev, B=B, Nq=N, ldq=Cc, ldo=Cc)
    return relerr(out, ref), relerr(out, refq)


def pp_tune(stages, deep, pair=0, noprio=0, thr=0):
    """idmvton_attn_args.tune for the ping-pong kernel: 8 waves; deep=0 -> 128 registers, two workgroups per CU (2 LDS stages);
    deep=1 -> one workgroup per CU, `stages` LDS stages, all fragments prefetched, max subtraction on the matrix pipe (needs a
    pre-multiplied q, else the deep=0 build runs); pair=1 pairs waves (w, w^1); thr selects the deferred-rescale threshold
    {0: 4, 1: 0, 2: 8, 3: 2}."""
    return ((thr << 2 | noprio << 1 | pair) << 24) | ((3 if deep else 2) << 16) | (stages << 8) | 8


def check_attn_spike(dtype, dev, tune=0, seed=0, prescaled=False):
    """Deferred-rescale branch (the running max is only moved when it grows by more than a threshold): moderate logits for most
    keys, then a few keys far down the walk (own segment tile 9 and the garment segment) whose logit towers over a subset of
    rows, so the branch fires late, for some rows only, with O / l already accumulated.  Full-tensor fp32 reference."""
    from idm_vton_amd import ops
    B, heads, N = 2, 2, 768
    Cc = heads * 64
    g = torch.Generator(device="cpu").manual_seed(1234 + seed)
    q = torch.randn(B, N, Cc, generator=g)
    k1 = torch.randn(B, N, Cc, generator=g) * 0.5
    k2 = torch.randn(B, N, Cc, generator=g) * 0.5
    v1, v2 = torch.randn(B, N, Cc, generator=g), torch.randn(B, N, Cc, generator=g)
    for (kk, key, rows, gain) in ((k1, 9 * 64 + 5, slice(0, N, 3), 3.0), (k2, 4 * 64 + 37, slice(1, N, 5), 5.0), (k2, 11 * 64 + 63, slice(2, N, 7), 8.0)):
        # key `key` points along the mean direction of the chosen query rows: q.k >> every other score for those rows
        d = q[:, rows].mean(dim=1)
        kk[:, key] = gain * d / d.norm(dim=-1, keepdim=True).clamp_min(1e-6) * 4.0
    q, k1, k2, v1, v2 = (t.to(dtype).to(dev) for t in (q, k1, k2, v1, v2))
    sp = lambda t: t.float().view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), torch.cat([sp(k1), sp(k2)], dim=2), torch.cat([sp(v1), sp(v2)], dim=2))
    ref = ref.transpose(1, 2).reshape(B, N, Cc)
    vt1, _ = _key_order_padded(v1, N)
    vt2, _ = _key_order_padded(v2, N)
    out = torch.empty(B, N, Cc, dtype=dtype, device=dev)
    segs = [dict(k=k1, vt=vt1, nk=N, ldk=Cc, ldvt=N), dict(k=k2, vt=vt2, nk=N, ldk=Cc, ldvt=N)]
    return _run_attn(q, out, segs, heads, tune, prescaled, sp, torch.cat([sp(k1), sp(k2)], dim=2), torch.cat([sp(v1), sp(v2)], dim=2), ref)


def check_attn_neg(dtype, dev, tune=0, prescaled=False):
    """Every logit strongly negative (q = -k direction): the running max must follow the data down, not stay at its start value."""
    from idm_vton_amd import ops
    B, heads, N = 1, 2, 192
    Cc = heads * 64
    g = torch.Generator(device="cpu").manual_seed(77)
    base = torch.randn(B, 1, Cc, generator=g)
    q = (base * 6 + 0.3 * torch.randn(B, N, Cc, generator=g)).to(dtype).to(dev)
    k = (-base * 6 + 0.3 * torch.randn(B, N, Cc, generator=g)).to(dtype).to(dev)       # q.k ~ -36 * 64 / 8 = -290 after scaling
    v = torch.randn(B, N, Cc, generator=g).to(dtype).to(dev)
    sp = lambda t: t.float().view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, Cc)
    vt, ld = _key_order_padded(v, N)
    out = torch.empty(B, N, Cc, dtype=dtype, device=dev)
    return _run_attn(q, out, [dict(k=k, vt=vt, nk=N, ldk=Cc, ldvt=ld)], heads, tune, prescaled, sp, sp(k), sp(v), ref)


def check_colscale(dtype, dev):
    """Column scaling in the GEMM epilogue: (x W^T + b) * s for the first n columns, before the residual."""
    from idm_vton_amd import ops
    M, N, K = 300, 192, 128
    x, w, b, rs = _r(M, K, dtype=dtype, dev=dev), _r(N, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=1), _r(N, dtype=dtype, dev=dev, seed=2), _r(M, N, dtype=dtype, dev=dev, seed=3)
    ref = x.float() @ w.float().t() + b.float()
    ref[:, :64] *= ops.QSCALE
    ref = ref + rs.float()
    out = ops.linear(x, w, bias=b, res=rs, colscale_n=64, colscale=ops.QSCALE)
    return relerr(out, ref)


def check_attn_cross(B, heads, N, dtype, dev, n_text=77, n_ip=16, ip_scale=1.0, seed=0, tune=0):
    """IPAttnProcessor2_0 semantics: SDPA over text keys + ip_scale * SDPA over image keys."""
    from idm_vton_amd import ops
    Cc = heads * 64
    q = _r(B, N, Cc, dtype=dtype, dev=dev, seed=seed)
    sp = lambda t: t.float().view(t.shape[0], t.shape[1], heads, 64).transpose(1, 2)
    segs, ref = [], 0
    for i, (nk, sc) in enumerate(((n_text, 1.0), (n_ip, ip_scale))):
        rows = (nk + 15) // 16 * 16
        k = torch.zeros(B, rows, Cc, dtype=dtype, device=dev)
        k[:, :nk] = _r(B, nk, Cc, dtype=dtype, dev=dev, seed=seed + 10 + i)
        v = _r(B, nk, Cc, dtype=dtype, dev=dev, seed=seed + 20 + i)
        vt, _ = _key_order_padded(v, nk)
        segs.append(dict(k=k, vt=vt, nk=nk, ldk=Cc, ldvt=rows, k_rows=rows))
        ref = ref + sc * F.scaled_dot_product_attention(sp(q), sp(k[:, :nk]), sp(v))
    ref = ref.transpose(1, 2).reshape(B, N, Cc)
    out = torch.empty(B, N, Cc, dtype=dtype, device=dev)
    from idm_vton_amd import ffi
    ops.attention(q, out, segs, heads, mode=ffi.ATTN_CROSS, ip_scale=ip_scale, tune=tune)
    return relerr(out, ref)


def check_xattn_fused(B, heads, N, K, dtype, dev, n_text=77, n_ip=16, ip_scale=1.0, tile_hint=0, seed=0):
    """attn2.to_q with the cross-attention as its epilogue (IDMVTON_EPI_XATTN) against fp32: q = x Wq^T, softmax(q k_t^T / 8) v_t + ip_scale *
    softmax(q k_i^T / 8) v_i per head; and against the two-launch form it replaces (same roundings: equal up to summation order)."""
    from idm_vton_amd import ffi, ops
    C = heads * 64
    M = B * N
    x = _r(M, K, dtype=dtype, dev=dev, seed=seed)
    wq = _r(C, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=seed + 1)
    segs, refs = [], []
    q = (x.float() @ wq.float().t()).to(dtype).float().view(B, N, heads, 64).transpose(1, 2)          # q is rounded once to the storage type
    out_ref = torch.zeros(B, heads, N, 64, device=dev)
    for i, nk in enumerate([n_text] + ([n_ip] if n_ip else [])):
        rows = (nk + 31) // 32 * 32
        k = torch.zeros(B, rows, C, dtype=dtype, device=dev)
        k[:, :nk] = _r(B, nk, C, dtype=dtype, dev=dev, seed=seed + 2 + i)
        v = _r(B, nk, C, dtype=dtype, dev=dev, seed=seed + 5 + i)
        vt = torch.zeros(B, C, rows, dtype=dtype, device=dev)
        vt[:, :, :nk] = v.transpose(1, 2)
        segs.append(dict(k=k, vt=ops.key_order(vt), nk=nk, ldk=C, ldvt=rows, k_rows=rows))
        kk = k[:, :nk].float().view(B, nk, heads, 64).transpose(1, 2)
        vv = v.float().view(B, nk, heads, 64).transpose(1, 2)
        out_ref += (ip_scale if i == 1 else 1.0) * F.scaled_dot_product_attention(q, kk, vv)
    ref = out_ref.transpose(1, 2).reshape(M, C)
    fused = ops.linear(x, ops.xattn_q_weight(wq), xattn=dict(segs=segs, tokens=N, ip_scale=ip_scale), tile_hint=tile_hint)
    q2 = ops.linear(x, wq)
    two = torch.empty(M, C, dtype=dtype, device=dev)
    if n_ip:
        ops.attention(q2, two, segs, heads, mode=ffi.ATTN_CROSS, ip_scale=ip_scale, B=B, Nq=N, ldq=C, ldo=C)
    else:
        ops.attention(q2, two, segs, heads, B=B, Nq=N, ldq=C, ldo=C)
    return max(relerr(fused, ref), relerr(fused, two.float()))


def check_attn_small(B, heads, L, d, dtype, dev, causal, Lq=None, seed=0, scale=1.0):
    """idmvton_attn_small (CLIP towers): fused-QKV layout, any even head_dim <= 128, optional causal mask."""
    from idm_vton_amd import ops
    H, Lq = heads * d, (L if Lq is None else Lq)
    qkv = _r(B, L, 3 * H, dtype=dtype, dev=dev, seed=seed, scale=scale)
    q, k, v = qkv[:, L - Lq:, :H].contiguous(), qkv[:, :, H:2 * H], qkv[:, :, 2 * H:]
    sp = lambda t: t.float().reshape(B, t.shape[1], heads, d).transpose(1, 2)
    mask = None
    if causal:
        mask = torch.ones(Lq, L, dtype=torch.bool, device=dev).tril(diagonal=L - Lq)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=mask).transpose(1, 2).reshape(B, Lq, H)
    out = torch.empty(B, Lq, H, dtype=dtype, device=dev)
    ops.attention_small(q, k, v, out, heads, d, scale=d ** -0.5, causal=causal, B=B, Lq=Lq, Lk=L, ldq=H, ldk=3 * H, ldv=3 * H, ldo=H)
    return relerr(out, ref)


def check_quickgelu(dtype, dev):
    """quick_gelu epilogue (CLIP-L MLP): (x W^T + b) * sigmoid(1.702 (x W^T + b)), then + residual."""
    from idm_vton_amd import ops
    M, N, K = 154, 3072, 768
    x, w, b = _r(M, K, dtype=dtype, dev=dev), _r(N, K, dtype=dtype, dev=dev, scale=2 * K ** -0.5, seed=1), _r(N, dtype=dtype, dev=dev, seed=2)
    pre = x.float() @ w.float().t() + b.float()
    return relerr(ops.linear(x, w, bias=b, quick_gelu=True), pre * torch.sigmoid(1.702 * pre))


# ------------------------------------------------------------------------------------------------ norms / elementwise
def check_layernorm(rows, Cc, dtype, dev, seed=0):
    from idm_vton_amd import ops
    x = _r(rows, Cc, dtype=dtype, dev=dev, scale=3.0, seed=seed) + 1.5
    g = _r(Cc, dtype=dtype, dev=dev, seed=seed + 1)
    b = _r(Cc, dtype=dtype, dev=dev, seed=seed + 2)
    ref = F.layer_norm(x.float(), (Cc,), g.float(), b.float(), 1e-5)
    o2 = torch.empty_like(x)
    out = ops.layernorm(x, g, b, 1e-5, out2=o2)
    return max(relerr(out, ref), relerr(o2, ref))


def check_stream_f32(M, N, K, dtype, dev, tile_hint=0, seed=0):
    """The fp32 residual stream (gemm_conv io_flags, layernorm x_f32): fp32 res in / fp32 out, fp32 res in / 16-bit out, 16-bit
    res in / fp32 out, each against the fp32 reference with an fp32-ONLY tolerance where the output is fp32 (the only roundings left
    are the 16-bit operands, which the reference shares), then LayerNorm of the fp32 result."""
    from idm_vton_amd import ops
    x = _r(M, K, dtype=dtype, dev=dev, seed=seed)
    w = _r(N, K, dtype=dtype, dev=dev, scale=K ** -0.5, seed=seed + 1)
    b = _r(N, dtype=dtype, dev=dev, seed=seed + 2)
    r32 = _r(M, N, dtype=torch.float32, dev=dev, scale=3.0, seed=seed + 3)               # NOT representable in 16 bits
    ref = x.float() @ w.float().t() + b.float() + r32
    o32 = ops.linear(x, w, bias=b, res=r32, out_f32=True, tile_hint=tile_hint)
    assert o32.dtype == torch.float32
    e = relerr(o32, ref) / 2e-5 * TOL[dtype]                                              # fp32 in, fp32 out: <= 2e-5, scaled to the caller's tolerance
    o16 = ops.linear(x, w, bias=b, res=r32, tile_hint=tile_hint)
    assert o16.dtype == dtype
    e = max(e, relerr(o16, ref))
    r16 = r32.to(dtype)
    o32b = ops.linear(x, w, bias=b, res=r16, out_f32=True, tile_hint=tile_hint)
    e = max(e, relerr(o32b, x.float() @ w.float().t() + b.float() + r16.float()) / 2e-5 * TOL[dtype])
    g = _r(N, dtype=dtype, dev=dev, seed=seed + 4)
    bt = _r(N, dtype=dtype, dev=dev, seed=seed + 5)
    if N <= 2048:
        y = ops.layernorm(o32, g, bt, 1e-5)
        assert y.dtype == dtype
        e = max(e, relerr(y, F.layer_norm(o32, (N,), g.float(), bt.float(), 1e-5)))
    return e


def check_groupnorm(B, HW, Cc, dtype, dev, groups=32, silu=True, split=0, eps=1e-5, seed=0):
    from idm_vton_amd import ops
    x = _r(B, HW, Cc, dtype=dtype, dev=dev, scale=2.0, seed=seed) + 0.7
    g = _r(Cc, dtype=dtype, dev=dev, seed=seed + 1)
    b = _r(Cc, dtype=dtype, dev=dev, seed=seed + 2)
    ref = F.group_norm(x.float().transpose(1, 2), groups, g.float(), b.float(), eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    stats = torch.empty(ops.gn_stats_doubles(B, HW, Cc, groups), dtype=torch.float64, device=dev)
    if split:
        out = ops.groupnorm(x[..., :split].contiguous(), g, b, groups, eps, silu, stats, x2=x[..., split:].contiguous())
    else:
        out = ops.groupnorm(x, g, b, groups, eps, silu, stats)
    return relerr(out, ref)


def check_groupnorm_reproducible(B, HW, Cc, dtype, dev, groups=32):
    """GroupNorm statistics are reduced without atomics: two runs on the same input give identical bits (returns 0.0)."""
    from idm_vton_amd import ops
    x = _r(B, HW, Cc, dtype=dtype, dev=dev, scale=2.0, seed=11) + 0.7
    g = _r(Cc, dtype=dtype, dev=dev, seed=12)
    b = _r(Cc, dtype=dtype, dev=dev, seed=13)
    stats = torch.full((ops.gn_stats_doubles(B, HW, Cc, groups),), float("nan"), dtype=torch.float64, device=dev)   # scratch needs no init
    outs = [ops.groupnorm(x, g, b, groups, 1e-5, True, stats).clone() for _ in range(3)]
    return 0.0 if all(torch.equal(outs[0], o) for o in outs[1:]) else 1.0


# ------------------------------------------------------------------------------------------------ split-precision (fp32-equivalent) VAE path
def _relerr64(x, ref):
    x, ref = x.detach().double().cpu(), ref.detach().double().cpu()
    if not torch.isfinite(x).all():
        return float("inf")
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-300)).item()


def check_split(rows, cols, dev, mode, seed=0):
    """idmvton_split: hi = bf16(x), lo = bf16(x - hi) bit for bit, in the layout of each mode; hi + lo reproduces x to 2^-16."""
    from idm_vton_amd import ffi, ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.randn(rows, cols, generator=g) * torch.exp(4 * torch.randn(rows, 1, generator=g))).to(dev)      # rows spread over ~10 binades
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    out = ops.split(x, torch.bfloat16, mode)
    if mode == ffi.SPLIT_ACT:
        ref = torch.cat([hi, lo], 1)
    elif mode == ffi.SPLIT_W3:
        ref = torch.cat([hi, hi, lo], 1)
    else:
        ref = torch.cat([hi.t(), hi.t(), lo.t()], 1).contiguous()
    exact = 0.0 if torch.equal(out.view(torch.int16), ref.view(torch.int16)) else 1.0
    return max(exact, _relerr64(hi.double() + lo.double(), x) / 256.0)            # second term: <= 2^-16 -> contributes <= 6e-8


def check_gn_precise(B, HW, Cc, dev, groups=32, silu=True, seed=0):
    """GroupNorm in the split-precision form: fp32 in, fp32 affine, [hi | lo] out; hi + lo against an fp64 reference."""
    from idm_vton_amd import ops
    x = _r(B, HW, Cc, dtype=torch.float32, dev=dev, scale=2.0, seed=seed) * 37.0 + 11.0
    g = _r(Cc, dtype=torch.float32, dev=dev, seed=seed + 1)
    b = _r(Cc, dtype=torch.float32, dev=dev, seed=seed + 2)
    ref = F.group_norm(x.double().transpose(1, 2), groups, g.double(), b.double(), 1e-6).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    stats = torch.empty(ops.gn_stats_doubles(B, HW, Cc, groups), dtype=torch.float64, device=dev)
    out = ops.groupnorm(x, g, b, groups, 1e-6, silu, stats, split_dtype=torch.bfloat16)
    assert out.shape == (B, HW, 2 * Cc)
    return _relerr64(out[..., :Cc].double() + out[..., Cc:].double(), ref)


def check_softmax_split(rows, n, dev, seed=0, n_valid=0):
    from idm_vton_amd import ops
    x = _r(rows, n, dtype=torch.float32, dev=dev, scale=3.0, seed=seed)
    keep = x.clone()
    out = ops.softmax_rows_split(x, 0.7, torch.bfloat16, n_valid=n_valid)
    nv = n_valid or n
    ref = torch.zeros(rows, n, dtype=torch.float64, device=dev)
    ref[:, :nv] = torch.softmax(0.7 * x[:, :nv].double(), dim=-1)
    assert torch.equal(x, keep)
    tail = float(out[:, nv:n].abs().max() + out[:, n + nv:].abs().max()) if nv < n else 0.0      # padded columns are exact zeros
    return max(_relerr64(out[:, :n].double() + out[:, n:].double(), ref), tail)


def check_softmax_rows(rows, n, dtype, dev, n_valid=0, seed=0):
    """In-place row softmax (VAE mid-block attention, 16-bit path); n_valid: the padded-key form."""
    from idm_vton_amd import ops
    x = _r(rows, n, dtype=dtype, dev=dev, scale=3.0, seed=seed)
    nv = n_valid or n
    ref = torch.zeros(rows, n, dtype=torch.float32, device=dev)
    ref[:, :nv] = torch.softmax(0.7 * x[:, :nv].float(), dim=-1)
    ops.softmax_rows(x, 0.7, n_valid=n_valid)
    tail = float(x[:, nv:].float().abs().max()) if nv < n else 0.0
    return max(relerr(x, ref), tail)


def check_plin(M, N, K, dev, exact_w=False, res=True, seed=0):
    """A Linear through the split-precision path (vae._PConv: [hi | lo] activations x [w_hi | w_hi][w_lo] weights, fp32 bias / residual /
    output) against fp64: fp32-equivalent (<= 2e-5 of the output range), where the 16-bit path gives 2e-3 / 1.6e-2."""
    from idm_vton_amd import ops
    from idm_vton_amd.vae import _PConv
    x = _r(M, K, dtype=torch.float32, dev=dev, seed=seed)
    w = _r(N, K, dtype=torch.float32, dev=dev, scale=K ** -0.5, seed=seed + 1)
    if exact_w:
        w = w.to(torch.bfloat16).float()
    b = _r(N, dtype=torch.float32, dev=dev, seed=seed + 2)
    rs = _r(M, N, dtype=torch.float32, dev=dev, seed=seed + 3) if res else None
    cv = _PConv(w, b)
    assert cv.three == (not exact_w)
    xp = ops.split(x, torch.bfloat16)
    out = ops.gemm_conv(cv.segs(xp, 0), cv.w, M, bias=cv.b, res=rs, out_f32=True)
    ref = x.double() @ w.double().t() + b.double() + (rs.double() if res else 0.0)
    return _relerr64(out, ref)


def check_pconv(B, Cin, Cout, H, W, dev, ups=False, shortcut=0, exact_w=False, seed=0):
    """3x3 conv (optionally fused nearest-2x, optionally a fused 1x1 shortcut on a second input) through the split-precision path vs fp64."""
    from idm_vton_amd import ops
    from idm_vton_amd.vae import _PConv
    x = _r(B, H, W, Cin, dtype=torch.float32, dev=dev, seed=seed)
    w = _r(Cout, Cin, 3, 3, dtype=torch.float32, dev=dev, scale=(9 * Cin) ** -0.5, seed=seed + 1)
    b = _r(Cout, dtype=torch.float32, dev=dev, seed=seed + 2)
    q = (lambda t: t.to(torch.bfloat16).float()) if exact_w else (lambda t: t)
    w = q(w)
    xin = x.double().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.double(), b.double(), padding=1)
    sc, xs = None, None
    if shortcut:
        x2 = _r(B, H, W, shortcut, dtype=torch.float32, dev=dev, seed=seed + 3)
        ws = q(_r(Cout, shortcut, 1, 1, dtype=torch.float32, dev=dev, scale=shortcut ** -0.5, seed=seed + 4))
        bs = _r(Cout, dtype=torch.float32, dev=dev, seed=seed + 5)
        ref = ref + F.conv2d(x2.double().permute(0, 3, 1, 2), ws.double(), bs.double())
        sc = (ws, bs)
        xs = ops.split(x2.reshape(-1, shortcut), torch.bfloat16).view(B, H * W, 2 * shortcut)
    cv = _PConv(w, b, shortcut=sc)
    xp = ops.split(x.reshape(-1, Cin), torch.bfloat16).view(B, H * W, 2 * Cin)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    out = ops.gemm_conv(cv.segs(xp, 1, xs), cv.w, B * Ho * Wo, Ho=Ho, Wo=Wo, Hi=H, Wi=W, ups=ups, bias=cv.b, out_f32=True)
    return _relerr64(out.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref)


def check_layout_split(B, h, w, dev, seed=0):
    from idm_vton_amd import ops
    src = _r(B, 4, h, w, dtype=torch.float32, dev=dev, seed=seed) * 5.0
    n = ops.to_nhwc(src, torch.bfloat16, cpad=64, scale=2.0, shift=-1.0, split=True)                     # [B][hw][128]
    v = (src * 2.0 - 1.0).permute(0, 2, 3, 1).reshape(B, h * w, 4)
    e1 = _relerr64(n[..., :4].double() + n[..., 64:68].double(), v)
    e2 = float(n[..., 4:64].abs().max() + n[..., 68:].abs().max())
    img = _r(B, h * w, 8, dtype=torch.float32, dev=dev, seed=seed + 1)
    back = ops.to_nchw(img, 3, (h, w), scale=0.5, shift=0.5)
    e3 = _relerr64(back, (img[..., :3] * 0.5 + 0.5).reshape(B, h, w, 3).permute(0, 3, 1, 2))
    return max(e1, e2, e3)


def check_elementwise(B, h, w, dtype, dev, seed=0):
    from idm_vton_amd import ops
    hw = h * w
    lat = _r(B, 4, h, w, dtype=torch.float32, dev=dev, seed=seed)
    cond = _r(2 * B, hw, 9, dtype=dtype, dev=dev, seed=seed + 1)
    out = torch.full((2 * B, hw, 64), 7.0, dtype=dtype, device=dev)
    ops.pack_input(lat, cond, out)
    ln = lat.permute(0, 2, 3, 1).reshape(B, hw, 4).to(dtype)
    ref = torch.cat([torch.cat([ln, ln], 0), cond, torch.zeros(2 * B, hw, 51, dtype=dtype, device=dev)], dim=-1)
    e1 = (out.float() - ref.float()).abs().max().item()
    eps = _r(2 * B, hw, 64, dtype=dtype, dev=dev, seed=seed + 2)
    noise = _r(B, 4, h, w, dtype=torch.float32, dev=dev, seed=seed + 3)
    coef = torch.tensor([0.98, -0.03, 0.1, 2.0], dtype=torch.float32, device=dev)
    e = eps[..., :4].float().reshape(2, B, h, w, 4).permute(0, 1, 4, 2, 3)
    e = e[0] + 2.0 * (e[1] - e[0])
    ref2 = 0.98 * lat - 0.03 * e + 0.1 * noise
    lat2 = lat.clone()
    ops.cfg_step(eps, lat2, noise, coef)
    e2 = relerr(lat2, ref2)
    src = _r(B, 3, h, w, dtype=torch.float32, dev=dev, seed=seed + 4)
    n = ops.to_nhwc(src, dtype, cpad=64, scale=2.0, shift=-1.0)
    back = ops.to_nchw(n, 3, (h, w), scale=0.5, shift=0.5)
    e3 = (back - (src * 2 - 1).to(dtype).float() * 0.5 - 0.5).abs().max().item()
    mom = _r(B, hw, 8, dtype=dtype, dev=dev, seed=seed + 5)
    z = ops.vae_sample(mom, noise, 0.13025)
    mm = mom.float().reshape(B, h, w, 8).permute(0, 3, 1, 2)
    refz = (mm[:, :4] + torch.exp(0.5 * mm[:, 4:].clamp(-30, 20)) * noise) * 0.13025
    e4 = relerr(z, refz)
    return max(e1, e2, e3, e4)


RING_TILES = ((_hint(5, 256, 256), "h5f0"), (_hint(5, 256, 257), "h5f1"),
              # bit 14 of the BM field: 5 persistent workgroups, so these small shapes walk several output tiles per workgroup (cross-tile prefetch)
              (_hint(5, 256, 257 | 0x4000), "h5f1_walk"), (_hint(5, 256, 256 | 0x4000), "h5f0_walk"), (_hint(5, 256, 192 | 0x4000), "h192_walk"),
              (_hint(2, 128, 256), "p128x256"), (_hint(2, 64, 64), "p64x64"),
              # variant 6: 8-wave 128x128 (2 forms), the 12-wave 320x192 / 256x192 tiles, 16-wave 256x256 / 128x256
              (_hint(6, 128, 128), "w8_128x128"), (_hint(6, 128, 129), "w8p_128x128"), (_hint(6, 320, 192), "w12_320x192"), (_hint(6, 256, 192), "w12_256x192"),
              (_hint(6, 256, 256), "w16_256x256"), (_hint(6, 128, 256), "w16_128x256"),
              (_hint(1, 256, 256), "r256x256"), (_hint(1, 128, 256), "r128x256"), (_hint(1, 128, 128), "r128x128"), (_hint(1, 128, 64), "r128x64"),
              (_hint(1, 64, 64), "r64x64"))


# tile families run with IDMVTON_IO_OUT_F8 (gemm_f8_out_* below)
F8_OUT_TILES = (("auto", 0), ("r128x128", _hint(1, 128, 128)), ("r128x256", _hint(1, 128, 256)), ("r256x256", _hint(1, 256, 256)),
               ("p64x64", _hint(2, 64, 64)), ("w8_128x128", _hint(6, 128, 128)), ("v0_128x128", _hint(0, 128, 128)), ("h256", _hint(5, 256, 256)), ("h256f1", _hint(5, 256, 257)),
               ("h192", _hint(5, 256, 192)),
               # ... and every other (variant, BN, BM) of idm-vton_amd/tune_gfx950.json: ops.load_tune mirrors the bf16-measured entries onto the
               # fp16 keys and gemm_key has no io_flags, so IDMVTON_IO_OUT_F8 launches of the fp16+fp8 engine select these tiles too
               ("w8p_128x128", _hint(6, 128, 129)), ("w12_256x192", _hint(6, 256, 192)),
               # (the 320-column tile has no e4m3 / V^T epilogue: the library refuses or re-routes such a launch, see gemm_f8_out_320_tile_is_refused)
               ("w16_256x256", _hint(6, 256, 256)), ("w16_128x256", _hint(6, 128, 256)), ("p128x256", _hint(2, 128, 256)),
               ("v0_64x64", _hint(0, 64, 64)), ("r64x64", _hint(1, 64, 64)))


def all_checks(dev="cuda"):
    """(name, thunk, tolerance) for every kernel-level check; sizes are the reference's real shapes where cheap."""
    out = []
    for dt in (torch.float16, torch.bfloat16):
        n = "f16" if dt == torch.float16 else "bf16"
        tol = TOL[dt]
        add = lambda name, fn, t=tol: out.append((f"{name}[{n}]", fn, t))
        add("probe_mfma", lambda dt=dt: check_probe_mfma(dt, dev), 1e-6 if dt == torch.float16 else 1e-6)
        if dt == torch.float16:
            # a wrong operand layout gives O(1) errors; 1.8e-5 measured = the instruction's internal summation of 64 products
            add("probe_mfma_scale_f8_32x32x64", lambda: check_probe_mfma_f8(dev), 1e-4)
            add("probe_mfma_scale_f8_32x32x64_scales", lambda: check_probe_mfma_f8(dev, True), 1e-4)
        for hint, tag in ((0, "auto"), ((128 << 16) | 128, "128x128"), ((128 << 16) | 64, "128x64"), ((64 << 16) | 64, "64x64")):
            add(f"linear_768x640x640_{tag}", lambda dt=dt, hint=hint: check_linear(768, 640, 640, dt, dev, tile_hint=hint))
        # LDS-ring variants (tile_hint variant 1): every epilogue / gather mode on every ring tile, incl. M/N tails, K = 1
        # and 2 tiles (shorter than the ring), and tile counts that are not multiples of the raster group
        for hint, tag in RING_TILES:
            add(f"ring_linear_768x640x640_{tag}", lambda dt=dt, hint=hint: check_linear(768, 640, 640, dt, dev, tile_hint=hint))
            add(f"ring_linear_ragged_1000x328x192_{tag}", lambda dt=dt, hint=hint: check_linear(1000, 328, 192, dt, dev, rowbias=True, tile_hint=hint))
            add(f"ring_linear_K64_{tag}", lambda dt=dt, hint=hint: check_linear(300, 192, 64, dt, dev, tile_hint=hint))
            add(f"ring_linear_K128_{tag}", lambda dt=dt, hint=hint: check_linear(2500, 136, 128, dt, dev, tile_hint=hint))
            add(f"ring_linear_K192_{tag}", lambda dt=dt, hint=hint: check_linear(520, 264, 192, dt, dev, tile_hint=hint))
            add(f"ring_linear_K320_{tag}", lambda dt=dt, hint=hint: check_linear(515, 520, 320, dt, dev, rowbias=True, tile_hint=hint))
            add(f"ring_linear_3072x1280x1280_{tag}", lambda dt=dt, hint=hint: check_linear(3072, 1280, 1280, dt, dev, tile_hint=hint))
            add(f"ring_vt_B2_N768_C640_{tag}", lambda dt=dt, hint=hint: check_vt(2, 768, 640, dt, dev, tile_hint=hint))
            add(f"ring_conv3x3_320_32x24_{tag}", lambda dt=dt, hint=hint: check_conv(2, 320, 320, 32, 24, dt, dev, temb=True, tile_hint=hint))
            add(f"ring_conv3x3_s2_{tag}", lambda dt=dt, hint=hint: check_conv(2, 128, 192, 17, 13, dt, dev, stride=2, tile_hint=hint))
            add(f"ring_conv3x3_ups_{tag}", lambda dt=dt, hint=hint: check_conv(2, 128, 128, 9, 7, dt, dev, ups=True, tile_hint=hint))
            add(f"ring_conv1x1_split_{tag}", lambda dt=dt, hint=hint: check_conv(2, 320, 64, 12, 10, dt, dev, k=1, split=192, tile_hint=hint))
            add(f"ring_conv3x3_shortcut_{tag}", lambda dt=dt, hint=hint: check_conv(2, 128, 128, 16, 12, dt, dev, shortcut=192, temb=True, tile_hint=hint))
            if (hint >> 16) & 0xfff >= 128:
                add(f"ring_geglu_1536x640_{tag}", lambda dt=dt, hint=hint: check_geglu(1536, 640, dt, dev, tile_hint=hint))
                add(f"ring_geglu_ragged_200x64_{tag}", lambda dt=dt, hint=hint: check_geglu(200, 64, dt, dev, tile_hint=hint))
        add("linear_ragged_200x328x192", lambda dt=dt: check_linear(200, 328, 192, dt, dev, rowbias=True))
        # the hand-scheduled 256x192 tile (plain Linear only: no conv gather)
        h192 = _hint(5, 256, 192)
        add("h192_linear_768x640x640", lambda dt=dt: check_linear(768, 640, 640, dt, dev, tile_hint=h192))
        add("h192_linear_ragged_1000x328x192", lambda dt=dt: check_linear(1000, 328, 192, dt, dev, rowbias=True, tile_hint=h192))
        add("h192_linear_K64", lambda dt=dt: check_linear(300, 192, 64, dt, dev, tile_hint=h192))
        add("h192_linear_K128", lambda dt=dt: check_linear(2500, 136, 128, dt, dev, tile_hint=h192))
        add("h192_linear_3072x1280x1280", lambda dt=dt: check_linear(3072, 1280, 1280, dt, dev, tile_hint=h192))
        add("h192_vt_B2_N768_C640", lambda dt=dt: check_vt(2, 768, 640, dt, dev, tile_hint=h192))
        add("h192_vt_B4_N768_C1280", lambda dt=dt: check_vt(4, 768, 1280, dt, dev, tile_hint=h192))
        add("h192_geglu_1536x640", lambda dt=dt: check_geglu(1536, 640, dt, dev, tile_hint=h192))
        add("h192_geglu_ragged_200x64", lambda dt=dt: check_geglu(200, 64, dt, dev, tile_hint=h192))
        add("h192_stream_f32_768x640x640", lambda dt=dt: check_stream_f32(768, 640, 640, dt, dev, tile_hint=h192))
        # 8-byte epilogue: forced (bit 15 of the hint) and by shape (N % 8 != 0)
        add("linear_768x640x640_narrow", lambda dt=dt: check_linear(768, 640, 640, dt, dev, rowbias=True, tile_hint=_hint(1, 128, 64) | 0x8000))
        add("linear_N324_narrow", lambda dt=dt: check_linear(500, 324, 128, dt, dev, rowbias=False))
        add("geglu_1536x640_narrow", lambda dt=dt: check_geglu(1536, 640, dt, dev, tile_hint=_hint(1, 128, 256) | 0x8000))
        add("linear_M4_temb", lambda dt=dt: check_linear(4, 1280, 1280, dt, dev, res=False))
        add("linear_3072x1280x1280", lambda dt=dt: check_linear(3072, 1280, 1280, dt, dev))
        add("linear_crossKV_308x640x2048", lambda dt=dt: check_linear(308, 640, 2048, dt, dev, bias=False, res=False))
        add("geglu_1536x640", lambda dt=dt: check_geglu(1536, 640, dt, dev))
        add("geglu_ragged_200x64", lambda dt=dt: check_geglu(200, 64, dt, dev))
        add("vt_qkv_B2_N768_C640", lambda dt=dt: check_vt(2, 768, 640, dt, dev))
        add("vt_qkv_B3_N64_C128", lambda dt=dt: check_vt(3, 64, 128, dt, dev))
        add("conv3x3_320_32x24", lambda dt=dt: check_conv(2, 320, 320, 32, 24, dt, dev, temb=True))
        add("conv3x3_s2_128_17x13", lambda dt=dt: check_conv(2, 128, 192, 17, 13, dt, dev, stride=2))
        add("conv3x3_ups_128_9x7", lambda dt=dt: check_conv(2, 128, 128, 9, 7, dt, dev, ups=True))
        add("conv1x1_split_192+128", lambda dt=dt: check_conv(2, 320, 64, 12, 10, dt, dev, k=1, split=192))
        add("conv3x3_shortcut_res", lambda dt=dt: check_conv(2, 128, 128, 16, 12, dt, dev, shortcut=192, temb=True))
        add("conv3x3_res", lambda dt=dt: check_conv(1, 64, 64, 8, 8, dt, dev, res=True))
        # every (waves, stages) instantiation of the attention kernel, incl. the LDS-ring forms (Q through LDS): ragged
        # tails, closed-form zero segment, fewer tiles than ring stages (N16: 1+1 tiles), large logits (rescale path)
        for nw in (2, 4, 8):
            for stg in (2, 3, 4):
                tn, tag = (stg << 8) | nw, f"w{nw}s{stg}"
                add(f"attn_self_2seg_cfg_N768_{tag}", lambda dt=dt, tn=tn: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=tn))
                add(f"attn_self_ragged_N200_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1, tune=tn))
                add(f"attn_self_big_logits_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0, tune=tn))
                add(f"attn_self_N16_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 1, 16, dt, dev, n_garm=16, b0=1, tune=tn))
                add(f"attn_self_1seg_N1000_{tag}", lambda dt=dt, tn=tn: check_attn_self(1, 3, 1000, dt, dev, tune=tn))
                add(f"attn_cross_77_16_N768_{tag}", lambda dt=dt, tn=tn: check_attn_cross(4, 4, 768, dt, dev, tune=tn))
                add(f"attn_cross_scale0.5_N200_{tag}", lambda dt=dt, tn=tn: check_attn_cross(2, 2, 200, dt, dev, ip_scale=0.5, tune=tn))
        # ping-pong kernel (attn_pp_kernel): every instantiation (stages x priority x prefetch depth), both wave pairings,
        # every rescale threshold; incl. inputs that FORCE the deferred-rescale branch late in the key walk (spike)
        for stg in (2, 3):
            for deep in (0, 1):                          # deep builds are only launched with a pre-multiplied q (else: the 128-register build)
                for pre in (False, True):
                    tn, tag = pp_tune(stg, deep), f"pp_s{stg}d{deep}{'q' if pre else ''}"
                    add(f"attn_self_2seg_cfg_N768_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=tn, prescaled=pre))
                    add(f"attn_self_ragged_N200_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1, tune=tn, prescaled=pre))
                    add(f"attn_self_odd_tiles_N320_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(2, 2, 320, dt, dev, n_garm=192, b0=1, tune=tn, prescaled=pre))
                    add(f"attn_self_big_logits_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0, tune=tn, prescaled=pre))
                    add(f"attn_self_N16_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(2, 1, 16, dt, dev, n_garm=16, b0=1, tune=tn, prescaled=pre))
                    add(f"attn_self_1seg_N1000_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_self(1, 3, 1000, dt, dev, tune=tn, prescaled=pre))
                    add(f"attn_self_spike_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_spike(dt, dev, tune=tn, prescaled=pre))
                    add(f"attn_self_neg_logits_{tag}", lambda dt=dt, tn=tn, pre=pre: check_attn_neg(dt, dev, tune=tn, prescaled=pre))
            for thr in (1, 2, 3):
                for pair in (0, 1):
                    for deep in (0, 1):
                        tn, tag = pp_tune(stg, deep, pair=pair, thr=thr), f"pp_s{stg}d{deep}thr{thr}pair{pair}"
                        add(f"attn_self_2seg_cfg_N768_{tag}", lambda dt=dt, tn=tn: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=tn, prescaled=True))
                        add(f"attn_self_big_logits_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0, tune=tn, prescaled=True))
                        add(f"attn_self_spike_{tag}", lambda dt=dt, tn=tn: check_attn_spike(dt, dev, tune=tn, prescaled=True))
        # round 6: attn_pf_kernel (fragments read a phase early: kernel 7 row sums on the matrix pipe, 8 on the VALU) and attn_sp_kernel (software-
        # pipelined, speculative exponentials, row-sum overflow test instead of a row max: kernel 16 with 8 waves = 256 query rows per workgroup or
        # 4 waves = 128; tune bits 26-27 select the row-sum limit {512, 32, 8192, 128}) -- every edge case of the ping-pong list, incl. the inputs that
        # FORCE the rescale branch late in the key walk (spike: logits tower over a row's earlier keys; with limit 32 ordinary tiles take it too)
        for kern, nw, tag0 in ((7, 8, "pf_lsum"), (8, 8, "pf_vsum"), (16, 8, "sp8"), (16, 4, "sp4")):
            for sel in (0, 1, 2, 3) if kern == 16 else (0, 2):
                tn, tag = (sel << 26) | (kern << 16) | (3 << 8) | nw, f"{tag0}_sel{sel}"
                add(f"attn_self_2seg_cfg_N768_{tag}", lambda dt=dt, tn=tn: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=tn, prescaled=True))
                add(f"attn_self_ragged_N200_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1, tune=tn, prescaled=True))
                add(f"attn_self_odd_tiles_N320_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 320, dt, dev, n_garm=192, b0=1, tune=tn, prescaled=True))
                add(f"attn_self_big_logits_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0, tune=tn, prescaled=True))
                add(f"attn_self_N16_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 1, 16, dt, dev, n_garm=16, b0=1, tune=tn, prescaled=True))
                add(f"attn_self_one_tile_N64_{tag}", lambda dt=dt, tn=tn: check_attn_self(1, 1, 64, dt, dev, tune=tn, prescaled=True))
                add(f"attn_self_two_tiles_N128_{tag}", lambda dt=dt, tn=tn: check_attn_self(1, 2, 128, dt, dev, tune=tn, prescaled=True))
                add(f"attn_self_1seg_N1000_{tag}", lambda dt=dt, tn=tn: check_attn_self(1, 3, 1000, dt, dev, tune=tn, prescaled=True))
                add(f"attn_self_b0_3_of_5_h3_{tag}", lambda dt=dt, tn=tn: check_attn_self(5, 3, 300, dt, dev, n_garm=130, b0=3, tune=tn, prescaled=True))
                add(f"attn_self_spike_{tag}", lambda dt=dt, tn=tn: check_attn_spike(dt, dev, tune=tn, prescaled=True))
                add(f"attn_self_neg_logits_{tag}", lambda dt=dt, tn=tn: check_attn_neg(dt, dev, tune=tn, prescaled=True))
            add(f"attn_self_N3072_h10_{tag0}", lambda dt=dt, kern=kern, nw=nw: check_attn_self(4, 10, 3072, dt, dev, n_garm=3072, b0=2, tune=(kern << 16) | (3 << 8) | nw, prescaled=True))
        add("attn_self_N3072_h10_pp_s3d1", lambda dt=dt: check_attn_self(4, 10, 3072, dt, dev, n_garm=3072, b0=2, tune=pp_tune(3, 1)))
        add("attn_self_N3072_h10_pp_s2d0", lambda dt=dt: check_attn_self(4, 10, 3072, dt, dev, n_garm=3072, b0=2, tune=pp_tune(2, 0)))
        add("attn_self_spike", lambda dt=dt: check_attn_spike(dt, dev))
        for tn, tag in ((0, "auto"), ((2 << 8) | 8, "w8s2"), ((3 << 8) | 4, "w4s3"), (pp_tune(2, 0), "pp_s2d0"), (pp_tune(3, 1), "pp_s3d1")):
            add(f"attn_self_prescaled_2seg_{tag}", lambda dt=dt, tn=tn: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=tn, prescaled=True))
            add(f"attn_self_prescaled_ragged_{tag}", lambda dt=dt, tn=tn: check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1, tune=tn, prescaled=True))
            add(f"attn_self_neg_logits_{tag}", lambda dt=dt, tn=tn: check_attn_neg(dt, dev, tune=tn))
        # fp8 (e4m3) attention on the block-scaled MFMA.  e4m3 has 3 mantissa bits: every q, k, v element carries up to 6 % (3.6 % rms)
        # relative rounding, and on N(0,1) operands the attention output (an average of ~N values) inherits ~3-4 % rms / 6-10 % max of
        # its own magnitude.  Measured on MI355X (profiles/r03_fp8_attention_checks.log): 6.2e-2 .. 1.03e-1 against fp32 SDPA on the
        # unquantised operands, 1.6e-2 .. 2.1e-2 against fp32 SDPA on the DEQUANTISED operands (P's e4m3 rounding + the kernel's own
        # arithmetic).  Stated tolerances of this variant: 1.2e-1 and 3e-2.
        for (nm, args) in (("2seg_cfg_N768", (4, 4, 768, 768, 2)), ("ragged_N200", (2, 2, 200, 200, 1)), ("1seg_N1000", (1, 3, 1000, 0, 0)),
                           ("N3072_h10", (4, 10, 3072, 3072, 2)), ("N16", (2, 1, 16, 16, 1))):
            add(f"attn_f8_{nm}", lambda dt=dt, a=args: check_attn_f8(a[0], a[1], a[2], dt, dev, n_garm=a[3], b0=a[4])[0], 1.2e-1)
            add(f"attn_f8_kernel_only_{nm}", lambda dt=dt, a=args: check_attn_f8(a[0], a[1], a[2], dt, dev, n_garm=a[3], b0=a[4])[1], 3e-2)
        add("quant_f8", lambda dt=dt: check_quant_f8(dt, dev), 0.0)
        # the projections of the fp8 path writing e4m3 themselves (IDMVTON_IO_OUT_F8), on every tile family the tuned table may select
        for hn, hv in F8_OUT_TILES:
            add(f"gemm_f8_out_{hn}", lambda dt=dt, hv=hv: check_gemm_f8_out(dt, dev, B=2, N=192, C=256, K=320, hint=hv), 0.0)
        add("gemm_f8_out_N768_C640", lambda dt=dt: check_gemm_f8_out(dt, dev, B=4, N=768, C=640, K=640), 0.0)
        add("gemm_f8_out_320_tile_is_refused", lambda dt=dt: check_f8_out_320_refused(dt, dev), 0.0)
        add("gemm_f8_out_bias_free_kv_only", lambda dt=dt: check_gemm_f8_kv(dt, dev), 0.0)
        # the same bytes through idmvton_attn_f8 vs the two-launch route: both are e4m3 roundings of the same values (one vs two roundings)
        add("gemm_f8_out_feeds_attn_f8", lambda dt=dt: check_gemm_f8_out(dt, dev, B=2, N=256, C=128, K=256, fused_attn=True), 6e-2)
        add("linear_colscale", lambda dt=dt: check_colscale(dt, dev))
        add("linear_quickgelu", lambda dt=dt: check_quickgelu(dt, dev))
        add("attn_small_text_causal_77_d64", lambda dt=dt: check_attn_small(2, 12, 77, 64, dt, dev, True))
        add("attn_small_vision_257_d80", lambda dt=dt: check_attn_small(2, 16, 257, 80, dt, dev, False))
        add("attn_small_causal_offset_d32", lambda dt=dt: check_attn_small(3, 2, 100, 32, dt, dev, True, Lq=37))
        add("attn_small_big_logits_d128", lambda dt=dt: check_attn_small(1, 2, 300, 128, dt, dev, False, scale=4.0))
        add("attn_small_L1", lambda dt=dt: check_attn_small(2, 2, 1, 64, dt, dev, True))
        add("attn_self_1seg_N768", lambda dt=dt: check_attn_self(2, 4, 768, dt, dev))
        add("attn_self_2seg_cfg_N768", lambda dt=dt: check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2))
        add("attn_self_2seg_ragged_N200", lambda dt=dt: check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1))
        add("attn_self_big_logits", lambda dt=dt: check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0))
        # the lazy running max of the ping-pong kernel: logits that climb tile after tile (moves on both halves of a tile), and jumps of
        # hundreds of binades above the standing max (exp2 overflows to +inf before the max is moved)
        add("attn_self_pp_big_logits_N2048", lambda dt=dt: check_attn_self(2, 4, 2048, dt, dev, n_garm=2048, b0=1, scale=4.0, tune=(2 << 16) | (2 << 8) | 8))
        add("attn_self_pp_huge_logits_N2048", lambda dt=dt: check_attn_self(2, 4, 2048, dt, dev, n_garm=2048, b0=1, scale=12.0, tune=(2 << 16) | (2 << 8) | 8))
        add("attn_self_pp_deep_huge_logits_N2048", lambda dt=dt: check_attn_self(2, 4, 2048, dt, dev, n_garm=2048, b0=1, scale=12.0, tune=(3 << 16) | (2 << 8) | 8, prescaled=True))
        add("attn_self_pp_huge_logits_ragged_N1000", lambda dt=dt: check_attn_self(2, 4, 1000, dt, dev, n_garm=1000, b0=1, scale=12.0, tune=(2 << 16) | (2 << 8) | 8))
        add("attn_self_N3072_h10", lambda dt=dt: check_attn_self(4, 10, 3072, dt, dev, n_garm=3072, b0=2))
        add("attn_self_N16", lambda dt=dt: check_attn_self(2, 1, 16, dt, dev, n_garm=16, b0=1))
        # BASELINE.json configs[3] (1024x1536): the two-segment walks of TryonNet at 6144 + 6144 keys (L1) and 1536 + 1536 (L2)
        add("attn_self_cfg4_N6144_h10", lambda dt=dt: check_attn_self(2, 10, 6144, dt, dev, n_garm=6144, b0=1))
        add("attn_self_cfg4_N1536_h20", lambda dt=dt: check_attn_self(2, 20, 1536, dt, dev, n_garm=1536, b0=1))
        add("attn_cross_77_16_N768", lambda dt=dt: check_attn_cross(4, 4, 768, dt, dev))
        for hint, tag in ((0, "auto"), (_hint(1, 128, 64), "128x64"), (_hint(1, 128, 128), "128x128"), (_hint(1, 128, 256), "128x256"), (_hint(6, 128, 128), "w8_128x128")):
            add(f"xattn_fused_B4_h20_N768_{tag}", lambda dt=dt, hint=hint: check_xattn_fused(4, 20, 768, 1280, dt, dev, tile_hint=hint))
            add(f"xattn_fused_text_only_B2_h4_N96_{tag}", lambda dt=dt, hint=hint: check_xattn_fused(2, 4, 96, 256, dt, dev, n_ip=0, tile_hint=hint))
        add("xattn_fused_B2_h10_N3072_ipscale0.5", lambda dt=dt: check_xattn_fused(2, 10, 3072, 640, dt, dev, ip_scale=0.5))
        add("xattn_fused_ragged_keys_33_5", lambda dt=dt: check_xattn_fused(3, 2, 160, 128, dt, dev, n_text=33, n_ip=5))
        add("attn_cross_scale0.5_N200", lambda dt=dt: check_attn_cross(2, 2, 200, dt, dev, ip_scale=0.5))
        for hint, tag in ((0, "auto"),) + tuple(RING_TILES):
            add(f"stream_f32_768x640x640_{tag}", lambda dt=dt, hint=hint: check_stream_f32(768, 640, 640, dt, dev, tile_hint=hint))
        add("stream_f32_ragged_1000x328x192", lambda dt=dt: check_stream_f32(1000, 328, 192, dt, dev))
        add("stream_f32_3072x1280x5120", lambda dt=dt: check_stream_f32(3072, 1280, 5120, dt, dev))
        add("layernorm_640", lambda dt=dt: check_layernorm(1000, 640, dt, dev))
        add("layernorm_1280", lambda dt=dt: check_layernorm(3072, 1280, dt, dev))
        add("layernorm_64", lambda dt=dt: check_layernorm(37, 64, dt, dev))
        add("groupnorm_320_silu", lambda dt=dt: check_groupnorm(2, 768, 320, dt, dev))
        add("groupnorm_1920_split1280", lambda dt=dt: check_groupnorm(2, 300, 1920, dt, dev, split=1280))
        add("groupnorm_2560_split1280", lambda dt=dt: check_groupnorm(2, 192, 2560, dt, dev, split=1280))
        add("groupnorm_128_nosilu_eps1e-6", lambda dt=dt: check_groupnorm(1, 4096, 128, dt, dev, silu=False, eps=1e-6))
        add("groupnorm_reproducible_320", lambda dt=dt: check_groupnorm_reproducible(4, 12288, 320, dt, dev), 0.0)
        add("groupnorm_reproducible_2560", lambda dt=dt: check_groupnorm_reproducible(2, 768, 2560, dt, dev), 0.0)
        add("elementwise", lambda dt=dt: check_elementwise(2, 16, 12, dt, dev))
        add("conv3x3_ups_to_odd_grid_63x47", lambda dt=dt: check_conv_ups_odd(2, 128, 128, 32, 24, 63, 47, dt, dev))
        add("conv3x3_ups_to_mixed_grid_17x26", lambda dt=dt: check_conv_ups_odd(1, 64, 192, 9, 13, 17, 26, dt, dev))
    # the split-precision (fp32-equivalent) path of the VAE decode: operand pairs are bf16 whatever the engine's storage type.  Bars: a
    # 3-term product carries ~2^-16 per operand (measured ~3e-6 of the output range); the 16-bit path sits at 2e-3 (fp16) / 1.6e-2 (bf16)
    from idm_vton_amd import ffi
    P = lambda name, fn, t: out.append((f"{name}[split]", fn, t))
    for mode, tag in ((ffi.SPLIT_ACT, "act"), (ffi.SPLIT_W3, "w3"), (ffi.SPLIT_W3T, "w3t")):
        P(f"split_{tag}_768x512", lambda mode=mode: check_split(768, 512, dev, mode), 1e-7)
        P(f"split_{tag}_ragged_200x72", lambda mode=mode: check_split(200, 72, dev, mode), 1e-7)
    P("gn_precise_512_silu", lambda: check_gn_precise(2, 768, 512, dev), 2e-5)
    P("gn_precise_128_nosilu", lambda: check_gn_precise(1, 4096, 128, dev, silu=False), 2e-5)
    P("softmax_split_64x3072", lambda: check_softmax_split(64, 3072, dev), 2e-5)
    P("softmax_split_padded_keys_825_of_832", lambda: check_softmax_split(37, 832, dev, n_valid=825), 2e-5)
    for dt in (torch.float16, torch.bfloat16):
        out.append((f"softmax_rows_64x3072[{dt}]", lambda dt=dt: check_softmax_rows(64, 3072, dt, dev), TOL[dt]))
        out.append((f"softmax_rows_padded_keys_825_of_832[{dt}]", lambda dt=dt: check_softmax_rows(37, 832, dt, dev, n_valid=825), TOL[dt]))
    P("plin_768x512x512_3term", lambda: check_plin(768, 512, 512, dev), 2e-5)
    P("plin_768x512x512_exact_weights_2term", lambda: check_plin(768, 512, 512, dev, exact_w=True), 2e-5)
    P("plin_ragged_1000x64x128", lambda: check_plin(1000, 64, 128, dev, res=False), 2e-5)
    P("pconv_128_to_128_16x12", lambda: check_pconv(2, 128, 128, 16, 12, dev), 2e-5)
    P("pconv_256_to_128_shortcut", lambda: check_pconv(1, 128, 128, 16, 12, dev, shortcut=256), 2e-5)
    P("pconv_ups_128", lambda: check_pconv(2, 128, 128, 9, 7, dev, ups=True), 2e-5)
    P("pconv_exact_weights_shortcut", lambda: check_pconv(1, 128, 64, 8, 8, dev, shortcut=128, exact_w=True), 2e-5)
    P("layout_split_and_f32_nhwc", lambda: check_layout_split(2, 16, 12, dev), 2e-5)
    return out
