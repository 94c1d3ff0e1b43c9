"""-m gpu: hypothesis-drawn shape sweeps of the C-ABI kernels against their fp32 torch references.  The fixed list of
tests/kernel_checks.py::all_checks holds the reference's real shapes and every tile / pipeline variant; this file draws the shapes
nobody thought of: ragged M / N tails, K of one to eight 64-wide steps, odd spatial sizes, token counts that are any multiple of 16,
garment segments of a different length than the own segment.  derandomize=True: the same examples on every run (a failure is
reproducible from the printed arguments); deadline off (first launches JIT nothing but page code in)."""
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import kernel_checks as kc

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = st.sampled_from([torch.float16, torch.bfloat16])
CFG = settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
TILES = st.sampled_from([0] + [h for h, _ in kc.RING_TILES] + [(128 << 16) | 128, (128 << 16) | 64, (64 << 16) | 64])


@CFG
@given(M=st.integers(1, 900), n4=st.integers(1, 160), k64=st.integers(1, 8), dt=DT, bias=st.booleans(), res=st.booleans(),
       rowbias=st.booleans(), hint=TILES)
def test_linear_any_shape(M, n4, k64, dt, bias, res, rowbias, hint):
    e = kc.check_linear(M, 4 * n4, 64 * k64, dt, DEV, bias=bias, res=res, rowbias=rowbias and M % 4 == 0, tile_hint=hint)
    assert e <= kc.TOL[dt], (M, 4 * n4, 64 * k64, dt, hint, e)


@CFG
@given(B=st.integers(1, 3), cin=st.sampled_from([64, 128, 192, 320]), co8=st.integers(1, 40), H=st.integers(1, 19), W=st.integers(1, 19),
       dt=DT, mode=st.sampled_from(["s1", "s2", "ups", "1x1"]), temb=st.booleans(), res=st.booleans())
def test_conv_any_shape(B, cin, co8, H, W, dt, mode, temb, res):
    kw = dict(stride=2) if mode == "s2" else (dict(ups=True) if mode == "ups" else (dict(k=1) if mode == "1x1" else {}))
    e = kc.check_conv(B, cin, 8 * co8, H, W, dt, DEV, temb=temb, res=res and mode in ("s1", "1x1"), **kw)
    assert e <= kc.TOL[dt], (B, cin, 8 * co8, H, W, dt, mode, e)


@CFG
@given(B=st.integers(1, 4), heads=st.integers(1, 5), n16=st.integers(1, 40), g16=st.integers(0, 40), dt=DT, b0f=st.integers(0, 4),
       pre=st.booleans(), tune=st.sampled_from([0, (2 << 8) | 8, (3 << 8) | 4, (2 << 8) | 2]))
def test_self_attention_any_shape(B, heads, n16, g16, dt, b0f, pre, tune):
    """Own segment of 16*n16 tokens, garment segment of 16*g16 (0: none) present for batches >= b0: lengths need not match."""
    b0 = min(b0f, B)
    e = kc.check_attn_self(B, heads, 16 * n16, dt, DEV, n_garm=16 * g16 if b0 < B else 0, b0=b0, tune=tune, prescaled=pre)
    assert e <= kc.TOL[dt], (B, heads, 16 * n16, 16 * g16, b0, dt, tune, e)


@CFG
@given(B=st.integers(1, 5), heads=st.integers(1, 5), n16=st.integers(1, 40), g16=st.integers(0, 40), dt=DT, b0f=st.integers(0, 5),
       kern=st.sampled_from([(16, 8), (16, 4), (7, 8), (8, 8)]), sel=st.integers(0, 3), scale=st.sampled_from([0.3, 1.0, 3.0]))
def test_self_attention_round6_kernels_any_shape(B, heads, n16, g16, dt, b0f, kern, sel, scale):
    """attn_sp_kernel (8 / 4 waves) and attn_pf_kernel on drawn shapes: any multiple of 16 tokens (one tile, odd tile counts, ragged last tiles),
    garment segment of another length, any split of the batch into unconditional / conditional elements (the work order's two classes, either
    may be empty), every row-sum limit / rescale threshold, small and large logits."""
    b0 = min(b0f, B)
    tune = (sel << 26) | (kern[0] << 16) | (3 << 8) | kern[1]
    e = kc.check_attn_self(B, heads, 16 * n16, dt, DEV, n_garm=16 * g16 if b0 < B else 0, b0=b0, scale=scale, tune=tune, prescaled=True)
    assert e <= kc.TOL[dt], (B, heads, 16 * n16, 16 * g16, b0, dt, kern, sel, scale, e)


@CFG
@given(B=st.integers(1, 4), heads=st.integers(1, 5), n16=st.integers(1, 48), dt=DT, scale=st.sampled_from([0.0, 0.5, 1.0, 2.0]))
def test_cross_attention_any_shape(B, heads, n16, dt, scale):
    e = kc.check_attn_cross(B, heads, 16 * n16, dt, DEV, ip_scale=scale)
    assert e <= kc.TOL[dt], (B, heads, 16 * n16, dt, scale, e)


@CFG
@given(rows=st.integers(1, 3000), c8=st.integers(1, 256), dt=DT)
def test_layernorm_any_shape(rows, c8, dt):
    e = kc.check_layernorm(rows, 8 * c8, dt, DEV)
    assert e <= kc.TOL[dt], (rows, 8 * c8, dt, e)


@CFG
@given(B=st.integers(1, 5), HW=st.integers(1, 1500), gch=st.sampled_from([(32, 320), (32, 640), (32, 1280), (32, 128), (8, 64), (4, 32), (32, 1920)]),
       dt=DT, silu=st.booleans(), split=st.booleans())
def test_groupnorm_any_shape(B, HW, gch, dt, silu, split):
    groups, C = gch
    e = kc.check_groupnorm(B, HW, C, dt, DEV, groups=groups, silu=silu, split=(C // 2 // 8 * 8) if split and C >= 128 else 0)
    assert e <= kc.TOL[dt], (B, HW, C, groups, dt, silu, split, e)
