"""CPU tests of the oracle: pinned against the reference's own Resampler (live when /root/reference is present, and via
the committed golden fixture), self-checked by SURVEY.md A.1 (parameter counts), A.5 (algebraic identities the kernels
exploit), Appendix C (state-dict keys) and regression-pinned by tests/golden/tiny_pipeline.safetensors."""
import importlib.util
import os

import pytest
import torch
from safetensors import safe_open

from oracle import layers as L
from oracle.resampler import Resampler
from oracle.scheduler import Scheduler
from oracle.unet import UNet2DConditionModel, UNetConfig
from oracle.vae import AutoencoderKL
from tests import parity_utils as pu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference/ip_adapter/resampler.py"


def _load(name):
    with safe_open(os.path.join(GOLD, name), "pt") as f:
        return {k: f.get_tensor(k) for k in f.keys()}, f.metadata()


def test_resampler_matches_golden_from_reference_file():
    t, meta = _load("resampler_ref.safetensors")
    m = Resampler(**eval(meta["kw"])).eval()
    m.load_state_dict({k[3:]: v for k, v in t.items() if k.startswith("sd.")})
    with torch.no_grad():
        y = m(t["x"])
    assert torch.equal(y, t["y"]), (y - t["y"]).abs().max()


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_resampler_matches_reference_file():
    spec = importlib.util.spec_from_file_location("ref_resampler", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    kw = dict(dim=64, depth=1, dim_head=32, heads=2, num_queries=8, embedding_dim=48, output_dim=96, ff_mult=2)
    torch.manual_seed(0)
    a, b = ref.Resampler(**kw).eval(), Resampler(**kw).eval()
    b.load_state_dict(a.state_dict())
    x = torch.randn(3, 21, 48)
    with torch.no_grad():
        assert torch.equal(a(x), b(x))


def test_parameter_counts_match_survey_a1():
    with torch.device("meta"):
        t = UNet2DConditionModel(UNetConfig.sdxl_tryon())
        g = UNet2DConditionModel(UNetConfig.sdxl_garmnet())
        v = AutoencoderKL()
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(t.encoder_hid_proj) == 82_961_664                      # SURVEY 0.2 (instantiated reference Resampler)
    assert round((n(t) - n(t.encoder_hid_proj)) / 1e6) == 2908      # 2.908 B
    assert round(n(g) / 1e6) == 2562                                # 2.562 B
    ip = sum(p.numel() for k, p in t.named_parameters() if "to_k_ip" in k or "to_v_ip" in k)
    assert round(ip / 1e5) == 3408                                  # 340.8 M IP K/V
    assert n(v) == 83_653_863                                       # SDXL VAE


def test_state_dict_keys_follow_appendix_c():
    with torch.device("meta"):
        t = UNet2DConditionModel(UNetConfig.sdxl_tryon())
    keys = set(t.state_dict().keys())
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias",
              "down_blocks.1.attentions.0.transformer_blocks.1.attn2.processor.to_k_ip.weight",
              "down_blocks.2.attentions.1.transformer_blocks.9.ff.net.0.proj.weight",
              "mid_block.attentions.0.transformer_blocks.9.attn1.to_out.0.bias", "up_blocks.0.upsamplers.0.conv.weight",
              "up_blocks.1.resnets.2.conv_shortcut.weight", "down_blocks.0.downsamplers.0.conv.bias",
              "encoder_hid_proj.latents", "encoder_hid_proj.layers.3.0.to_kv.weight", "encoder_hid_proj.layers.0.1.3.weight",
              "conv_norm_out.weight", "conv_out.bias"):
        assert k in keys, k
    from idm_vton_amd import config as pc
    assert dict(pc.unet_param_shapes(pc.UNetConfig.sdxl_tryon())) == {k: tuple(v.shape) for k, v in t.state_dict().items()}


def test_feature_count_and_order():
    m = pu.build("tiny", torch.float32, "cpu")
    o_t, o_g, _ = m["oracle"]
    with torch.no_grad():
        _, feats = o_g(torch.randn(1, 4, 16, 16), 500, torch.randn(1, 77, m["xd"]))
    # traversal order: down1 (2x1 @8x8), down2 (2x2 @4x4), mid (2), up0 (3x2 @4x4), up1 (3x1 @8x8)
    assert [f.shape[1] for f in feats] == [64] * 2 + [16] * 4 + [16] * 2 + [16] * 6 + [64] * 3
    with torch.device("meta"):
        assert sum(len(t.transformer_blocks) for t in UNet2DConditionModel(UNetConfig.sdxl_garmnet()).modules()
                   if hasattr(t, "transformer_blocks")) == 70


def test_identity_query_truncation():
    """rows [N:2N] of the concatenated self-attention are discarded, so only N query rows are needed."""
    torch.manual_seed(0)
    attn = L.Attention(128, 2, 64)
    x, gfeat = torch.randn(2, 24, 128), torch.randn(2, 24, 128)
    m = torch.cat([x, gfeat], dim=1)
    with torch.no_grad():
        full = attn(m)[:, :24]
        q = attn.to_q(x).view(2, 24, 2, 64).transpose(1, 2)
        k = attn.to_k(m).view(2, 48, 2, 64).transpose(1, 2)
        v = attn.to_v(m).view(2, 48, 2, 64).transpose(1, 2)
        o = L.sdpa(q, k, v).transpose(1, 2).reshape(2, 24, 128)
        trunc = attn.to_out[0](o)
    assert torch.allclose(full, trunc, atol=1e-6)


def test_identity_zero_garment_closed_form():
    """an all-zero garment half gives N keys with logit 0 and value 0 (to_k/to_v have no bias):
    out = sum_self p v / (sum_self p + N exp(-m))."""
    torch.manual_seed(1)
    N, d = 40, 64
    q, k, v = torch.randn(3, N, d), torch.randn(3, N, d), torch.randn(3, N, d)
    kk = torch.cat([k, torch.zeros_like(k)], dim=1)
    vv = torch.cat([v, torch.zeros_like(v)], dim=1)
    ref = L.sdpa(q, kk, vv)
    s = (q @ k.transpose(-2, -1)) * d ** -0.5
    m = s.max(-1, keepdim=True).values.clamp_min(0.0)
    p = torch.exp(s - m)
    out = (p @ v) / (p.sum(-1, keepdim=True) + N * torch.exp(-m))
    assert torch.allclose(out, ref, atol=2e-6)


def test_scheduler_timesteps_and_final_step():
    s = Scheduler("ddpm")
    ts = s.set_timesteps(30)
    assert ts[0].item() == 958 and ts[1].item() == 925 and ts[-1].item() == 1 and len(ts) == 30
    c_eps, c_x, sigma = s.coeffs(1)                     # prev_t < 0 -> alpha_bar_prev = 1 -> x_prev = x0, sigma ~ 0
    ab = float(s.alphas_cumprod[1])
    assert abs(c_x - 1 / ab ** 0.5) < 1e-9 and abs(c_eps + (1 - ab) ** 0.5 / ab ** 0.5) < 1e-9 and sigma <= 1e-9
    d = Scheduler("ddim")
    d.set_timesteps(30)
    x, e = torch.randn(4), torch.randn(4)
    ab_t, ab_p = float(d.alphas_cumprod[958]), float(d.alphas_cumprod[925])
    x0 = (x - (1 - ab_t) ** 0.5 * e) / ab_t ** 0.5
    assert torch.allclose(d.step(e, 958, x), ab_p ** 0.5 * x0 + (1 - ab_p) ** 0.5 * e, atol=1e-6)


def test_product_scheduler_matches_oracle_scheduler():
    from idm_vton_amd.scheduler import StepScheduler
    for kind in ("ddpm", "ddim"):
        a, b = Scheduler(kind), StepScheduler(kind)
        ta, tb = a.set_timesteps(30), b.set_timesteps(30)
        assert list(ta.tolist()) == list(tb.tolist())
        for t in tb:
            ce, cx, sg = a.coeffs(t)
            cx2, ce2, sg2 = b.coeffs(t)
            assert abs(ce - ce2) < 1e-7 and abs(cx - cx2) < 1e-7 and abs(sg - sg2) < 1e-7


def test_tiny_pipeline_regression_golden():
    """The oracle reproduces its own committed golden run (fp16-rounded storage of the fixture => 2e-3)."""
    from oracle import pipeline as opipe
    t, _ = _load("tiny_pipeline.safetensors")
    m = pu.build("tiny", torch.float16, "cpu")
    o_t, o_g, o_v = m["oracle"]
    inp = pu.make_inputs(1, 128, 128, m["xd"], m["pooled"], m["enc_dim"], 4, torch.float16)
    tr = {}
    img = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=4, guidance_scale=2.0, trace=tr, **inp)
    assert pu.relerr(tr["step_latents"][-1], t["latents_3"]) < 2e-3
    assert pu.relerr(img, t["image"]) < 2e-3
