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


def test_ddim_last_step_uses_final_alpha_cumprod():
    """diffusers DDIMScheduler: the step whose previous timestep is < 0 (leading spacing, steps_offset 1: t = 1) uses
    final_alpha_cumprod = alphas_cumprod[0] when set_alpha_to_one is false (the SDXL scheduler_config value), 1.0 when true;
    DDPMScheduler always uses 1.0 there.  alphas_cumprod[0] = 1 - beta_start = 0.99915 for scaled_linear 0.00085..0.012."""
    from idm_vton_amd.scheduler import StepScheduler
    for cls in (Scheduler, StepScheduler):
        d = cls("ddim")
        ts = d.set_timesteps(30)
        assert int(ts[-1]) == 1 and abs(float(d.alphas_cumprod[0]) - 0.99915) < 1e-6
        ab_t, ab_p = float(d.alphas_cumprod[1]), float(d.alphas_cumprod[0])
        c = d.coeffs(1)
        c_x, c_eps = (c[1], c[0]) if cls is Scheduler else (c[0], c[1])
        assert abs(c_x - (ab_p / ab_t) ** 0.5) < 1e-9
        assert abs(c_eps - ((1 - ab_p) ** 0.5 - (ab_p * (1 - ab_t) / ab_t) ** 0.5)) < 1e-9
        one = cls("ddim", set_alpha_to_one=True)
        one.set_timesteps(30)
        c1 = one.coeffs(1)
        assert abs((c1[1] if cls is Scheduler else c1[0]) - (1.0 / ab_t) ** 0.5) < 1e-9
        p = cls("ddpm")
        p.set_timesteps(30)
        cp = p.coeffs(1)                                    # DDPM at t=1: x_prev = x0 exactly (previous alpha-bar 1), sigma from the clamp
        assert abs((cp[1] if cls is Scheduler else cp[0]) - (1.0 / ab_t) ** 0.5) < 1e-6


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


# ------------------------------------------------------------------------------------------------------------------
# Oracle pinned to the REFERENCE'S OWN first-party code: tests/golden/reference_unet_tiny.safetensors holds outputs of
# /root/reference/src/unet_hacked_{tryon,garmnet}.py, attentionhacked_*.py and ip_adapter/attention_processor.py executed
# unmodified (oracle/make_golden_ref.py; only `diffusers` itself is the test-side stand-in tests/compat/refstub).
# ------------------------------------------------------------------------------------------------------------------
REF_TINY = dict(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), num_attention_heads=(1, 2, 4),
                cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32,
                encoder_hid_dim=128, resampler=dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, ff_mult=4))
PIN_TOL = 2e-5          # fp32 vs fp32, different (equivalent) op order: e.g. softmax written out vs F.scaled_dot_product_attention


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def _oracle_pair():
    import dataclasses
    from idm_vton_amd import config as pc
    from oracle import unet as ou
    tcfg = pc.UNetConfig(mode="tryon", in_channels=13, **REF_TINY)
    gcfg = pc.UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None, **REF_TINY)
    as_o = lambda c: ou.UNetConfig(**{f.name: getattr(c, f.name) for f in dataclasses.fields(ou.UNetConfig)})
    o_t, o_g = ou.UNet2DConditionModel(as_o(tcfg)).eval(), ou.UNet2DConditionModel(as_o(gcfg)).eval()
    sd_t = pc.random_state_dict(pc.unet_param_shapes(tcfg), 101, torch.float32, "cpu")
    sd_g = pc.random_state_dict(pc.unet_param_shapes(gcfg), 102, torch.float32, "cpu")
    o_t.load_state_dict(sd_t)
    o_g.load_state_dict(sd_g)
    return o_t, o_g, sd_t, sd_g


def _check_against_reference_fixture(t, meta):
    o_t, o_g, sd_t, sd_g = _oracle_pair()
    # same weights as the reference run (seeded generator; the sums are stored with the fixture)
    assert abs(float(sum(v.double().abs().sum() for v in sd_t.values())) - float(meta["wsum_t"])) < 1e-6 * float(meta["wsum_t"])
    assert abs(float(sum(v.double().abs().sum() for v in sd_g.values())) - float(meta["wsum_g"])) < 1e-6 * float(meta["wsum_g"])
    with torch.no_grad():
        # GarmentNet (src/unet_hacked_garmnet.py:917-1284): every exported norm1 output, in order
        _, feats = o_g(t["garm.cloth_lat"], 481, t["garm.cloth_text"])
        assert len(feats) == int(meta["n_garm_feats"]) == 17
        for i, f in enumerate(feats):
            assert _rel(f, t[f"garm.feat{i:02d}"]) <= PIN_TOL, (i, _rel(f, t[f"garm.feat{i:02d}"]))
        # Resampler as the UNet's encoder_hid_proj (tryon_pipeline.py:1726)
        ie = o_t.encoder_hid_proj(t["tryon.ip_states"])
        assert _rel(ie, t["tryon.image_embeds"]) <= PIN_TOL
        # TryonNet (src/unet_hacked_tryon.py:1006-1395) on the reference's own features, zero half materialised (:1796)
        B2 = t["tryon.lmi"].shape[0]
        time_ids = torch.tensor([[128, 128, 0, 0, 128, 128]], dtype=torch.float32).repeat(B2, 1)
        feats_cfg = [torch.cat([torch.zeros_like(t[f"garm.feat{i:02d}"]), t[f"garm.feat{i:02d}"]]) for i in range(17)]
        eps = o_t(t["tryon.lmi"], 481, t["tryon.pe"], added_cond_kwargs=dict(text_embeds=t["tryon.add_text"], time_ids=time_ids,
                  image_embeds=t["tryon.image_embeds"]), garment_features=feats_cfg)[0]
        assert _rel(eps, t["tryon.eps"]) <= PIN_TOL, _rel(eps, t["tryon.eps"])


def _check_blocks_against_reference_fixture(t):
    from oracle.unet import BasicTransformerBlock
    dim, heads, hd, xd = 64, 1, 64, 64
    with torch.no_grad():
        bt = BasicTransformerBlock(dim, heads, hd, xd, "tryon", 16).eval()
        bt.load_state_dict({k[len("blk_t.sd."):]: v for k, v in t.items() if k.startswith("blk_t.sd.")})     # strict: same keys
        y, idx, _ = bt(t["blk.x"], t["blk.enc_t"], [t["blk.garm"]], 0)
        assert idx == 1 and _rel(y, t["blk_t.y"]) <= PIN_TOL, _rel(y, t["blk_t.y"])
        bg = BasicTransformerBlock(dim, heads, hd, xd, "garmnet", 16).eval()
        bg.load_state_dict({k[len("blk_g.sd."):]: v for k, v in t.items() if k.startswith("blk_g.sd.")})
        y, _, feat = bg(t["blk.x"], t["blk.enc_g"])
        assert _rel(y, t["blk_g.y"]) <= PIN_TOL and torch.equal(feat, t["blk_g.feat"])
        # the two processors on their own (ip_adapter/attention_processor.py:203-278, :1907-2010) inside the block's modules
        a2 = bt.attn2
        n2 = bt.norm2(t["blk.x"])
        ref_like = a2(n2, encoder_hidden_states=t["blk.enc_t"])
        q = a2.to_q(n2)
        text, ip = t["blk.enc_t"][:, :77], t["blk.enc_t"][:, 77:]
        man = (torch.softmax(q @ a2.to_k(text).transpose(1, 2) / 8.0, -1) @ a2.to_v(text)
               + torch.softmax(q @ a2.processor.to_k_ip(ip).transpose(1, 2) / 8.0, -1) @ a2.processor.to_v_ip(ip))
        assert _rel(ref_like, a2.to_out[0](man)) <= PIN_TOL


def _check_vae_blocks_against_reference_fixture(t):
    """oracle/vae.py's blocks == the VAE building blocks the reference carries (src/unet_block_hacked_tryon.py: UNetMidBlock2D
    :505-627 with the reference's AttnProcessor2_0 on the 4-D map, DownEncoderBlock2D :1292-1349, UpDecoderBlock2D :2511-2568):
    same state-dict keys (strict load), same outputs."""
    from oracle import vae as ov
    G = 8
    blocks = {"mid": ov.MidBlock(64, G), "down": ov.DownEncoderBlock2D(32, 64, 2, G, True), "down_last": ov.DownEncoderBlock2D(64, 64, 2, G, False),
              "up": ov.UpDecoderBlock2D(64, 32, 3, G, True), "up_last": ov.UpDecoderBlock2D(32, 32, 3, G, False)}
    with torch.no_grad():
        for name, blk in blocks.items():
            pre = f"vae.{name}.sd."
            blk.eval().load_state_dict({k[len(pre):]: v for k, v in t.items() if k.startswith(pre)})          # strict: same keys
            y = blk(t[f"vae.{name}.x"])
            assert y.shape == t[f"vae.{name}.y"].shape, name
            assert _rel(y, t[f"vae.{name}.y"]) <= PIN_TOL, (name, _rel(y, t[f"vae.{name}.y"]))
    assert t["vae.down.y"].shape[-2:] == (6, 5) and t["vae.up.y"].shape[-2:] == (12, 10)     # pad(0,1,0,1)+s2 p0; nearest x2


def test_oracle_vae_blocks_match_reference_code_golden():
    t, _ = _load("reference_unet_tiny.safetensors")
    _check_vae_blocks_against_reference_fixture(t)


def test_oracle_unets_match_reference_code_golden():
    """oracle TryonNet / GarmentNet == the reference's own UNet2DConditionModel forwards (committed fixture)."""
    t, meta = _load("reference_unet_tiny.safetensors")
    _check_against_reference_fixture(t, meta)


def test_oracle_blocks_match_reference_code_golden():
    """oracle BasicTransformerBlock (tryon: garment concat + IP cross-attention; garmnet: norm1 export) and the attention
    processors == the reference's own classes (committed fixture)."""
    t, _ = _load("reference_unet_tiny.safetensors")
    _check_blocks_against_reference_fixture(t)


@pytest.mark.skipif(not os.path.exists("/root/reference/src/unet_hacked_tryon.py"), reason="reference checkout not present (GPU box)")
def test_reference_code_golden_is_fresh(tmp_path):
    """Re-run the reference's code here (subprocess: oracle/make_golden_ref.py) and compare with the committed fixture, so the
    fixture cannot drift from /root/reference or from tests/compat/refstub."""
    from oracle import make_golden as mg
    out = mg.reference_unet_fixture(str(tmp_path / "fresh.safetensors"))
    with safe_open(out, "pt") as f:
        fresh = {k: f.get_tensor(k) for k in f.keys()}
    t, _ = _load("reference_unet_tiny.safetensors")
    assert fresh.keys() == t.keys()
    for k in t:
        assert torch.equal(fresh[k], t[k]), k


def _check_pipeline_against_reference_fixture(t):
    """oracle/pipeline.py:run == the reference's own StableDiffusionXLInpaintPipeline.__call__ (src/tryon_pipeline.py:1254-1894):
    same UNet weights, the reference's recorded random draws fed in the order SURVEY.md A.4 states (initial latents, masked-image
    posterior, pose posterior [global generator], cloth posterior, one DDPM draw per step with t > 0)."""
    import dataclasses
    from idm_vton_amd import config as pc
    from oracle import pipeline as opipe
    from oracle.vae import AutoencoderKL as OVae, VAEConfig as OVaeCfg
    o_t, o_g, _, _ = _oracle_pair()
    vcfg = pc.VAEConfig(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
    o_v = OVae(OVaeCfg(**{f.name: getattr(vcfg, f.name) for f in dataclasses.fields(OVaeCfg)})).eval()
    o_v.load_state_dict(pc.random_state_dict(pc.vae_param_shapes(vcfg), 103, torch.float32, "cpu", std=0.05))
    draws = [t[f"pipe.draw{i}"] for i in range(sum(k.startswith("pipe.draw") for k in t))]
    n_steps = sum(k.startswith("pipe.latents") for k in t)
    assert len(draws) == 4 + n_steps == 7
    inp = {k[len("pipe.in."):]: v for k, v in t.items() if k.startswith("pipe.in.")}
    inp.pop("clip_pixels")
    tr = {}
    img = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=n_steps, guidance_scale=2.0, trace=tr,
                    ip_hidden_states=t["pipe.ip_hidden_states"],
                    noise=dict(latents=draws[0], masked=draws[1], pose=draws[2], cloth=draws[3], steps=torch.stack(draws[4:])), **inp)
    for i in range(n_steps):
        assert _rel(tr["step_latents"][i], t[f"pipe.latents{i}"]) <= PIN_TOL, (i, _rel(tr["step_latents"][i], t[f"pipe.latents{i}"]))
    assert _rel(img, t["pipe.image"]) <= PIN_TOL, _rel(img, t["pipe.image"])
    # second reference run: strength 0.6 of 5 steps (the last 3 timesteps; the init-image posterior is the FIRST draw, then the
    # latent noise: prepare_latents :883-889) and guidance_scale 1.0 (no classifier-free guidance; conditional IP rows only)
    d2 = [t[f"pipe2.draw{i}"] for i in range(sum(k.startswith("pipe2.draw") for k in t))]
    n2 = sum(k.startswith("pipe2.latents") for k in t)
    assert n2 == 3 and len(d2) == 5 + n2
    B = inp["image"].shape[0]
    tr2 = {}
    img2 = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=5, strength=0.6, guidance_scale=1.0, trace=tr2,
                     ip_hidden_states=t["pipe.ip_hidden_states"][B:],
                     noise=dict(image=d2[0], latents=d2[1], masked=d2[2], pose=d2[3], cloth=d2[4], steps=torch.stack(d2[5:])), **inp)
    for i in range(n2):
        assert _rel(tr2["step_latents"][i], t[f"pipe2.latents{i}"]) <= PIN_TOL, (i, _rel(tr2["step_latents"][i], t[f"pipe2.latents{i}"]))
    assert _rel(img2, t["pipe2.image"]) <= PIN_TOL, _rel(img2, t["pipe2.image"])


def test_oracle_pipeline_matches_reference_code_golden():
    t, _ = _load("reference_unet_tiny.safetensors")
    _check_pipeline_against_reference_fixture(t)
