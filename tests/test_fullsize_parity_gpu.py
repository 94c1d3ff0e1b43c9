"""-m gpu: the HIP path against the ORACLE at BASELINE.json's full sizes (config 2: 768x1024 -> latent 128x96, SDXL-size UNets;
config 4: 1024x1536 -> latent 192x128), same weights, same inputs, same injected noise.

One set of seeded random-init SDXL-architecture weights is generated in HBM (bench.build_engine), rounded to bf16; the oracle
(CPU fp32) loads exactly those values, the product runs them in bf16 and -- the same values, exactly representable -- in fp16.
The oracle runs once per stage (module-scoped fixture: GarmentNet ~8 s, CFG TryonNet ~25 s per forward on the GPU box's host
cores) and both storage dtypes are compared with it:

  C  GarmentNet: the 70 exported norm1 features            (src/unet_hacked_garmnet.py:917-1284)
  D  TryonNet CFG noise prediction, oracle features in      (src/unet_hacked_tryon.py:1006-1395, tryon_pipeline.py:1796-1808)
  E  two full denoising steps from the product's prepared conditioning, injected DDPM noise  (tryon_pipeline.py:1764-1866)
  F  VAE decode at 1024x768 (12 288-token mid attention)    (tryon_pipeline.py:1876)
  G  config 4: GarmentNet features at latent 192x128 (6144 / 1536 tokens)

Bars.  err = max|x - ref| / max|ref|.  The north star asks for <= 1e-3 max-rel latent error "within a stated fp16 tolerance".
Measured on MI355X (profiles/r02_fullsize_parity.json) the fp16 path meets 1e-3 .. 2.5e-3 per stage and the bf16 path
(BASELINE config 2's dtype: 8 mantissa bits, unit roundoff 3.9e-3) 0.6e-2 .. 2e-2; the bars below are those measurements with
~1.5x headroom, NOT the contract value -- DESIGN.md section 5 states where 1e-3 is met and where storage rounding exceeds it.
"""
import dataclasses
import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
H, W = 1024, 768
BARS = {torch.float16: dict(garment_feat_max=5e-3, tryon_eps=6e-3, latents_step=6e-3, vae_decode=8e-3, garment_feat_max_cfg4=5e-3),
        torch.bfloat16: dict(garment_feat_max=3.5e-2, tryon_eps=4e-2, latents_step=4e-2, vae_decode=6e-2, garment_feat_max_cfg4=3.5e-2)}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _rel(x, ref):
    x, ref = x.detach().float().cpu(), ref.detach().float().cpu()
    assert torch.isfinite(x).all() and torch.isfinite(ref).all()
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()


def _ram_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return 1e9


@pytest.fixture(scope="module")
def world():
    """Engines (bf16, fp16), oracle models on the same weight values, seeded inputs, and the oracle's outputs per stage."""
    if _ram_gb() < 64:
        pytest.skip("the SDXL-size fp32 oracle needs ~45 GB of host memory")
    import bench
    from idm_vton_amd import config as pc
    from oracle import unet as ou, vae as ov
    dev = torch.device("cuda", 0)
    t0 = time.time()
    eng_bf, (tcfg, gcfg, vcfg), state = bench.build_engine(torch.bfloat16, dev, 0, 30, return_state=True)
    eng_fp, _ = bench.build_engine(torch.float16, dev, 0, 30, state=state)
    cpu32 = lambda sd: {k: v.float().cpu() for k, v in sd.items()}
    as_o = lambda c, cls: cls(**{f.name: getattr(c, f.name) for f in dataclasses.fields(cls)})

    def oracle(cls, cfg, sd):
        with torch.device("meta"):
            m = cls(cfg)
        m.load_state_dict(cpu32(sd), assign=True)
        return m.eval()
    o_t = oracle(ou.UNet2DConditionModel, as_o(tcfg, ou.UNetConfig), state[0])
    o_g = oracle(ou.UNet2DConditionModel, as_o(gcfg, ou.UNetConfig), state[1])
    o_v = oracle(ov.AutoencoderKL, as_o(vcfg, ov.VAEConfig), state[2])
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    W_ = dict(dev=dev, eng={torch.bfloat16: eng_bf, torch.float16: eng_fp}, oracle=(o_t, o_g, o_v), ref={}, timing={"build_s": time.time() - t0},
              results={})
    yield W_
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "fullsize_parity.json"), "w") as f:
        json.dump(dict(results={str(k): v for k, v in W_["results"].items()}, oracle_seconds=W_["timing"]), f, indent=1)


def _q(t):
    """Values exactly representable in bf16 AND fp16 (normal range), as fp32."""
    return t.to(torch.bfloat16).float()


def _inputs(B, h, w, seed=11):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(z=_q(r(B, 4, h, w)), text_cloth=_q(r(B, 77, 2048)), lmi=_q(r(2 * B, 13, h, w)), pe=_q(r(2 * B, 77, 2048)),
                add_text=_q(r(2 * B, 1280)), ip=_q(r(2 * B, 16, 2048)),
                time_ids=torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32).repeat(2 * B, 1))


@torch.no_grad()
def _oracle_stage_cd(world, h, w, key):
    """Oracle GarmentNet features and (config 2 only) TryonNet eps; cached."""
    if key in world["ref"]:
        return world["ref"][key]
    o_t, o_g, _ = world["oracle"]
    inp = _inputs(1, h, w)
    t0 = time.time()
    _, feats = o_g(inp["z"], 481, inp["text_cloth"])
    world["timing"][f"oracle_garmnet_{key}_s"] = time.time() - t0
    eps = None
    if key == "cfg2":
        t0 = time.time()
        added = dict(text_embeds=inp["add_text"], time_ids=inp["time_ids"], image_embeds=inp["ip"])
        feats_cfg = [torch.cat([torch.zeros_like(d), d]) for d in feats]                       # tryon_pipeline.py:1796
        eps = o_t(inp["lmi"], 481, inp["pe"], added_cond_kwargs=added, garment_features=feats_cfg)[0]
        world["timing"]["oracle_tryon_cfg2_s"] = time.time() - t0
    world["ref"][key] = (inp, feats, eps)
    return world["ref"][key]


@torch.no_grad()
def _product_garment(eng, inp, h, w, dt, dev):
    from idm_vton_amd import ops
    g = eng.unet_encoder
    ctx = g.encode_context(inp["text_cloth"].to(dev))
    temb = g.time_embeddings([481], 1)[0]
    x = ops.to_nhwc(inp["z"].to(dev).contiguous(), dt, cpad=g.cin_pad)
    _, feats = g.forward(x, temb, ctx, 1, h, w)
    return feats


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_config2_garmentnet_features_vs_oracle(world, dt):
    inp, f_o, _ = _oracle_stage_cd(world, 128, 96, "cfg2")
    f_p = _product_garment(world["eng"][dt], inp, 128, 96, dt, world["dev"])
    assert len(f_p) == len(f_o) == 70
    errs = [_rel(a, b) for a, b in zip(f_p, f_o)]
    world["results"].setdefault(dt, {}).update(garment_feat_first=errs[0], garment_feat_last=errs[-1], garment_feat_max=max(errs))
    assert max(errs) <= BARS[dt]["garment_feat_max"], errs


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@torch.no_grad()
def test_config2_tryonnet_eps_vs_oracle(world, dt):
    from idm_vton_amd import ops
    inp, f_o, eps_o = _oracle_stage_cd(world, 128, 96, "cfg2")
    dev, eng = world["dev"], world["eng"][dt]
    t = eng.unet
    ctx = t.encode_context(inp["pe"].to(dev), inp["ip"].to(dev))
    temb = t.time_embeddings([481], 2, dict(text_embeds=inp["add_text"].to(dev), time_ids=inp["time_ids"].to(dev)))[0]
    x = ops.to_nhwc(inp["lmi"].to(dev).contiguous(), dt, cpad=t.cin_pad)
    feats = [d.to(dev, dt).contiguous() for d in f_o]                                      # oracle features in: isolates TryonNet
    eps, _ = t.forward(x, temb, ctx, 2, 128, 96, garment_feats=feats)
    eps = eps.view(2, 128, 96, -1)[..., :4].permute(0, 3, 1, 2)
    e = _rel(eps, eps_o)
    world["results"].setdefault(dt, {}).update(tryon_eps=e)
    assert e <= BARS[dt]["tryon_eps"], e


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@torch.no_grad()
def test_config2_two_denoising_steps_vs_oracle(world, dt):
    """prepare() on the product (VAE encodes, Resampler, conditioning), then two loop iterations on both sides from the SAME
    prepared tensors (the product's, as stored in its storage dtype), same injected DDPM noise: isolates the loop body."""
    import bench
    from oracle import pipeline as opipe
    from oracle.scheduler import Scheduler
    dev, eng = world["dev"], world["eng"][dt]
    o_t, o_g, _ = world["oracle"]
    inp = bench.synth_inputs(1, H, W, 2, dev, 0)
    for k in ("prompt_embeds", "negative_prompt_embeds", "pooled_prompt_embeds", "negative_pooled_prompt_embeds", "text_embeds_cloth", "ip_hidden_states"):
        inp[k] = _q(inp[k].cpu()).to(dev)
    st = eng.prepare(num_inference_steps=2, guidance_scale=2.0, scheduler="ddpm", **inp)
    h, w = st["h"], st["w"]
    lat0 = st["latents"].clone().cpu()
    cond = st["cond"].float().cpu().view(2, h, w, 9).permute(0, 3, 1, 2)                    # [2B][9][h][w]: mask | masked | pose
    cloth = st["cloth"].float().cpu().view(1, h, w, -1)[..., :4].permute(0, 3, 1, 2)
    tr = {}
    lat_p = eng.denoise(st, trace=tr)
    key = ("steps", dt)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]]).cpu()
    added = dict(text_embeds=torch.cat([inp["negative_pooled_prompt_embeds"], inp["pooled_prompt_embeds"]]).cpu(),
                 time_ids=torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32).repeat(2, 1),
                 image_embeds=st["trace"]["image_embeds"].float().cpu())
    sched = Scheduler("ddpm")
    ts = sched.set_timesteps(2)
    assert list(ts) == list(st["timesteps"])
    t0 = time.time()
    otr = dict(step_latents=[], step_eps=[])
    opipe.denoise(o_t, o_g, sched, ts, lat0, cond[:, :1], cond[:, 1:5], cond[:, 5:9], cloth, pe, added, inp["text_embeds_cloth"].cpu(),
                  2.0, inp["noise"]["steps"].cpu(), otr)
    world["timing"][f"oracle_two_steps_{dt}_s"] = time.time() - t0
    errs = [_rel(a, b) for a, b in zip(tr["step_latents"], otr["step_latents"])]
    world["results"].setdefault(dt, {}).update(latents_step1=errs[0], latents_step2=errs[1])
    assert max(errs) <= BARS[dt]["latents_step"], errs


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@torch.no_grad()
def test_config2_vae_decode_vs_oracle(world, dt):
    """1024x768 decode: 12 288-token single-head mid attention, 128..512-channel convs at full resolution."""
    _, _, o_v = world["oracle"]
    if "vae" not in world["ref"]:
        g = torch.Generator().manual_seed(5)
        z = _q(torch.randn(1, 4, 128, 96, generator=g))
        t0 = time.time()
        world["ref"]["vae"] = (z, o_v.decode(z))
        world["timing"]["oracle_vae_decode_s"] = time.time() - t0
    z, img_o = world["ref"]["vae"]
    img_p = world["eng"][dt].vae.decode(z.to(world["dev"]))
    e = _rel(img_p, img_o)
    world["results"].setdefault(dt, {}).update(vae_decode=e)
    assert e <= BARS[dt]["vae_decode"], e


@pytest.mark.parametrize("dt", [torch.bfloat16], ids=["bf16"])
def test_config4_garmentnet_features_vs_oracle(world, dt):
    """BASELINE.json configs[3]: 1024x1536 (latent 192x128: 6144 / 1536 tokens), one GarmentNet forward against the oracle."""
    inp, f_o, _ = _oracle_stage_cd(world, 192, 128, "cfg4")
    f_p = _product_garment(world["eng"][dt], inp, 192, 128, dt, world["dev"])
    errs = [_rel(a, b) for a, b in zip(f_p, f_o)]
    world["results"].setdefault(dt, {}).update(garment_feat_max_cfg4=max(errs))
    assert f_p[0].shape == (1, 6144, 640) and f_p[-1].shape == (1, 6144, 640)
    assert max(errs) <= BARS[dt]["garment_feat_max_cfg4"], errs
