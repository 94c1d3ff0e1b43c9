"""Full-size parity legs (BASELINE.json configs[1] and configs[3]) shared by tests/test_fullsize_parity_gpu.py and
tools/gpu_parity_table.py.  TEST INFRASTRUCTURE: imports oracle/.

Every number is  err(x, ref)  against the fp32 ORACLE (oracle/unet.py, oracle/pipeline.py, oracle/vae.py) holding the SAME weight
values (one seeded bf16-rounded set, exactly representable in fp16 and fp32) and fed the SAME inputs and injected noise.  Legs:

  ref_fp16   the REFERENCE'S OWN NUMERIC POLICY, measured, not assumed: the oracle modules in fp16 under torch.autocast(fp16),
             exactly as /root/reference/inference.py:233-262 (weights .to(float16)) and :339 (`with torch.cuda.amp.autocast()`)
             run the model: Linear / conv / attention in fp16 with fp32 accumulation, LayerNorm / GroupNorm / softmax in fp32,
             residual adds and the latents in fp16 (pipeline dtype).  SDPA is the fused-kernel policy (fp32 logits and softmax, P
             rounded to fp16 for PV) the reference gets from F.scaled_dot_product_attention (ip_adapter/attention_processor.py:258).
             Its error against the fp32 oracle IS "the stated fp16 tolerance" the north star refers to; the fp16 HIP path must not
             exceed it (bars below are multiples of this leg, not of the implementation's own numbers).
  hip_f16, hip_bf16          the product, storage fp16 / bf16 (default: the transformer residual stream is rounded to the storage dtype
                             after every add, as the reference's autocast does)
  hip_f16_s32, hip_bf16_s32  the product with the fp32 residual stream inside each Transformer2DModel (HipUNet(stream_f32=True): A/B)
  hip_f16_fp8                BASELINE.json configs[4]: fp16 storage, every self-attention on e4m3 operands (HipUNet(attn_fp8=True),
                             csrc/attention_f8.hip); e4m3 has 3 mantissa bits, so this leg carries its own measured bars

The fp32 oracle is EXECUTED ON THE GPU by torch (rocBLAS fp32 GEMMs, torch's own im2col convolution -- MIOpen is switched off so a
fresh box does not spend minutes in kernel search) so that 30-step, batch-2 and 192x128-latent comparisons take seconds instead of
the 20+ minutes the CPU execution needs; `anchor` holds that execution to the CPU execution of the same module on the GarmentNet
forward (fp32 summation-order differences only).

Metrics: rel = max|x - ref| / max|ref|  (the metric of every kernel check);  rms = ||x - ref||_2 / ||ref||_2.
"""
import contextlib
import dataclasses
import json
import os
import time

import torch

H2, W2 = 1024, 768            # config 2 (latent 128 x 96)
H4, W4 = 1536, 1024           # config 4 (latent 192 x 128)


def rel(x, ref):
    x, ref = x.detach().double(), ref.detach().double().to(x.device)
    if not bool(torch.isfinite(x).all()):
        return float("inf")
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def rms(x, ref):
    x, ref = x.detach().double(), ref.detach().double().to(x.device)
    if not bool(torch.isfinite(x).all()):
        return float("inf")
    return ((x - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def _q(t):
    """Values exactly representable in bf16 AND fp16 (normal range), as fp32."""
    return t.to(torch.bfloat16).float()


def _fused_policy_sdpa(q, k, v):
    """What a fused attention kernel computes on 16-bit inputs (the reference's F.scaled_dot_product_attention): fp32 logits and
    softmax, probabilities rounded to the input dtype for the PV product, fp32 accumulation, one rounding of the output."""
    with torch.autocast("cuda", enabled=False):
        s = (q.float() @ k.float().transpose(-2, -1)) * (q.shape[-1] ** -0.5)
        p = torch.softmax(s, dim=-1).to(q.dtype)
        return (p.float() @ v.float()).to(q.dtype)


@contextlib.contextmanager
def ref_policy():
    """Run oracle modules the way inference.py:339 runs the reference's: fp16 autocast (+ the fused-SDPA numerics)."""
    from oracle import layers
    keep = layers.sdpa
    layers.sdpa = _fused_policy_sdpa
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            yield
    finally:
        layers.sdpa = keep


# leg -> (storage dtype, fp32 residual stream, self-attention on e4m3 operands)
LEGS = {"hip_bf16": (torch.bfloat16, False, False), "hip_f16": (torch.float16, False, False),
        "hip_bf16_s32": (torch.bfloat16, True, False), "hip_f16_s32": (torch.float16, True, False),
        # BASELINE.json configs[4] ("fp16 + fp8 MFMA attention"; the pipeline call is /root/reference/inference_dc.py:550, same shapes as config 2)
        "hip_f16_fp8": (torch.float16, False, True)}


class World:
    """Engines for every leg, oracle modules (fp32 and fp16 copies) on the device, one weight set."""

    def __init__(self, dev, legs=tuple(LEGS), log=print):
        import bench
        from oracle import unet as ou, vae as ov
        self.dev, self.log = dev, log
        self.results, self.timing = {}, {}
        torch.backends.cudnn.enabled = False               # oracle convs: torch's own kernels, no MIOpen search / JIT (the product uses neither)
        t0 = time.time()
        eng0, cfgs, state = bench.build_engine(torch.bfloat16, dev, 0, 30, return_state=True, stream_f32=False)
        self.eng = {}
        for name in legs:
            dt, s32, f8 = LEGS[name]
            self.eng[name] = eng0 if name == "hip_bf16" else bench.build_engine(dt, dev, 0, 30, state=state, stream_f32=s32, attn_fp8=f8)[0]
        tcfg, gcfg, vcfg = cfgs
        as_o = lambda c, cls: cls(**{f.name: getattr(c, f.name) for f in dataclasses.fields(cls)})

        def oracle(cls, cfg, sd, dtype):
            with torch.device("meta"):
                m = cls(cfg)
            m.load_state_dict({k: v.to(dev, dtype) for k, v in sd.items()}, assign=True)
            return m.eval()
        self.state = state
        self.ocfg = (as_o(tcfg, ou.UNetConfig), as_o(gcfg, ou.UNetConfig), as_o(vcfg, ov.VAEConfig))
        self.o32 = (oracle(ou.UNet2DConditionModel, self.ocfg[0], state[0], torch.float32),
                    oracle(ou.UNet2DConditionModel, self.ocfg[1], state[1], torch.float32),
                    oracle(ov.AutoencoderKL, self.ocfg[2], state[2], torch.float32))
        self.o16 = (oracle(ou.UNet2DConditionModel, self.ocfg[0], state[0], torch.float16),
                    oracle(ou.UNet2DConditionModel, self.ocfg[1], state[1], torch.float16))
        self.timing["build_s"] = time.time() - t0
        self.cache = {}

    def put(self, stage, leg, **kv):
        self.results.setdefault(stage, {}).setdefault(leg, {}).update(kv)
        self.log(f"[parity] {stage:28s} {leg:14s} " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kv.items()))

    def dump(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        import bench
        with open(path, "w") as f:
            json.dump(dict(results=self.results, seconds=self.timing, source_hash=bench.kernel_source_hash(),
                           metric="rel = max|x-ref|/max|ref|, rms = ||x-ref||/||ref||; ref = fp32 oracle, same weights / inputs / noise"), f, indent=1)


def _unet_inputs(B, h, w, H, W, seed=11):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(z=_q(r(B, 4, h, w)), text_cloth=_q(r(B, 77, 2048)), lmi=_q(r(2 * B, 13, h, w)), pe=_q(r(2 * B, 77, 2048)),
                add_text=_q(r(2 * B, 1280)), ip=_q(r(2 * B, 16, 2048)),
                time_ids=torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32).repeat(2 * B, 1))


def _errs(xs, refs):
    return dict(rel=max(rel(a, b) for a, b in zip(xs, refs)), rms=max(rms(a, b) for a, b in zip(xs, refs)))


# ------------------------------------------------------------------------------------------------ anchor: GPU- vs CPU-executed oracle
@torch.no_grad()
def stage_anchor(Wd):
    """The fp32 oracle executed by torch on the GPU == the same module executed on the CPU (GarmentNet, config 2, 70 features)."""
    from oracle import unet as ou
    inp = _unet_inputs(1, 128, 96, H2, W2)
    t0 = time.time()
    with torch.device("meta"):
        cpu = ou.UNet2DConditionModel(Wd.ocfg[1])
    cpu.load_state_dict({k: v.float().cpu() for k, v in Wd.state[1].items()}, assign=True)
    cpu.eval()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    _, f_cpu = cpu(inp["z"], 481, inp["text_cloth"])
    Wd.timing["anchor_cpu_garmnet_s"] = time.time() - t0
    del cpu
    t0 = time.time()
    _, f_gpu = Wd.o32[1](inp["z"].to(Wd.dev), 481, inp["text_cloth"].to(Wd.dev))
    torch.cuda.synchronize()
    Wd.timing["anchor_gpu_garmnet_s"] = time.time() - t0
    Wd.put("anchor_gpu_vs_cpu_oracle", "oracle_fp32", **_errs(f_gpu, f_cpu))


# ------------------------------------------------------------------------------------------------ C / D: single forwards
@torch.no_grad()
def stage_unets(Wd, key, h, w, H, W, B=1, tryon=True):
    """C: GarmentNet's 70 exported features (src/unet_hacked_garmnet.py:917-1284);  D: TryonNet CFG noise prediction with the
    oracle's features in (src/unet_hacked_tryon.py:1006-1395, tryon_pipeline.py:1796-1808)."""
    from idm_vton_amd import ops
    dev = Wd.dev
    inp = {k: v.to(dev) for k, v in _unet_inputs(B, h, w, H, W).items()}
    o_t, o_g, _ = Wd.o32
    t0 = time.time()
    _, f32 = o_g(inp["z"], 481, inp["text_cloth"])
    added = dict(text_embeds=inp["add_text"], time_ids=inp["time_ids"], image_embeds=inp["ip"])
    eps32 = None
    if tryon:
        fcfg = [torch.cat([torch.zeros_like(d), d]) for d in f32]                          # tryon_pipeline.py:1796
        eps32 = o_t(inp["lmi"], 481, inp["pe"], added_cond_kwargs=added, garment_features=fcfg)[0]
        del fcfg
    torch.cuda.synchronize()
    Wd.timing[f"oracle32_{key}_s"] = time.time() - t0
    # reference policy
    hf = lambda t: t.half()
    with ref_policy():
        _, f16 = Wd.o16[1](hf(inp["z"]), 481, hf(inp["text_cloth"]))
        Wd.put(f"{key}_garment_features", "ref_fp16", **_errs(f16, f32))
        if tryon:
            fc16 = [torch.cat([torch.zeros_like(d), d]).half() for d in f32]
            a16 = dict(text_embeds=hf(inp["add_text"]), time_ids=inp["time_ids"], image_embeds=hf(inp["ip"]))
            e16 = Wd.o16[0](hf(inp["lmi"]), 481, hf(inp["pe"]), added_cond_kwargs=a16, garment_features=fc16)[0]
            Wd.put(f"{key}_tryon_eps", "ref_fp16", rel=rel(e16, eps32), rms=rms(e16, eps32))
            del fc16, e16
    del f16
    for leg, eng in Wd.eng.items():
        dt = LEGS[leg][0]
        g, t = eng.unet_encoder, eng.unet
        ctx = g.encode_context(inp["text_cloth"])
        temb = g.time_embeddings([481], B)[0]
        x = ops.to_nhwc(inp["z"].contiguous(), dt, cpad=g.cin_pad)
        _, fp = g.forward(x, temb, ctx, B, h, w)
        assert len(fp) == len(f32) == 70
        Wd.put(f"{key}_garment_features", leg, **_errs(fp, f32))
        del fp
        if tryon:
            ctx = t.encode_context(inp["pe"], inp["ip"])
            temb = t.time_embeddings([481], 2 * B, dict(text_embeds=inp["add_text"], time_ids=inp["time_ids"]))[0]
            x = ops.to_nhwc(inp["lmi"].contiguous(), dt, cpad=t.cin_pad)
            feats = [d.to(dt).contiguous() for d in f32]                                   # oracle features in: isolates TryonNet
            eps, _ = t.forward(x, temb, ctx, 2 * B, h, w, garment_feats=feats)
            eps = eps.view(2 * B, h, w, -1)[..., :4].permute(0, 3, 1, 2)
            Wd.put(f"{key}_tryon_eps", leg, rel=rel(eps, eps32), rms=rms(eps, eps32))
            del feats, eps
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ E: the loop
@torch.no_grad()
def stage_loop(Wd, key, H, W, B, steps, scheduler, record=(0, 1, -1)):
    """`steps` full loop iterations (tryon_pipeline.py:1764-1866) on every leg from ONE prepared state: the bf16 engine's prepare()
    output (VAE encodes, Resampler, preprocessing), stored in bf16 -- exactly representable on every leg -- with injected DDPM noise."""
    import bench
    from oracle import pipeline as opipe
    from oracle.scheduler import Scheduler
    dev = Wd.dev
    inp = bench.synth_inputs(B, H, W, steps, dev, 0)
    for k in ("prompt_embeds", "negative_prompt_embeds", "pooled_prompt_embeds", "negative_pooled_prompt_embeds", "text_embeds_cloth", "ip_hidden_states"):
        inp[k] = _q(inp[k].cpu()).to(dev)
    base_eng = Wd.eng["hip_bf16"]
    st0 = base_eng.prepare(num_inference_steps=steps, guidance_scale=2.0, scheduler=scheduler, **inp)
    h, w = st0["h"], st0["w"]
    base = dict(latents=st0["latents"].clone(), cond=st0["cond"].clone(), cloth=st0["cloth"].clone(),
                image_embeds=st0["trace"]["image_embeds"].clone())
    cond = base["cond"].float().view(2 * B, h, w, 9).permute(0, 3, 1, 2).contiguous()      # [2B][9][h][w]: mask | masked | pose
    cloth = base["cloth"].float().view(B, h, w, -1)[..., :4].permute(0, 3, 1, 2).contiguous()
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
    added = dict(text_embeds=torch.cat([inp["negative_pooled_prompt_embeds"], inp["pooled_prompt_embeds"]]),
                 time_ids=torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32, device=dev).repeat(2 * B, 1),
                 image_embeds=base["image_embeds"].float())
    sched = Scheduler(scheduler)
    ts = sched.set_timesteps(steps)
    assert list(ts) == list(st0["timesteps"])
    nz = inp["noise"]["steps"] if scheduler == "ddpm" else None
    o_t, o_g, _ = Wd.o32
    t0 = time.time()
    tr32 = dict(step_latents=[], step_eps=[])
    opipe.denoise(o_t, o_g, sched, ts, base["latents"].clone(), cond[:, :1], cond[:, 1:5], cond[:, 5:9], cloth, pe, added,
                  inp["text_embeds_cloth"], 2.0, nz, tr32)
    torch.cuda.synchronize()
    Wd.timing[f"oracle32_{key}_s"] = time.time() - t0
    ref = tr32["step_latents"]
    idx = sorted({i % steps for i in record})
    rep = lambda lats: {f"step{i + 1}": dict(rel=rel(lats[i], ref[i]), rms=rms(lats[i], ref[i])) for i in idx}

    def flat(d):
        out = {}
        for s, m in d.items():
            for k, v in m.items():
                out[f"{k}_{s}"] = v
        out["rel"] = max(m["rel"] for m in d.values())
        out["rms"] = max(m["rms"] for m in d.values())
        return out
    # reference policy: fp16 modules + autocast, fp16 latents and conditioning, scheduler arithmetic in fp16 (pipeline dtype fp16)
    t0 = time.time()
    hf = lambda t: None if t is None else t.half()
    tr16 = dict(step_latents=[], step_eps=[])
    with ref_policy():
        a16 = dict(text_embeds=hf(added["text_embeds"]), time_ids=added["time_ids"], image_embeds=hf(added["image_embeds"]))
        opipe.denoise(Wd.o16[0], Wd.o16[1], sched, ts, hf(base["latents"]), hf(cond[:, :1]), hf(cond[:, 1:5]), hf(cond[:, 5:9]), hf(cloth),
                      hf(pe), a16, hf(inp["text_embeds_cloth"]), 2.0, hf(nz), tr16)
    torch.cuda.synchronize()
    Wd.timing[f"ref_fp16_{key}_s"] = time.time() - t0
    Wd.put(f"{key}_latents", "ref_fp16", **flat(rep(tr16["step_latents"])))
    del tr16
    inp_p = {k: v for k, v in inp.items() if k != "ip_hidden_states"}
    for leg, eng in Wd.eng.items():
        dt = LEGS[leg][0]
        st = eng.prepare(num_inference_steps=steps, guidance_scale=2.0, scheduler=scheduler, image_embeds=base["image_embeds"].to(dt), **inp_p)
        st["latents"].copy_(base["latents"])
        st["cond"].copy_(base["cond"])
        st["cloth"].copy_(base["cloth"])
        st["cloth_k"].copy_(base["cloth"].repeat(st["k"], 1, 1))
        tr = {}
        eng.denoise(st, trace=tr)
        torch.cuda.synchronize()
        Wd.put(f"{key}_latents", leg, **flat(rep(tr["step_latents"])))
        del st, tr
    return


# ------------------------------------------------------------------------------------------------ F / G: VAE
@torch.no_grad()
def stage_vae(Wd, H, W, B=1):
    """F: decode of a 128x96 latent to 1024x768 (tryon_pipeline.py:1876); G: encode of a 1024x768 image + posterior sample
    (tryon_pipeline.py:924,1646,1654).  The reference runs its VAE in fp32 for the masked-image / cloth encodes and the decode
    (force_upcast, :913-927,1870-1880), so there is no reduced-precision reference policy to measure here."""
    dev = Wd.dev
    o_v = Wd.o32[2]
    g = torch.Generator().manual_seed(5)
    z = _q(torch.randn(B, 4, H // 8, W // 8, generator=g)).to(dev)
    img = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(dev)
    nz = torch.randn(B, 4, H // 8, W // 8, generator=g).to(dev)
    t0 = time.time()
    dec32 = o_v.decode(z)
    enc32 = o_v.encode_sample(img, nz) * o_v.cfg.scaling_factor
    torch.cuda.synchronize()
    Wd.timing["oracle32_vae_s"] = time.time() - t0
    for leg, eng in Wd.eng.items():
        if leg.endswith(("_s32", "_fp8")):
            continue                                          # the VAE has no transformer stream / fp8 attention: same kernels as the default legs
        d = eng.vae.decode(z)
        Wd.put("cfg2_vae_decode", leg, rel=rel(d, dec32), rms=rms(d, dec32))
        e = eng.vae.encode_sample(img, nz)
        Wd.put("cfg2_vae_encode_sample", leg, rel=rel(e, enc32), rms=rms(e, enc32))
        del d, e
    torch.cuda.synchronize()


def run_all(Wd, out_path=None, stages=("anchor", "cfg2_unets", "cfg2_b2_2steps", "cfg2_30steps", "cfg2_b2_ddpm30", "vae", "cfg4_unets", "cfg4_1step")):
    def guard(name, fn, *a, **k):
        if name not in stages:
            return
        t0 = time.time()
        try:
            fn(*a, **k)
        except Exception as e:                                # keep going: every stage's numbers are worth a GPU-minute
            import traceback
            traceback.print_exc()
            Wd.results.setdefault("_errors", {})[name] = repr(e)
        Wd.timing[f"stage_{name}_s"] = time.time() - t0
        torch.cuda.empty_cache()
        if out_path:
            Wd.dump(out_path)
    guard("anchor", stage_anchor, Wd)
    guard("cfg2_unets", stage_unets, Wd, "cfg2", 128, 96, H2, W2)
    guard("cfg2_b2_2steps", stage_loop, Wd, "cfg2_b2_ddpm2", H2, W2, 2, 2, "ddpm", record=(0, 1))
    guard("cfg2_30steps", stage_loop, Wd, "cfg2_b1_ddim30", H2, W2, 1, 30, "ddim", record=(0, 9, 19, 29))
    # the operating point of the script the boundary drops into: /root/reference/inference.py:232 builds DDPMScheduler and :397-414 runs 30
    # ancestral steps at batch 2 (CFG batch 4) -- injected per-step noise, latents against the oracle at steps 1, 10, 20, 30
    guard("cfg2_b2_ddpm30", stage_loop, Wd, "cfg2_b2_ddpm30", H2, W2, 2, 30, "ddpm", record=(0, 9, 19, 29))
    guard("vae", stage_vae, Wd, H2, W2)
    guard("cfg4_unets", stage_unets, Wd, "cfg4", 192, 128, H4, W4)
    guard("cfg4_1step", stage_loop, Wd, "cfg4_b1_ddpm1", H4, W4, 1, 1, "ddpm", record=(0,))
    return Wd.results
