"""-m gpu: model-level parity of the HIP engine against the oracle (same weights, inputs and injected noise).

Tolerances are stated per storage dtype for err = max|x - ref| / max|ref|.  The north-star asks for
<= 1e-3 max-rel latent error "within a stated fp16 tolerance": single kernels meet ~3e-4 in fp16 (tests/test_kernels_gpu.py);
through ~50 chained blocks and 4 full denoising steps the fp16 engine stays within 5e-3 of the fp32 oracle, bf16 within 4e-2.
DESIGN.md records the measured figures."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# `image` = the VAE decode of latents that already differ by `latents`: the random-init test VAE (std 0.05 weights, 4 levels) roughly
# doubles a relative latent difference.  The decode itself runs the split-precision path (idm_vton_amd/vae.py) and is held to its own bar on
# IDENTICAL latents (`decode`), so the image bar only has to cover the amplified latent drift: round 4 took it back to the round-2 values
# (round 3 had widened it to 1.5e-2 / 1e-1 together with numeric changes -- ADVICE r3).
TOL = {torch.float16: dict(stage=4e-3, latents=5e-3, image=1e-2, decode=2e-4),
       torch.bfloat16: dict(stage=3e-2, latents=4e-2, image=6e-2, decode=2e-4)}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_tiny_pipeline_parity(dtype):
    from tests import parity_checks
    r = parity_checks.run("tiny", dtype, B=1, H=128, W=128, steps=4)
    t = TOL[dtype]
    for k in ("resampler", "vae_encode", "garment_feat_max", "tryon_eps", "tryon_eps_materialised_zeros",
              "prep_masked_lat", "prep_pose_lat"):
        assert r[k] <= t["stage"], (k, r)
    assert r["vae_decode"] <= t["decode"], r                 # same latents in: the decode alone, fp32-equivalent on every engine
    assert r["closed_form_vs_materialised"] <= t["stage"], r
    assert r["latents_final"] <= t["latents"], r
    assert r["image"] <= t["image"], r


@pytest.mark.parametrize("H,W", [(1000, 760), (320, 320), (264, 200)], ids=["1000x760", "320x320", "264x200"])
def test_sizes_the_reference_accepts_match_the_oracle(H, W):
    """VERDICT r3 item 7.  Any H x W divisible by 8 runs, as in the reference: 1000x760 -> latent 125x95 -> 63x48 -> 32x24 (odd levels: the
    up path needs diffusers' `upsample_size`, src/unet_hacked_tryon.py:1084-1090,1357-1379 -- fused into the upsampler convolution's gather);
    320x320 -> 40x40 -> 20x20 -> 10x10 = 100 tokens at the coarsest level (not a multiple of 16: token rows padded inside the transformer, the
    filler masked as keys); 264x200 -> 33x25 -> 17x13 -> 9x7: both at once.  Every stage against the oracle, fp16 bars of the 128x128 test."""
    from tests import parity_checks
    r = parity_checks.run("tiny", torch.float16, B=1, H=H, W=W, steps=2)
    t = TOL[torch.float16]
    for k in ("vae_encode", "garment_feat_max", "tryon_eps", "tryon_eps_materialised_zeros", "prep_masked_lat", "prep_pose_lat"):
        assert r[k] <= t["stage"], (k, r)
    assert r["vae_decode"] <= t["decode"] and r["latents_final"] <= t["latents"] and r["image"] <= t["image"], r


def test_tiny_pipeline_graph_replay_matches_eager():
    """hipGraph replay of the captured step gives the same latents as eager launches."""
    from tests import parity_checks
    r = parity_checks.run("tiny", torch.float16, B=2, H=128, W=128, steps=3, use_graph=True)
    assert r["latents_final"] <= TOL[torch.float16]["latents"], r


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_two_stream_overlap_matches_serial(use_graph):
    """GarmentNet(step i+1) on a second HIP stream overlapped with TryonNet(step i): same latents as the serial loop,
    eager and as hipGraphs with parallel branches; odd step count exercises both set parities + the tail."""
    from tests import parity_checks
    r = parity_checks.run("tiny", torch.float16, B=2, H=128, W=128, steps=5, use_graph=use_graph, overlap=True)
    # same kernels on the same data, and no kernel uses atomics: identical bits
    assert r["overlap_vs_serial"] == 0.0 and r["overlap_second_call"] == 0.0, r
    assert r["latents_final"] <= TOL[torch.float16]["latents"], r


def test_serial_loop_is_bit_reproducible():
    """No kernel on the path uses atomics or depends on scheduling: two runs of the same pipeline give identical latents."""
    from idm_vton_amd.pipeline import TryonEngine
    from tests import parity_utils as pu
    dt = torch.float16
    m = pu.build("tiny", dt, "cuda")
    p_t, p_g, p_v, p_r = m["product"]
    inp = pu.make_inputs(2, 128, 128, m["xd"], m["pooled"], m["enc_dim"], 3, dt)
    eng = TryonEngine(p_t, p_g, p_v, p_r, dt, "cuda")
    outs = []
    for use_graph in (False, False, True, True):
        st = eng.prepare(num_inference_steps=3, guidance_scale=2.0, scheduler="ddpm", **inp)
        outs.append(eng.denoise(st, use_graph=use_graph).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def test_garment_timestep_batching_does_not_change_results():
    """GarmentNet runs for `garment_steps` consecutive timesteps in one batch (idm_vton_amd/pipeline.py): per (image, timestep)
    the arithmetic is the reference's one call per step (tryon_pipeline.py:1781-1787).  Every batching factor -- including blocks
    that do not divide the step count -- must give the latents of the one-call-per-step form, in all four execution forms."""
    from idm_vton_amd.pipeline import TryonEngine
    from tests import parity_utils as pu
    dt = torch.float16
    m = pu.build("tiny", dt, "cuda")
    p_t, p_g, p_v, p_r = m["product"]
    inp = pu.make_inputs(2, 128, 128, m["xd"], m["pooled"], m["enc_dim"], 5, dt)
    outs = {}
    for k in (1, 2, 3, 5, 8):
        eng = TryonEngine(p_t, p_g, p_v, p_r, dt, "cuda")
        eng.garment_steps = k
        for kw in (dict(), dict(use_graph=True), dict(overlap=True), dict(use_graph=True, overlap=True)):
            st = eng.prepare(num_inference_steps=5, guidance_scale=2.0, scheduler="ddpm", **inp)
            outs[(k, tuple(sorted(kw)))] = eng.denoise(st, **kw).clone()
    ref = outs[(1, ())]
    for key, o in outs.items():
        if key[0] == 1:
            assert torch.equal(o, ref), key                      # execution forms are bit-identical
        assert torch.equal(o, outs[(key[0], ())]), key           # ... for every batching factor
        # across factors only GroupNorm's partial-sum blocking (a function of the batch size) may differ, in the last bit
        assert pu.relerr(o, ref) < 1e-3, (key, pu.relerr(o, ref))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_mid_pipeline_parity(dtype):
    """The wider configuration of tests/parity_utils.py (320/640/1280 channels, 5/10/20 heads: the SDXL widths, 128x128 GEMM
    tiles, group size 10) through every stage against the oracle."""
    from tests import parity_checks
    r = parity_checks.run("mid", dtype, B=1, H=128, W=128, steps=2)
    t = TOL[dtype]
    for k in ("resampler", "vae_encode", "vae_decode", "garment_feat_max", "tryon_eps", "tryon_eps_materialised_zeros"):
        assert r[k] <= t["stage"], (k, r)
    assert r["latents_final"] <= t["latents"] and r["image"] <= t["image"], r


@pytest.mark.parametrize("case", ["strength0.6", "guidance1.0", "guidance0.5_strength0.8"])
def test_strength_and_no_cfg_branches_match_oracle(case):
    """The two `__call__` branches the try-on scripts do not take: strength < 1 (tryon_pipeline.py:987-995 timesteps subset,
    :883-893 start from add_noise(encode(image))) and guidance_scale <= 1 (:440-442 no classifier-free guidance: the oracle runs
    the conditional branch alone, the engine its batched step with guidance 1)."""
    from idm_vton_amd.pipeline import TryonEngine
    from oracle import pipeline as opipe
    from oracle.scheduler import Scheduler
    from tests import parity_utils as pu
    strength = 0.6 if case == "strength0.6" else (0.8 if "strength0.8" in case else 1.0)
    g = 1.0 if case == "guidance1.0" else (0.5 if "guidance0.5" in case else 2.0)
    dt, B, steps = torch.float16, 1, 5
    m = pu.build("tiny", dt, "cuda")
    o_t, o_g, o_v = m["oracle"]
    p_t, p_g, p_v, p_r = m["product"]
    inp = pu.make_inputs(B, 128, 128, m["xd"], m["pooled"], m["enc_dim"], steps, dt)
    inp["noise"]["image"] = torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(77))
    n_exec = min(int(steps * strength), steps)
    inp["noise"]["steps"] = inp["noise"]["steps"][:n_exec]
    o_inp = dict(inp)
    p_inp = dict(inp)
    if g <= 1:                                            # the reference hands the pipeline the conditional rows only
        o_inp["ip_hidden_states"] = inp["ip_hidden_states"][B:]
        p_inp["ip_hidden_states"] = inp["ip_hidden_states"][B:]
        p_inp["negative_prompt_embeds"] = p_inp["negative_pooled_prompt_embeds"] = None
    tr = {}
    lat_o = opipe.run(o_t, o_g, o_v, Scheduler("ddpm"), num_inference_steps=steps, guidance_scale=g, strength=strength,
                      return_latents=True, trace=tr, **o_inp)
    assert len(tr["step_latents"]) == n_exec
    eng = TryonEngine(p_t, p_g, p_v, p_r, dt, "cuda")
    for kw in (dict(), dict(use_graph=True, overlap=True)):
        st = eng.prepare(num_inference_steps=steps, guidance_scale=g, scheduler="ddpm", strength=strength, **p_inp)
        assert len(st["timesteps"]) == n_exec
        assert pu.relerr(st["latents"], tr["latents0"]) < 2e-3, "start latents (add_noise of the encoded image)"
        lat_p = eng.denoise(st, **kw)
        assert pu.relerr(lat_p, lat_o) <= TOL[dt]["latents"], (case, kw, pu.relerr(lat_p, lat_o))


@pytest.mark.parametrize("opt", ["stream_f32"])
def test_engine_options_match_oracle(opt):
    """The HipUNet option that is off by default -- the fp32 residual stream (stream_f32) -- through every stage against the oracle, same
    bars as the default engine (the kernels behind it are checked tile by tile in tests/kernel_checks.py: check_stream_f32)."""
    from tests import parity_checks
    for dtype, kw in ((torch.float16, dict()), (torch.bfloat16, dict()), (torch.float16, dict(use_graph=True, overlap=True))):
        t = TOL[dtype]
        r = parity_checks.run("tiny", dtype, B=2, H=128, W=128, steps=3, unet_kw={opt: True}, **kw)
        for k in ("garment_feat_max", "tryon_eps", "tryon_eps_materialised_zeros"):
            assert r[k] <= t["stage"], (k, r)
        assert r["latents_final"] <= t["latents"] and r["image"] <= t["image"], r


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_time_and_added_embeddings_match_oracle(dtype):
    """SURVEY.md 8a row a13 on its own (src/unet_hacked_tryon.py:1118-1213): Timesteps(flip_sin_to_cos, shift 0) -> time_embedding,
    the six add_time_ids through Timesteps(256) in the reference's ORDER, concatenated after the pooled text embedding ->
    add_embedding, summed, and every ResnetBlock2D's time_emb_proj(silu(emb)) -- the engine computes all of it for all timesteps in
    one table before the loop.  Distinct time-id values and both ends of the schedule, so a flipped sin/cos half, a swapped id or a
    wrong frequency denominator cannot cancel."""
    import torch.nn.functional as F
    from tests import parity_utils as pu
    m = pu.build("tiny", dtype, "cuda")
    o_t, o_g, _ = m["oracle"]
    p_t, p_g, _, _ = m["product"]
    B2 = 2
    g = torch.Generator().manual_seed(9)
    add_text = torch.randn(B2, m["pooled"], generator=g).to(dtype).float()
    time_ids = torch.tensor([[1024.0, 768.0, 3.0, 17.0, 512.0, 384.0], [96.0, 1536.0, 0.0, 5.0, 640.0, 8.0]])
    ts = [999, 481, 34, 1]
    for net_p, net_o, added_p, added_o in ((p_t, o_t, dict(text_embeds=add_text.cuda(), time_ids=time_ids.cuda()), dict(text_embeds=add_text, time_ids=time_ids)),
                                          (p_g, o_g, None, None)):
        table = net_p.time_embeddings(ts, B2, added_p).float().cpu()                  # [steps][B][sum of Cout]
        sample = torch.zeros(B2, 4, 8, 8)
        for si, t in enumerate(ts):
            emb = net_o.time_embed(sample, t, added_o)
            for name, (off, co) in net_p.temb_slices.items():
                ref = net_o.get_submodule(name).time_emb_proj(F.silu(emb))
                e = pu.relerr(table[si, :, off:off + co], ref)
                assert e <= (4e-3 if dtype == torch.float16 else 3e-2), (name, t, e)


def test_fp8_attention_engine_matches_oracle_within_its_stated_tolerance():
    """HipUNet(attn_fp8=True) (BASELINE.json configs[4]): e4m3 carries 3 mantissa bits, each self-attention output is within ~3e-2 of
    fp32 (tests/kernel_checks.py::check_attn_f8); through the tiny pipeline the TryonNet noise prediction and the latents after 3
    steps stay within 8e-2 of the fp32 oracle.  The stated tolerance of this VARIANT, not of the default engine."""
    from tests import parity_checks
    for kw in (dict(), dict(use_graph=True, overlap=True)):
        r = parity_checks.run("tiny", torch.float16, B=2, H=128, W=128, steps=3, unet_kw=dict(attn_fp8=True), **kw)
        assert r["garment_feat_max"] <= 8e-2 and r["tryon_eps"] <= 8e-2 and r["latents_final"] <= 8e-2, r
