"""-m gpu: the drop-in boundary on the GPU -- the reference's plugin API (attention processors, Resampler) against plain
PyTorch fp32 references of the reference's arithmetic (ip_adapter/attention_processor.py:203-278,1907-2010), and the
`StableDiffusionXLInpaintPipeline` call surface end to end against the engine it wraps (bit-exact) and the oracle."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT, DEV = torch.float16, "cuda"


def _rel(x, ref):
    x, ref = x.float().cpu(), ref.float().cpu()
    assert torch.isfinite(x).all()
    return ((x - ref).abs().max() / ref.abs().max()).item()


def _heads(t, h):
    return t.float().view(t.shape[0], t.shape[1], h, 64).transpose(1, 2)


def _mk_attn(query_dim, cross_dim, heads, processor, seed):
    from idm_vton_amd.boundary.modules import Attention
    torch.manual_seed(seed)
    a = Attention(query_dim, cross_dim, heads, 64, processor=processor).to(DEV, DT)
    return a


@pytest.mark.parametrize("four_d", [False, True], ids=["tokens", "nchw"])
def test_attn_processor_self_attention(four_d):
    from ip_adapter.attention_processor import AttnProcessor2_0
    attn = _mk_attn(128, None, 2, AttnProcessor2_0(), 0)
    x = (torch.randn(2, 128, 10, 10) if four_d else torch.randn(2, 100, 128)).to(DEV, DT)
    out = attn(x)
    t = x.view(2, 128, 100).transpose(1, 2) if four_d else x
    q, k, v = (F.linear(t.float(), w.weight.float()) for w in (attn.to_q, attn.to_k, attn.to_v))
    o = F.scaled_dot_product_attention(_heads(q, 2), _heads(k, 2), _heads(v, 2)).transpose(1, 2).reshape(2, 100, 128)
    ref = F.linear(o, attn.to_out[0].weight.float(), attn.to_out[0].bias.float())
    if four_d:
        ref = ref.transpose(1, 2).reshape(2, 128, 10, 10)
    assert out.shape == x.shape and _rel(out, ref) < 4e-3


def test_attn_processor_cross_attention_ragged_keys():
    from ip_adapter.attention_processor import AttnProcessor2_0
    attn = _mk_attn(128, 192, 2, AttnProcessor2_0(), 1)
    x, enc = torch.randn(2, 72, 128).to(DEV, DT), torch.randn(2, 77, 192).to(DEV, DT)
    out = attn(x, encoder_hidden_states=enc)
    q = F.linear(x.float(), attn.to_q.weight.float())
    k, v = F.linear(enc.float(), attn.to_k.weight.float()), F.linear(enc.float(), attn.to_v.weight.float())
    o = F.scaled_dot_product_attention(_heads(q, 2), _heads(k, 2), _heads(v, 2)).transpose(1, 2).reshape(2, 72, 128)
    ref = F.linear(o, attn.to_out[0].weight.float(), attn.to_out[0].bias.float())
    assert _rel(out, ref) < 4e-3


def test_ip_attn_processor_text_plus_image_tokens():
    from ip_adapter.attention_processor import IPAttnProcessor2_0
    proc = IPAttnProcessor2_0(hidden_size=128, cross_attention_dim=192, scale=0.75, num_tokens=16)
    attn = _mk_attn(128, 192, 2, proc, 2)
    assert "processor.to_k_ip.weight" in attn.state_dict()                       # reference key layout (Appendix C)
    x, enc = torch.randn(2, 64, 128).to(DEV, DT), torch.randn(2, 77 + 16, 192).to(DEV, DT)
    out = attn(x, encoder_hidden_states=enc)
    text, ip = enc[:, :77].float(), enc[:, 77:].float()
    q = _heads(F.linear(x.float(), attn.to_q.weight.float()), 2)
    sd = lambda kk, vv: F.scaled_dot_product_attention(q, _heads(kk, 2), _heads(vv, 2)).transpose(1, 2).reshape(2, 64, 128)
    o = sd(F.linear(text, attn.to_k.weight.float()), F.linear(text, attn.to_v.weight.float()))
    o = o + 0.75 * sd(F.linear(ip, proc.to_k_ip.weight.float()), F.linear(ip, proc.to_v_ip.weight.float()))
    ref = F.linear(o, attn.to_out[0].weight.float(), attn.to_out[0].bias.float())
    assert _rel(out, ref) < 4e-3
    with pytest.raises(ValueError):
        attn(x)                                                                    # no encoder_hidden_states
    with pytest.raises(NotImplementedError):
        attn(x, encoder_hidden_states=enc, attention_mask=torch.ones(2, 1, 93, device=DEV))


def test_resampler_boundary_matches_oracle():
    from ip_adapter.resampler import Resampler
    from oracle.resampler import Resampler as ORes
    kw = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=128, output_dim=192, ff_mult=4)
    torch.manual_seed(3)
    o = ORes(**kw).eval()
    with torch.no_grad():
        for p in o.parameters():
            p.copy_(p.to(DT).float())
    b = Resampler(**kw)
    b.load_state_dict(o.state_dict(), strict=True)
    b = b.to(DEV, DT)
    x = torch.randn(2, 257, 128).to(DT)
    with torch.no_grad():
        ref = o(x.float())
        out = b(x.to(DEV))
    assert out.shape == (2, 16, 192) and _rel(out, ref) < 6e-3
    with pytest.raises(RuntimeError, match="GPU only"):
        b(x)                                                                       # CPU tensor: no fallback


class _FakeCLIPVision(torch.nn.Module):
    """Stand-in for CLIPVisionModelWithProjection: deterministic 257-token hidden states from the pixels."""

    def __init__(self, dim):
        super().__init__()
        self.proj = torch.nn.Linear(3, dim)

    def forward(self, pixel_values, output_hidden_states=False):
        p = F.adaptive_avg_pool2d(pixel_values.float(), (16, 16)).flatten(2).transpose(1, 2)            # [B,256,3]
        t = torch.cat([p.mean(1, keepdim=True), p], dim=1)
        h = self.proj(t.to(self.proj.weight.dtype))
        return SimpleNamespace(hidden_states=[h * 0.5, h, h * 2.0], image_embeds=h[:, 0])


def test_pipeline_boundary_end_to_end():
    from idm_vton_amd import config as pc
    from idm_vton_amd.boundary.scheduler import DDPMScheduler
    from idm_vton_amd.boundary.vae import AutoencoderKL
    from src.tryon_pipeline import StableDiffusionXLInpaintPipeline
    from src.unet_hacked_garmnet import UNet2DConditionModel as G
    from src.unet_hacked_tryon import UNet2DConditionModel as T
    from tests import parity_utils as pu
    kw = dict(pu.TINY)
    tcfg = pc.UNetConfig(mode="tryon", in_channels=13, sample_size=16, **kw)
    gcfg = pc.UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None, sample_size=16, **kw)
    vcfg = pc.VAEConfig(**pu.TINY_VAE)
    rnd = lambda sd: {k: v.to(DT) for k, v in sd.items()}
    t = T(tcfg, torch_dtype=DT); t.load_state_dict(rnd(pc.random_state_dict(pc.unet_param_shapes(tcfg), 1, torch.float32, "cpu")))
    g = G(gcfg, torch_dtype=DT); g.load_state_dict(rnd(pc.random_state_dict(pc.unet_param_shapes(gcfg), 2, torch.float32, "cpu")))
    v = AutoencoderKL(vcfg, torch_dtype=DT); v.load_state_dict(rnd(pc.random_state_dict(pc.vae_param_shapes(vcfg), 3, torch.float32, "cpu", std=0.05)))
    torch.manual_seed(5)
    enc = _FakeCLIPVision(kw["encoder_hid_dim"]).to(DT)
    pipe = StableDiffusionXLInpaintPipeline(vae=v, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=t,
                                            unet_encoder=g, scheduler=DDPMScheduler(), image_encoder=enc).to(DEV)
    assert str(pipe.device) == "cuda"
    B, H, W, steps = 2, 128, 128, 3
    inp = pu.make_inputs(B, H, W, kw["cross_attention_dim"], 64, kw["encoder_hid_dim"], steps, DT)
    clip_pix = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    call = dict(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                pooled_prompt_embeds=inp["pooled_prompt_embeds"], negative_pooled_prompt_embeds=inp["negative_pooled_prompt_embeds"],
                num_inference_steps=steps, strength=1.0, pose_img=inp["pose_img"], text_embeds_cloth=inp["text_embeds_cloth"],
                cloth=inp["cloth"], mask_image=inp["mask_image"], image=inp["image"], height=H, width=W, guidance_scale=2.0,
                ip_adapter_image=clip_pix)
    torch.manual_seed(123)                                                         # the pose posterior uses the GLOBAL generator
    images = pipe(generator=torch.Generator(DEV).manual_seed(7), **call)
    assert isinstance(images, tuple) and len(images) == 1 and len(images[0]) == B and images[0][0].size == (W, H)
    torch.manual_seed(123)
    img_pt = pipe(generator=torch.Generator(DEV).manual_seed(7), output_type="pt", **call)[0]
    # the same through the engine the pipeline wraps, with the noise drawn in the reference's order
    eng = pipe.hip_engine()
    gen = torch.Generator(DEV).manual_seed(7)
    torch.manual_seed(123)
    # dtypes as the reference draws them (randn_tensor): latents in the prompt dtype, VAE posterior samples in fp32 (upcast VAE),
    # DDPM variance noise in the model-output dtype
    draw = lambda gg, dt_: torch.randn((B, 4, H // 8, W // 8), generator=gg, device=DEV, dtype=dt_).float()
    n_lat, n_masked, n_pose, n_cloth = draw(gen, DT), draw(gen, torch.float32), draw(None, torch.float32), draw(gen, torch.float32)
    n_steps = torch.stack([draw(gen, DT) for _ in range(steps)])
    with torch.no_grad():
        pos = enc(clip_pix.to(DEV, DT), output_hidden_states=True).hidden_states[-2]
        neg = enc(torch.zeros_like(clip_pix).to(DEV, DT), output_hidden_states=True).hidden_states[-2]
    ref = eng(image=inp["image"], mask_image=inp["mask_image"], pose_img=inp["pose_img"], cloth=inp["cloth"],
              prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
              pooled_prompt_embeds=inp["pooled_prompt_embeds"], negative_pooled_prompt_embeds=inp["negative_pooled_prompt_embeds"],
              text_embeds_cloth=inp["text_embeds_cloth"], noise=dict(latents=n_lat, masked=n_masked, pose=n_pose, cloth=n_cloth, steps=n_steps),
              num_inference_steps=steps, guidance_scale=2.0, ip_hidden_states=torch.cat([neg, pos]), scheduler="ddpm")
    assert torch.equal(img_pt, ref)
    # per-step callbacks and `interrupt` (/root/reference/src/tryon_pipeline.py:1766-1767,1840-1863) run the engine's serial un-captured loop:
    # same latents as the captured loop, the callback sees every step's latents, a replaced `latents` is used, an interrupt skips the rest
    seen = []

    def cb(p_, i, t_, kw_):
        seen.append((i, int(t_), kw_["latents"].clone()))
        return {}
    torch.manual_seed(123)
    lat_plain = pipe(generator=torch.Generator(DEV).manual_seed(7), output_type="latent", **call)[0]
    torch.manual_seed(123)
    lat_cb = pipe(generator=torch.Generator(DEV).manual_seed(7), output_type="latent", callback_on_step_end=cb, **call)[0]
    assert torch.equal(lat_plain, lat_cb) and [i for i, _, _ in seen] == list(range(steps))
    assert torch.equal(seen[-1][2].float(), lat_cb.to(seen[-1][2].dtype).float())

    def stop_after_first(p_, i, t_, kw_):
        p_._interrupt = True
        return {"latents": torch.zeros_like(kw_["latents"])}
    torch.manual_seed(123)
    lat_int = pipe(generator=torch.Generator(DEV).manual_seed(7), output_type="latent", callback_on_step_end=stop_after_first, **call)[0]
    assert float(lat_int.abs().max()) == 0.0                      # step 0's latents replaced by zeros, every later step skipped
    with pytest.raises(NotImplementedError, match="callback_on_step_end_tensor_inputs"):
        pipe(output_type="latent", callback_on_step_end=cb, callback_on_step_end_tensor_inputs=["prompt_embeds"], **call)
    # the other two branches of __call__: strength < 1 (draw order: init-image posterior FIRST, then the latent noise; 5 steps * 0.6
    # -> the last 3) and guidance_scale <= 1 (no CFG: negative embeddings unused, conditional IP rows only)
    torch.manual_seed(123)
    img2 = pipe(generator=torch.Generator(DEV).manual_seed(7), output_type="pt", **{**call, "strength": 0.6, "guidance_scale": 1.0, "num_inference_steps": 5})[0]
    gen = torch.Generator(DEV).manual_seed(7)
    torch.manual_seed(123)
    n_img, n_lat = draw(gen, torch.float32), draw(gen, DT)
    n_masked, n_pose, n_cloth = draw(gen, torch.float32), draw(None, torch.float32), draw(gen, torch.float32)
    n_steps = torch.stack([draw(gen, DT) for _ in range(3)])
    ref2 = eng(image=inp["image"], mask_image=inp["mask_image"], pose_img=inp["pose_img"], cloth=inp["cloth"],
               prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=None, pooled_prompt_embeds=inp["pooled_prompt_embeds"],
               negative_pooled_prompt_embeds=None, text_embeds_cloth=inp["text_embeds_cloth"],
               noise=dict(image=n_img, latents=n_lat, masked=n_masked, pose=n_pose, cloth=n_cloth, steps=n_steps),
               num_inference_steps=5, strength=0.6, guidance_scale=1.0, ip_hidden_states=pos, scheduler="ddpm", image_dtype=DT)
    assert torch.equal(img2, ref2) and not torch.equal(img2, img_pt)
    pil0 = torch.from_numpy(__import__("numpy").asarray(images[0][0])).float() / 255.0
    assert (pil0 - img_pt[0].permute(1, 2, 0).cpu()).abs().max() <= 0.5 / 255 + 1e-6
    # unet forward surface: tuple / .sample returns, garment feature list (src/unet_hacked_garmnet.py:1281-1284)
    lat = torch.randn(B, 4, 16, 16, device=DEV, dtype=DT)
    (sample,), feats = g(lat, 481, inp["text_embeds_cloth"].to(DEV), return_dict=False)
    assert sample is None and len(feats) == 17 and feats[0].shape == (B, 64, 128)
    added = dict(text_embeds=inp["pooled_prompt_embeds"].to(DEV), time_ids=torch.tensor([[H, W, 0, 0, H, W]] * B, device=DEV, dtype=DT),
                 image_embeds=torch.randn(B, 16, kw["cross_attention_dim"], device=DEV, dtype=DT))
    out = t(torch.randn(B, 13, 16, 16, device=DEV, dtype=DT), 481, inp["prompt_embeds"].to(DEV), added_cond_kwargs=added,
            garment_features=feats, return_dict=False)[0]
    assert out.shape == (B, 4, 16, 16) and torch.isfinite(out).all()
    with pytest.raises(ValueError, match="text_embeds"):
        t(torch.randn(B, 13, 16, 16, device=DEV, dtype=DT), 481, inp["prompt_embeds"].to(DEV), added_cond_kwargs={}, garment_features=feats)
    # IPAttnProcessor2_0.scale is honoured by the fused forward (hidden = text + scale * ip, attention_processor.py:1995):
    # with scale 0 the image tokens have no effect at all, with scale 1 they do
    x13 = torch.randn(B, 13, 16, 16, device=DEV, dtype=DT)
    other = dict(added, image_embeds=torch.randn(B, 16, kw["cross_attention_dim"], device=DEV, dtype=DT))
    run = lambda ad: t(x13, 481, inp["prompt_embeds"].to(DEV), added_cond_kwargs=ad, garment_features=feats, return_dict=False)[0]
    base1, alt1 = run(added), run(other)
    assert not torch.equal(base1, alt1)
    for proc in t.attn_processors.values():
        if hasattr(proc, "to_k_ip"):
            proc.scale = 0.0
    base0, alt0 = run(added), run(other)
    assert torch.equal(base0, alt0) and not torch.equal(base0, base1)


def test_vae_fp16_module_survives_activations_beyond_fp16_range():
    """the stock SDXL VAE overflows fp16 activations, which is why the reference upcasts it to fp32
    (tryon_pipeline.py:911-930, 1868-1880).  Weights here are scaled so the first conv's outputs reach ~1e5 (> 65504): an fp16
    module with force_upcast (the SDXL default) must still agree with the fp32 oracle -- the HIP VAE stores bf16 for it -- while
    fp16 STORAGE (force_upcast=False) demonstrably cannot."""
    import dataclasses
    from idm_vton_amd import config as pc
    from idm_vton_amd.boundary.vae import AutoencoderKL
    from oracle import vae as ov
    from tests import parity_utils as pu
    vcfg = pc.VAEConfig(**pu.TINY_VAE)
    sd = pc.random_state_dict(pc.vae_param_shapes(vcfg), 7, torch.float32, "cpu", std=0.05)
    for k in ("encoder.conv_in.weight", "encoder.conv_in.bias", "decoder.conv_in.weight", "decoder.conv_in.bias"):
        sd[k] = (sd[k] * 4e5).clamp(-6e4, 6e4)                  # still fp16-representable weights, activations far beyond fp16
    sd = {k: v.half().float() for k, v in sd.items()}
    o = ov.AutoencoderKL(ov.VAEConfig(**{f.name: getattr(vcfg, f.name) for f in dataclasses.fields(ov.VAEConfig)})).eval()
    o.load_state_dict(sd)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        act = torch.nn.functional.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
        assert act.abs().max() > 65504, "the fixture no longer exceeds the fp16 range"
        mean_o, _ = o.encode_moments(x)
        dec_o = o.decode(mean_o)
    v = AutoencoderKL(vcfg, torch_dtype=torch.float16)
    v.load_state_dict({k: t.half() for k, t in sd.items()})
    v = v.to(DEV)
    assert v.config.force_upcast and v.hip_engine().dtype == torch.bfloat16
    mean_p = v.encode(x.to(DEV)).latent_dist.mode()
    dec_p = v.decode(mean_o.to(DEV)).sample
    assert torch.isfinite(mean_p).all() and torch.isfinite(dec_p).all()
    assert _rel(mean_p, mean_o) < 4e-2 and _rel(dec_p, dec_o) < 4e-2
    v.config.force_upcast = False                               # fp16 storage: what the upcast exists to avoid
    assert v.hip_engine().dtype == torch.float16
    bad = v.encode(x.to(DEV)).latent_dist.mode()
    assert (not torch.isfinite(bad).all()) or _rel(bad, mean_o) > 0.2
