"""Builds the SAME model twice -- oracle (fp32 torch, CPU) and product (HIP kernels, GPU) -- from one seeded state dict
whose values are first rounded to the product's storage dtype, so both sides hold bit-identical weights and the
comparison isolates kernel arithmetic ("same storage dtype policy")."""
import dataclasses

import torch


TINY = dict(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), num_attention_heads=(1, 2, 4),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32,
            encoder_hid_dim=128, resampler=dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, ff_mult=4))
TINY_VAE = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
# a wider/deeper mid-size config: exercises 128x128 GEMM tiles, 10/20-head style attention, odd group sizes (320/32=10)
MID = dict(block_out_channels=(320, 640, 1280), transformer_layers_per_block=(1, 1, 2), num_attention_heads=(5, 10, 20),
           cross_attention_dim=512, addition_time_embed_dim=64, projection_class_embeddings_input_dim=256 + 6 * 64,
           encoder_hid_dim=256, resampler=dict(dim=256, depth=1, dim_head=64, heads=4, num_queries=16, ff_mult=4))


def build(kind, dtype, device, seed=0, unet_kw=None):
    """-> dict(oracle=(unet, garm, vae), product=(HipUNet, HipUNet, HipVAE, HipResampler), cfgs)"""
    from idm_vton_amd import config as pc
    from idm_vton_amd.resampler import HipResampler
    from idm_vton_amd.unet import HipUNet
    from idm_vton_amd.vae import HipVAE
    from oracle import unet as ou, vae as ov

    kw = dict(TINY if kind == "tiny" else MID)
    tcfg = pc.UNetConfig(mode="tryon", in_channels=13, **kw)
    gcfg = pc.UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None, **kw)
    vcfg = pc.VAEConfig(**TINY_VAE)
    rnd = lambda sd: {k: v.to(dtype).float() for k, v in sd.items()}
    sd_t = rnd(pc.random_state_dict(pc.unet_param_shapes(tcfg), seed + 1, torch.float32, "cpu"))
    sd_g = rnd(pc.random_state_dict(pc.unet_param_shapes(gcfg), seed + 2, torch.float32, "cpu"))
    sd_v = rnd(pc.random_state_dict(pc.vae_param_shapes(vcfg), seed + 3, torch.float32, "cpu", std=0.05))

    as_o = lambda c, cls: cls(**{f.name: getattr(c, f.name) for f in dataclasses.fields(cls)})
    o_t = ou.UNet2DConditionModel(as_o(tcfg, ou.UNetConfig)).eval()
    o_g = ou.UNet2DConditionModel(as_o(gcfg, ou.UNetConfig)).eval()
    o_v = ov.AutoencoderKL(as_o(vcfg, ov.VAEConfig)).eval()
    o_t.load_state_dict(sd_t)
    o_g.load_state_dict(sd_g)
    o_v.load_state_dict(sd_v)
    prod = None
    if device != "cpu":
        p_t = HipUNet(tcfg, sd_t, dtype, device, **(unet_kw or {}))
        p_g = HipUNet(gcfg, sd_g, dtype, device, **(unet_kw or {}))
        p_v = HipVAE(vcfg, sd_v, dtype, device)
        p_r = HipResampler(sd_t, prefix="encoder_hid_proj.", dtype=dtype, device=device, **tcfg.resampler)
        prod = (p_t, p_g, p_v, p_r)
    return dict(oracle=(o_t, o_g, o_v), product=prod, cfgs=(tcfg, gcfg, vcfg), xd=kw["cross_attention_dim"],
                pooled=kw["projection_class_embeddings_input_dim"] - 6 * kw["addition_time_embed_dim"],
                enc_dim=kw["encoder_hid_dim"])


def make_inputs(B, H, W, xd, pooled, enc_dim, steps, dtype, seed=42):
    """Seeded synthetic inputs; embeddings are pre-rounded to the storage dtype."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    q = lambda t: t.to(dtype).float()
    h, w = H // 8, W // 8
    mask = torch.zeros(B, 1, H, W)
    mask[:, :, H // 4: 3 * H // 4, W // 4: 3 * W // 4] = 1
    return dict(image=torch.rand(B, 3, H, W, generator=g), mask_image=mask, pose_img=r(B, 3, H, W).clamp(-1, 1),
                cloth=r(B, 3, H, W).clamp(-1, 1), prompt_embeds=q(r(B, 77, xd)), negative_prompt_embeds=q(r(B, 77, xd)),
                pooled_prompt_embeds=q(r(B, pooled)), negative_pooled_prompt_embeds=q(r(B, pooled)),
                text_embeds_cloth=q(r(B, 77, xd)), ip_hidden_states=q(r(2 * B, 257, enc_dim)),
                noise=dict(latents=r(B, 4, h, w), masked=r(B, 4, h, w), pose=r(B, 4, h, w), cloth=r(B, 4, h, w),
                           steps=r(steps, B, 4, h, w)))


def relerr(x, ref):
    x, ref = x.detach().float().cpu(), ref.detach().float().cpu()
    if not torch.isfinite(x).all():
        return float("inf")
    return ((x - ref).abs().max() / ref.abs().max().clamp_min(1e-20)).item()
