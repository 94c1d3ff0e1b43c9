"""Oracle try-on pipeline (pure torch, fp32, CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates `StableDiffusionXLInpaintPipeline.__call__` (src/tryon_pipeline.py:1254-1894) for the path
`inference.py:397-414` takes (13-channel UNet, precomputed prompt embeds; strength and guidance_scale as arguments), in the reference's
own execution order (including the 2N-query self-attention and the zeros-cat of the garment features), with every
random draw supplied by the caller in the order of SURVEY.md A.4 so the run is device-independent.

CLIP encoders (encode_prompt / encode_image) are outside the loop and not restated: the caller passes
prompt embeddings and the CLIP-H penultimate hidden states `[uncond ; cond]` directly.
"""
import torch
import torch.nn.functional as F


def preprocess_image(image):
    """VaeImageProcessor.preprocess for a [0,1] tensor batch: 2x-1."""
    return 2.0 * image - 1.0


def preprocess_mask(mask):
    """mask_processor (do_binarize, no normalise): >=0.5 -> 1 (tryon_pipeline.py:419-422)."""
    return (mask >= 0.5).to(mask.dtype)


@torch.no_grad()
def denoise(unet, unet_encoder, sched, timesteps, latents, mask_l, masked_lat, pose_lat, cloth_lat, pe, added,
            text_embeds_cloth, guidance_scale, noise_steps=None, trace=None, cfg=True):
    """The loop body of `__call__` (src/tryon_pipeline.py:1764-1866) from prepared conditioning: mask_l / masked_lat / pose_lat
    [2B,*,h,w] (both CFG halves; [B,...] when cfg is False = guidance_scale <= 1, :1769,1795,1814), cloth_lat [B,4,h,w],
    pe [2B,77,D], added = {text_embeds, time_ids, image_embeds}."""
    for i, t in enumerate(timesteps):                                                   # :1765
        lmi = torch.cat([latents] * 2) if cfg else latents                              # :1769 (scale_model_input = id)
        lmi = torch.cat([lmi, mask_l, masked_lat, pose_lat], dim=1)                     # :1777
        _, feats = unet_encoder(cloth_lat, t, text_embeds_cloth)                        # :1787
        if cfg:
            feats = [torch.cat([torch.zeros_like(d), d]) for d in feats]                # :1795-1796
        eps = unet(lmi, t, pe, added_cond_kwargs=added, garment_features=feats)[0]      # :1799-1808
        if cfg:
            eu, et = eps.chunk(2)                                                       # :1814-1815
            eps = eu + guidance_scale * (et - eu)                                       # :1816
        latents = sched.step(eps, t, latents, noise_steps[i] if noise_steps is not None else None)  # :1823
        if trace is not None:
            trace["step_eps"].append(eps.clone())
            trace["step_latents"].append(latents.clone())
    return latents


@torch.no_grad()
def run(unet, unet_encoder, vae, sched, *, image, mask_image, pose_img, cloth, prompt_embeds, negative_prompt_embeds,
        pooled_prompt_embeds, negative_pooled_prompt_embeds, text_embeds_cloth, ip_hidden_states, noise,
        num_inference_steps=30, guidance_scale=2.0, height=None, width=None, return_latents=False, trace=None, strength=1.0):
    """noise: dict(latents[B,4,h,w], masked[B,4,h,w], pose[B,4,h,w], cloth[B,4,h,w], steps[n,B,4,h,w]; image[B,4,h,w] when
    strength < 1: the posterior draw of the init-image encode, the FIRST draw of the reference in that case).
    guidance_scale <= 1: no classifier-free guidance -- negative_* are ignored and ip_hidden_states holds the cond rows only.

    `trace`: optional dict that receives intermediate tensors for per-stage parity checks.
    """
    B = image.shape[0]
    height = height or image.shape[-2]
    width = width or image.shape[-1]
    sf = vae.cfg.scaling_factor
    cfg = guidance_scale > 1                                                            # do_classifier_free_guidance (:440-442)
    rep = 2 if cfg else 1
    timesteps = sched.set_timesteps(num_inference_steps)                                # :1561
    init_t = min(int(num_inference_steps * strength), num_inference_steps)              # get_timesteps :987-995
    timesteps = timesteps[max(num_inference_steps - init_t, 0):]
    if len(timesteps) < 1:                                                              # :1568-1572
        raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline"
                         f"steps is {len(timesteps)} which is < 1 and not appropriate for this pipeline.")

    init_image = preprocess_image(image).float()                                        # :1588-1591
    mask = preprocess_mask(mask_image)                                                  # :1593-1595
    masked_image = init_image * (mask < 0.5)                                            # :1602

    if strength == 1.0 or noise.get("image") is None:
        latents = noise["latents"] * sched.init_noise_sigma                             # :889-893  (RNG #1)
    else:                                                                               # image + noise start (:883-891)
        image_latents = sf * vae.encode_sample(init_image, noise["image"])              # _encode_vae_image :911-932 (RNG #0)
        latents = sched.add_noise(image_latents, noise["latents"], timesteps[0])        # (RNG #1)
    mask_l = F.interpolate(mask, size=(height // 8, width // 8))                        # :939-941 (nearest)
    mask_l = torch.cat([mask_l] * rep)                                                  # :955
    masked_lat = sf * vae.encode_sample(masked_image, noise["masked"])                  # :964 -> :911-932 (RNG #2)
    masked_lat = torch.cat([masked_lat] * rep)                                          # :977-979
    pose_lat = sf * vae.encode_sample(pose_img, noise["pose"])                          # :1644-1647 (RNG #3)
    pose_lat = torch.cat([pose_lat] * rep)                                              # :1649-1652
    cloth_lat = sf * vae.encode_sample(cloth, noise["cloth"])                           # :1654 (RNG #4)

    add_time_ids = torch.tensor([[height, width, 0, 0, height, width]], dtype=prompt_embeds.dtype)  # :1681-1705
    add_time_ids = add_time_ids.repeat(rep * B, 1)                                      # :1707-1713 (neg == pos here)
    pe = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0) if cfg else prompt_embeds                       # :1710
    add_text = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0) if cfg else pooled_prompt_embeds   # :1711
    image_embeds = unet.encoder_hid_proj(ip_hidden_states)                              # :1726 ([uncond ; cond] rows, or cond only)
    added = {"text_embeds": add_text, "time_ids": add_time_ids, "image_embeds": image_embeds}
    if trace is not None:
        trace.update(masked_lat=masked_lat, pose_lat=pose_lat, cloth_lat=cloth_lat, image_embeds=image_embeds,
                     mask_l=mask_l, latents0=latents.clone(), step_latents=[], step_eps=[])

    latents = denoise(unet, unet_encoder, sched, timesteps, latents, mask_l, masked_lat, pose_lat, cloth_lat, pe, added,
                      text_embeds_cloth, guidance_scale, noise.get("steps"), trace, cfg=cfg)
    if return_latents:
        return latents
    img = vae.decode(latents / sf)                                                      # :1876
    return (img / 2 + 0.5).clamp(0, 1)                                                  # postprocess
