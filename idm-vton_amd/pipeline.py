"""The try-on denoising engine: host mirror of StableDiffusionXLInpaintPipeline.__call__
(/root/reference/src/tryon_pipeline.py:1254-1894) for the path inference.py:397-414 takes, on the HIP kernels.

Per step (tryon_pipeline.py:1765-1866): pack 13-channel input -> GarmentNet -> TryonNet (garment K/V injected, CFG halves
in one batch, unconditional garment half in closed form) -> fused CFG + scheduler update.  Everything step-invariant is
computed once before the loop (time-embedding tables for every timestep, text / image-token K and V^T of all 140 attn2,
mask / masked-image / pose latents).  Optionally the step is captured into one hipGraph and replayed.
"""
import torch

from . import ops
from .scheduler import StepScheduler


def _copy_state(dst, src):
    """Copy every tensor the captured step reads (latents, conditioning, K/V^T caches) into the graph's buffers."""
    for k in ("latents", "cond", "cloth"):
        dst[k].copy_(src[k])
    for ck in ("ctx_t", "ctx_g"):
        for p, ent in src[ck]["kv"].items():
            for name, t in ent.items():
                dst[ck]["kv"][p][name].copy_(t)


class TryonEngine:
    def __init__(self, unet, unet_encoder, vae, resampler=None, dtype=torch.bfloat16, device="cuda"):
        self.unet, self.unet_encoder, self.vae, self.resampler = unet, unet_encoder, vae, resampler
        self.dtype, self.device = dtype, torch.device(device)
        self._graphs = {}

    # -------------------------------------------------------------------------------------------- preparation
    @torch.no_grad()
    def prepare(self, *, image, mask_image, pose_img, cloth, prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds,
                negative_pooled_prompt_embeds, text_embeds_cloth, noise, num_inference_steps, guidance_scale,
                ip_hidden_states=None, image_embeds=None, scheduler="ddpm", height=None, width=None):
        """Everything before the loop (tryon_pipeline.py:1495-1762).  image in [0,1]; pose_img / cloth in [-1,1];
        noise: dict(latents, masked, pose, cloth [B,4,h,w] fp32; steps [n,B,4,h,w] fp32 or None) -- RNG order SURVEY A.4."""
        dev, dt = self.device, self.dtype
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        image, mask_image, pose_img, cloth = f32(image), f32(mask_image), f32(pose_img), f32(cloth)
        B = image.shape[0]
        H = height or image.shape[-2]
        W = width or image.shape[-1]
        h, w = H // 8, W // 8
        sched = StepScheduler(scheduler)
        timesteps = sched.set_timesteps(num_inference_steps)                               # :1561-1567

        init_image = 2.0 * image - 1.0                                                     # preprocess :1588-1591
        mask = (mask_image >= 0.5).float()                                                 # mask_processor :1593-1595
        masked_image = init_image * (mask < 0.5)                                           # :1602
        latents = f32(noise["latents"]) * sched.init_noise_sigma                           # :889-893
        mask_l = torch.nn.functional.interpolate(mask, size=(h, w))                        # :939-941
        masked_lat = self.vae.encode_sample(masked_image, f32(noise["masked"]))            # :964
        pose_lat = self.vae.encode_sample(pose_img, f32(noise["pose"]))                    # :1644-1647
        cloth_lat = self.vae.encode_sample(cloth, f32(noise["cloth"]))                     # :1654
        # step-invariant 9 conditioning channels of the 13-channel input, NHWC, both CFG halves (:955,977,1649-1652,1777)
        cond = torch.cat([mask_l, masked_lat, pose_lat], dim=1).permute(0, 2, 3, 1).reshape(B, h * w, 9)
        cond = torch.cat([cond, cond], dim=0).to(dt).contiguous()
        cloth_nhwc = ops.to_nhwc(cloth_lat, dt, cpad=self.unet_encoder.cin_pad)

        pe = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(dev)             # :1710
        add_text = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0).to(dev)   # :1711
        time_ids = torch.tensor([[H, W, 0, 0, H, W]], dtype=torch.float32, device=dev).repeat(2 * B, 1)  # :1681-1713
        if image_embeds is None:
            image_embeds = self.resampler(ip_hidden_states.to(dev))                        # :1726 (encoder_hid_proj)
        ctx_t = self.unet.encode_context(pe, image_embeds)
        ctx_g = self.unet_encoder.encode_context(text_embeds_cloth.to(dev))
        temb_t = self.unet.time_embeddings(timesteps, 2 * B, dict(text_embeds=add_text, time_ids=time_ids))
        temb_g = self.unet_encoder.time_embeddings(timesteps, B)
        coef = torch.tensor([list(sched.coeffs(t)) + [guidance_scale] for t in timesteps], dtype=torch.float32, device=dev)
        steps_noise = f32(noise["steps"]) if noise.get("steps") is not None and scheduler == "ddpm" else None
        return dict(B=B, h=h, w=w, timesteps=timesteps, latents=latents.contiguous(), cond=cond, cloth=cloth_nhwc,
                    ctx_t=ctx_t, ctx_g=ctx_g, temb_t=temb_t, temb_g=temb_g, coef=coef, steps_noise=steps_noise,
                    x_in=torch.empty(2 * B, h * w, self.unet.cin_pad, dtype=dt, device=dev),
                    trace=dict(masked_lat=masked_lat, pose_lat=pose_lat, cloth_lat=cloth_lat, image_embeds=image_embeds))

    # -------------------------------------------------------------------------------------------- one step
    def _step(self, st, temb_t, temb_g, coef, noise):
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        _, feats = self.unet_encoder.forward(st["cloth"], temb_g, st["ctx_g"], B, h, w)    # :1787
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_feats=feats)   # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    @torch.no_grad()
    def denoise(self, st, use_graph=False, trace=None):
        n = len(st["timesteps"])
        if not use_graph:
            for i in range(n):
                nz = st["steps_noise"][i] if st["steps_noise"] is not None else None
                eps = self._step(st, st["temb_t"][i], st["temb_g"][i], st["coef"][i], nz)
                if trace is not None:
                    trace.setdefault("step_latents", []).append(st["latents"].clone())
            return st["latents"]
        # ---- hipGraph: one step captured ONCE per shape on persistent buffers, replayed n times per call ----
        key = (st["B"], st["h"], st["w"], n, st["steps_noise"] is not None)
        if key not in self._graphs:
            tt, tg, cf = st["temb_t"][0].clone(), st["temb_g"][0].clone(), st["coef"][0].clone()
            nz = st["steps_noise"][0].clone() if st["steps_noise"] is not None else None
            saved = st["latents"].clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._step(st, tt, tg, cf, nz)                                             # warm-up (allocator, lazy init)
            torch.cuda.current_stream().wait_stream(side)
            st["latents"].copy_(saved)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._step(st, tt, tg, cf, nz)
            self._graphs[key] = (graph, st, tt, tg, cf, nz)
        graph, sst, tt, tg, cf, nz = self._graphs[key]
        if sst is not st:
            _copy_state(sst, st)                                                           # new call -> persistent buffers
        for i in range(n):
            tt.copy_(st["temb_t"][i]); tg.copy_(st["temb_g"][i]); cf.copy_(st["coef"][i])
            if nz is not None:
                nz.copy_(st["steps_noise"][i])
            graph.replay()
        return sst["latents"]

    @torch.no_grad()
    def decode(self, latents):
        img = self.vae.decode(latents / self.vae.cfg.scaling_factor)                       # :1876
        return (img / 2 + 0.5).clamp(0, 1)                                                 # postprocess

    @torch.no_grad()
    def __call__(self, *, return_latents=False, use_graph=False, **kw):
        st = self.prepare(**kw)
        lat = self.denoise(st, use_graph=use_graph)
        return lat if return_latents else self.decode(lat)
