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
        self._side = None

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
        """Serial form of one loop iteration (tryon_pipeline.py:1765-1866) on the current stream: parity tests, traces and
        the per-kernel roofline leg of bench.py use this; the throughput path is the two-stream form below."""
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        _, feats = self.unet_encoder.forward(st["cloth"], temb_g, st["ctx_g"], B, h, w)    # :1787
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_feats=feats)   # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    # GarmentNet's inputs (cloth latent, cloth text, timestep) do not depend on the latents, so GarmentNet for step i+1 --
    # and the attn1 K / V^T projections of its 70 features with TryonNet's weights -- run on a second HIP stream while
    # TryonNet runs step i.  Two feature sets alternate; there is no other coupling between the streams.
    def _garment_side(self, st, temb_g, fset):
        B, h, w = st["B"], st["h"], st["w"]
        self.unet_encoder.forward(st["cloth"], temb_g, st["ctx_g"], B, h, w, feats_buf=fset["feats"])      # :1787
        self.unet.project_garment_kv(fset["feats"], out=fset["kv"])

    def _tryon_main(self, st, temb_t, coef, noise, fset):
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_kv=fset["kv"])  # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    def _feature_sets(self, st, temb_g0):
        """Two persistent {70 features, 70 (K, V^T)} sets; set 0 is filled for the first step on the current stream."""
        B, h, w = st["B"], st["h"], st["w"]
        _, feats = self.unet_encoder.forward(st["cloth"], temb_g0, st["ctx_g"], B, h, w)
        kv = self.unet.project_garment_kv(feats)
        s0 = dict(feats=feats, kv=kv)
        s1 = dict(feats=[torch.empty_like(f) for f in feats], kv=[(torch.empty_like(k), torch.empty_like(v)) for k, v in kv])
        return [s0, s1]

    def _denoise_overlap_eager(self, st):
        n = len(st["timesteps"])
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        sets = self._feature_sets(st, st["temb_g"][0])
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        side.wait_stream(main)                                       # prepare()'s tensors and set 0 are complete
        for i in range(n):
            cur, nxt = i & 1, (i + 1) & 1
            if i + 1 < n:
                with torch.cuda.stream(side):
                    if i >= 1:
                        side.wait_event(free[nxt])                   # TryonNet step i-1 is done reading set nxt
                    self._garment_side(st, st["temb_g"][i + 1], sets[nxt])
                    ready[nxt].record(side)
            if i >= 1:
                main.wait_event(ready[cur])
            nz = st["steps_noise"][i] if st["steps_noise"] is not None else None
            self._tryon_main(st, st["temb_t"][i], st["coef"][i], nz, sets[cur])
            free[cur].record(main)
        main.wait_stream(side)
        return st["latents"]

    def _denoise_overlap_graph(self, st):
        """hipGraph form of the two-stream loop: per parity one graph with two parallel branches {TryonNet step i on set p |
        GarmentNet step i+1 into set p^1}, plus a TryonNet-only graph for the last step.  Consecutive graph launches are
        ordered on the launching stream, which is exactly the dependency the two sets need.  Captures use
        capture_error_mode="thread_local": with torch.distributed / RCCL initialised a watchdog thread polls events, which
        the default global mode would treat as a capture violation."""
        n = len(st["timesteps"])
        has_noise = st["steps_noise"] is not None
        key = (st["B"], st["h"], st["w"], has_noise, "overlap")
        if key not in self._graphs:
            tt, tg, cf = st["temb_t"][0].clone(), st["temb_g"][0].clone(), st["coef"][0].clone()
            nz = st["steps_noise"][0].clone() if has_noise else None
            saved = st["latents"].clone()
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):                            # warm-up off the default stream (allocator, lazy init)
                sets = self._feature_sets(st, tg)
                self._garment_side(st, tg, sets[1])
                self._tryon_main(st, tt, cf, nz, sets[0])
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            st["latents"].copy_(saved)
            side = torch.cuda.Stream()
            pair, last = {}, {}
            for par in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    cap = torch.cuda.current_stream()
                    side.wait_stream(cap)                            # fork
                    with torch.cuda.stream(side):
                        self._garment_side(st, tg, sets[par ^ 1])
                    self._tryon_main(st, tt, cf, nz, sets[par])
                    cap.wait_stream(side)                            # join
                pair[par] = g
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                    self._tryon_main(st, tt, cf, nz, sets[par])
                last[par] = g2
            st["latents"].copy_(saved)
            self._graphs[key] = dict(st=st, tt=tt, tg=tg, cf=cf, nz=nz, sets=sets, pair=pair, last=last)
        G = self._graphs[key]
        sst, tt, tg, cf, nz = G["st"], G["tt"], G["tg"], G["cf"], G["nz"]
        if sst is not st:
            _copy_state(sst, st)                                                           # new call -> persistent buffers
        self._garment_side(sst, st["temb_g"][0].contiguous(), G["sets"][0])               # step 0's features (eager, once)
        for i in range(n):
            tt.copy_(st["temb_t"][i]); cf.copy_(st["coef"][i])
            if nz is not None:
                nz.copy_(st["steps_noise"][i])
            if i + 1 < n:
                tg.copy_(st["temb_g"][i + 1])
                G["pair"][i & 1].replay()
            else:
                G["last"][i & 1].replay()
        return sst["latents"]

    @torch.no_grad()
    def denoise(self, st, use_graph=False, trace=None, overlap=False):
        n = len(st["timesteps"])
        if overlap:
            return self._denoise_overlap_graph(st) if use_graph else self._denoise_overlap_eager(st)
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
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
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
            if trace is not None:
                trace.setdefault("step_latents", []).append(sst["latents"].clone())
        return sst["latents"]

    @torch.no_grad()
    def decode(self, latents):
        img = self.vae.decode(latents / self.vae.cfg.scaling_factor)                       # :1876
        return (img / 2 + 0.5).clamp(0, 1)                                                 # postprocess

    @torch.no_grad()
    def __call__(self, *, return_latents=False, use_graph=False, overlap=False, **kw):
        st = self.prepare(**kw)
        lat = self.denoise(st, use_graph=use_graph, overlap=overlap)
        return lat if return_latents else self.decode(lat)
