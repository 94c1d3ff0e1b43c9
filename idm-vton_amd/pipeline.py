"""The try-on denoising engine: host mirror of StableDiffusionXLInpaintPipeline.__call__
(/root/reference/src/tryon_pipeline.py:1254-1894) for the path inference.py:397-414 takes, on the HIP kernels.

Per step (tryon_pipeline.py:1765-1866): pack 13-channel input -> GarmentNet -> TryonNet (garment K/V injected, CFG halves
in one batch, unconditional garment half in closed form) -> fused CFG + scheduler update.  Everything step-invariant is
computed once before the loop (time-embedding tables for every timestep, text / image-token K and V^T of all 140 attn2,
mask / masked-image / pose latents).  Optionally the step is captured into one hipGraph and replayed.

GarmentNet's inputs (cloth latent, cloth text, timestep) never depend on the latents, so the loop runs it for `garment_steps`
CONSECUTIVE TIMESTEPS IN ONE BATCH (batch = images x timesteps, each batch element with its own time embedding): the same
per-(image, timestep) arithmetic as the reference's one call per step (:1781-1787), but its GEMMs see 2-3x the rows (M = 1536
-> 4608 at the 1280-channel level), which is where the small-M projections of the loop lose their efficiency.  A "block" is
that GarmentNet batch plus the `garment_steps` TryonNet steps that consume its features.
"""
import torch

from . import ops
from .scheduler import StepScheduler


def _copy_state(dst, src):
    """Copy every tensor the captured step reads (latents, conditioning, K/V^T caches) into the graph's buffers."""
    for k in ("latents", "cond", "cloth", "cloth_k"):
        dst[k].copy_(src[k])
    for ck in ("ctx_t", "ctx_g", "ctx_gk"):
        for p, ent in src[ck]["kv"].items():
            for name, t in ent.items():
                dst[ck]["kv"][p][name].copy_(t)


class TryonEngine:
    def __init__(self, unet, unet_encoder, vae, resampler=None, dtype=torch.bfloat16, device="cuda"):
        self.unet, self.unet_encoder, self.vae, self.resampler = unet, unet_encoder, vae, resampler
        self.dtype, self.device = dtype, torch.device(device)
        self._graphs = {}
        self._side = None
        self.garment_steps = 6                               # timesteps per GarmentNet batch (see the module docstring)

    # -------------------------------------------------------------------------------------------- preparation
    @torch.no_grad()
    def prepare(self, *, image, mask_image, pose_img, cloth, prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds,
                negative_pooled_prompt_embeds, text_embeds_cloth, noise, num_inference_steps, guidance_scale,
                ip_hidden_states=None, image_embeds=None, scheduler="ddpm", height=None, width=None, strength=1.0,
                image_dtype=None):
        """Everything before the loop (tryon_pipeline.py:1495-1762).  image in [0,1]; pose_img / cloth in [-1,1];
        noise: dict(latents, masked, pose, cloth [B,4,h,w] fp32; steps [n,B,4,h,w] fp32 or None; image [B,4,h,w] when strength < 1)
        -- RNG order SURVEY A.4.
        strength < 1 (:987-995, 883-893): the last int(n*strength) timesteps, starting from add_noise(encode(image), noise, t_0).
        guidance_scale <= 1 (:440-442: no classifier-free guidance): the reference runs the conditional branch alone; here the
        batched step runs with guidance 1 -- u + 1*(t - u) = t to one fp32 rounding -- so negative_* may be None and
        ip_hidden_states / image_embeds may hold the B conditional rows only."""
        dev, dt = self.device, self.dtype
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        image, mask_image, pose_img, cloth = f32(image), f32(mask_image), f32(pose_img), f32(cloth)
        B = image.shape[0]
        H = height or image.shape[-2]
        W = width or image.shape[-1]
        h, w = H // 8, W // 8
        sched = StepScheduler(scheduler)
        timesteps = sched.set_timesteps(num_inference_steps)                               # :1561
        init_t = min(int(num_inference_steps * strength), num_inference_steps)             # get_timesteps :987-995
        timesteps = timesteps[max(num_inference_steps - init_t, 0):]
        if len(timesteps) < 1:                                                             # :1568-1572
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline"
                             f"steps is {len(timesteps)} which is < 1 and not appropriate for this pipeline.")
        if guidance_scale <= 1:                                                            # no CFG: see the docstring
            guidance_scale = 1.0
            if negative_prompt_embeds is None:
                negative_prompt_embeds = torch.zeros_like(prompt_embeds)
            if negative_pooled_prompt_embeds is None:
                negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)
            if ip_hidden_states is not None and ip_hidden_states.shape[0] == B:
                ip_hidden_states = torch.cat([torch.zeros_like(ip_hidden_states), ip_hidden_states])
            if image_embeds is not None and image_embeds.shape[0] == B:
                image_embeds = torch.cat([torch.zeros_like(image_embeds), image_embeds])

        init_image = 2.0 * image - 1.0                                                     # preprocess :1588-1591
        mask = (mask_image >= 0.5).float()                                                 # mask_processor :1593-1595
        masked_image = init_image * (mask < 0.5)                                           # :1602
        if strength == 1.0 or noise.get("image") is None:
            latents = f32(noise["latents"]) * sched.init_noise_sigma                       # :889-893
        else:                                                                              # image + noise start (:883-891)
            src = init_image if image_dtype is None else init_image.to(image_dtype).float()    # prepare_latents casts the image (:884)
            ab = float(sched.alphas_cumprod[int(timesteps[0])])
            latents = ab ** 0.5 * self.vae.encode_sample(src, f32(noise["image"])) + (1.0 - ab) ** 0.5 * f32(noise["latents"])
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
        # GarmentNet over k consecutive timesteps per batch: batch index = j*B + b (timestep-major), the last block padded by
        # repeating the final timestep (its extra rows are never read)
        n = len(timesteps)
        k = max(1, min(self.garment_steps, n))
        blocks = [(s0, min(k, n - s0)) for s0 in range(0, n, k)]
        tidx = torch.tensor([[min(s0 + j, n - 1) for j in range(k)] for s0, _ in blocks], device=dev)
        temb_gk = temb_g[tidx].reshape(len(blocks), k * B, -1).contiguous()
        cloth_k = cloth_nhwc.repeat(k, 1, 1).contiguous()
        ctx_gk = ctx_g if k == 1 else self.unet_encoder.encode_context(text_embeds_cloth.to(dev).repeat(k, 1, 1))
        coef = torch.tensor([list(sched.coeffs(t)) + [guidance_scale] for t in timesteps], dtype=torch.float32, device=dev)
        steps_noise = f32(noise["steps"]) if noise.get("steps") is not None and scheduler == "ddpm" else None
        return dict(B=B, h=h, w=w, timesteps=timesteps, latents=latents.contiguous(), cond=cond, cloth=cloth_nhwc,
                    ctx_t=ctx_t, ctx_g=ctx_g, temb_t=temb_t, temb_g=temb_g, coef=coef, steps_noise=steps_noise,
                    k=k, blocks=blocks, temb_gk=temb_gk, cloth_k=cloth_k, ctx_gk=ctx_gk,
                    x_in=torch.empty(2 * B, h * w, self.unet.cin_pad, dtype=dt, device=dev),
                    trace=dict(masked_lat=masked_lat, pose_lat=pose_lat, cloth_lat=cloth_lat, image_embeds=image_embeds))

    # -------------------------------------------------------------------------------------------- one step
    def _step(self, st, temb_t, temb_g, coef, noise):
        """One loop iteration in the reference's own order (tryon_pipeline.py:1765-1866: GarmentNet for THIS timestep, then
        TryonNet) on the current stream.  Debugging aid (tools/gpu_debug.py); the loop itself runs in blocks, below."""
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        _, feats = self.unet_encoder.forward(st["cloth"], temb_g, st["ctx_g"], B, h, w)    # :1787
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_feats=feats)   # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    # ---- blocks: one GarmentNet batch over k timesteps + the k TryonNet steps that consume it ------------------------------
    # The GarmentNet batch of block b+1 -- and the attn1 K / V^T projections of its features with TryonNet's weights -- can run on a
    # second HIP stream while TryonNet runs the steps of block b.  Two feature sets alternate; there is no other coupling.
    def _garment_side(self, st, temb_gk, fset):
        B, h, w, k = st["B"], st["h"], st["w"], st["k"]
        self.unet_encoder.forward(st["cloth_k"], temb_gk, st["ctx_gk"], k * B, h, w, feats_buf=fset["feats"])   # :1787, k timesteps
        self.unet.project_garment_kv(fset["feats"], out=fset["kv"])

    def _tryon_main(self, st, temb_t, coef, noise, kv_j):
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_kv=kv_j)        # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    def _new_set(self, st, like=None):
        """A persistent {70 features, 70 (K, V^T)} set for k timesteps + per-timestep views of its K / V^T."""
        B, h, w, k = st["B"], st["h"], st["w"], st["k"]
        if like is None:
            _, feats = self.unet_encoder.forward(st["cloth_k"], st["temb_gk"][0], st["ctx_gk"], k * B, h, w)
            kv = self.unet.project_garment_kv(feats)
        else:
            feats = [torch.empty_like(f) for f in like["feats"]]
            kv = [(torch.empty_like(kk), torch.empty_like(vv)) for kk, vv in like["kv"]]
        per_step = []
        for j in range(k):
            per_step.append([(kk[j * B * (kk.shape[0] // (k * B)):(j + 1) * B * (kk.shape[0] // (k * B))], vv[j * B:(j + 1) * B]) for kk, vv in kv])
        return dict(feats=feats, kv=kv, step=per_step)

    def _noise(self, st, i):
        return st["steps_noise"][i] if st["steps_noise"] is not None else None

    def _denoise_serial_eager(self, st, trace=None):
        fset = None
        for bi, (s0, c) in enumerate(st["blocks"]):
            if fset is None:
                fset = self._new_set(st)                                   # runs block 0's GarmentNet batch
            else:
                self._garment_side(st, st["temb_gk"][bi], fset)
            for j in range(c):
                i = s0 + j
                self._tryon_main(st, st["temb_t"][i], st["coef"][i], self._noise(st, i), fset["step"][j])
                if trace is not None:
                    trace.setdefault("step_latents", []).append(st["latents"].clone())
        return st["latents"]

    def _denoise_overlap_eager(self, st):
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        s0set = self._new_set(st)
        sets = [s0set, self._new_set(st, like=s0set)]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        side.wait_stream(main)                                       # prepare()'s tensors and set 0 are complete
        nb = len(st["blocks"])
        for bi, (s0, c) in enumerate(st["blocks"]):
            cur, nxt = bi & 1, (bi + 1) & 1
            if bi + 1 < nb:
                with torch.cuda.stream(side):
                    if bi >= 1:
                        side.wait_event(free[nxt])                   # TryonNet block bi-1 is done reading set nxt
                    self._garment_side(st, st["temb_gk"][bi + 1], sets[nxt])
                    ready[nxt].record(side)
            if bi >= 1:
                main.wait_event(ready[cur])
            for j in range(c):
                i = s0 + j
                self._tryon_main(st, st["temb_t"][i], st["coef"][i], self._noise(st, i), sets[cur]["step"][j])
            free[cur].record(main)
        main.wait_stream(side)
        return st["latents"]

    def _graph_state(self, st):
        """Persistent buffers + captured graphs for one shape.  The graphs are SMALL: ('garm', p) = the GarmentNet batch into feature
        set p, ('tryon', p, j) = one TryonNet step on timestep slice j of set p (2 + 2k graphs, captured on first use).  The loop
        replays them like the eager form launches kernels -- GarmentNet graphs on the side stream, TryonNet graphs on the main stream,
        two events per set -- so the overlap form has no fork/join inside a graph and a replay never queues more than one step.
        (One graph per 6-step block, GarmentNet as a parallel branch, measured 1.2 % slower than eager launch; this form matches it.)
        Captures use capture_error_mode="thread_local": with torch.distributed / RCCL initialised a watchdog thread polls events,
        which the default global mode would treat as a capture violation."""
        has_noise = st["steps_noise"] is not None
        key = (st["B"], st["h"], st["w"], st["k"], has_noise)
        if key in self._graphs:
            return self._graphs[key]
        tt, cf, tgk = st["temb_t"][0].clone(), st["coef"][0].clone(), st["temb_gk"][0].clone()
        nz = st["steps_noise"][0].clone() if has_noise else None
        saved = st["latents"].clone()
        warm = torch.cuda.Stream()
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):                                # warm-up off the default stream (allocator, lazy init)
            s0set = self._new_set(st)
            sets = [s0set, self._new_set(st, like=s0set)]
            self._garment_side(st, tgk, sets[1])
            self._tryon_main(st, tt, cf, nz, sets[0]["step"][0])
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        st["latents"].copy_(saved)
        G = dict(st=st, tt=tt, cf=cf, nz=nz, tgk=tgk, sets=sets, graphs={}, side=torch.cuda.Stream(),
                 ready=[torch.cuda.Event(), torch.cuda.Event()], free=[torch.cuda.Event(), torch.cuda.Event()],
                 # graphs that replay one after another on ONE stream may share a memory pool: all TryonNet graphs (main stream), all
                 # GarmentNet graphs (side stream)
                 pools=dict(tryon=torch.cuda.graph_pool_handle(), garm=torch.cuda.graph_pool_handle()))
        self._graphs[key] = G
        return G

    def _graph(self, G, kind, par, j=0):
        gk = (kind, par, j)
        if gk in G["graphs"]:
            return G["graphs"][gk]
        st = G["st"]
        keep = st["latents"].clone()
        torch.cuda.synchronize()                                     # nothing of this engine in flight while a capture starts
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=G["pools"][kind], capture_error_mode="thread_local"):
            if kind == "garm":
                self._garment_side(st, G["tgk"], G["sets"][par])
            else:
                self._tryon_main(st, G["tt"], G["cf"], G["nz"], G["sets"][par]["step"][j])
        st["latents"].copy_(keep)                                    # capture does not execute, but keep the state explicit
        G["graphs"][gk] = g
        return g

    def _denoise_graph(self, st, overlap, trace=None):
        G = self._graph_state(st)
        sst = G["st"]
        if sst is not st:
            _copy_state(sst, st)                                                           # new call -> persistent buffers
        blocks, nb, k = st["blocks"], len(st["blocks"]), st["k"]
        # capture everything this call needs before the loop (a capture must not interleave with work in flight on the side stream)
        for p in ((0, 1) if overlap and nb > 1 else (0,)):
            self._graph(G, "garm", p)
            for j in range(k):
                self._graph(G, "tryon", p, j)
        main, side = torch.cuda.current_stream(), G["side"]
        ready, free = G["ready"], G["free"]

        def tryon_block(s0, c, p):
            for j in range(c):
                i = s0 + j
                G["tt"].copy_(st["temb_t"][i]); G["cf"].copy_(st["coef"][i])
                if G["nz"] is not None:
                    G["nz"].copy_(st["steps_noise"][i])
                G["graphs"][("tryon", p, j)].replay()
                if trace is not None:
                    trace.setdefault("step_latents", []).append(sst["latents"].clone())

        if not overlap:
            for bi, (s0, c) in enumerate(blocks):
                G["tgk"].copy_(st["temb_gk"][bi])
                G["graphs"][("garm", 0, 0)].replay()
                tryon_block(s0, c, 0)
            return sst["latents"]
        G["tgk"].copy_(st["temb_gk"][0])
        G["graphs"][("garm", 0, 0)].replay()                          # block 0's features, on the main stream
        side.wait_stream(main)
        for bi, (s0, c) in enumerate(blocks):
            cur, nxt = bi & 1, (bi + 1) & 1
            if bi + 1 < nb:
                with torch.cuda.stream(side):
                    if bi >= 1:
                        side.wait_event(free[nxt])                   # TryonNet block bi-1 is done reading set nxt
                    G["tgk"].copy_(st["temb_gk"][bi + 1])
                    G["graphs"][("garm", nxt, 0)].replay()
                    ready[nxt].record(side)
            if bi >= 1:
                main.wait_event(ready[cur])
            tryon_block(s0, c, cur)
            free[cur].record(main)
        main.wait_stream(side)
        return sst["latents"]

    @torch.no_grad()
    def denoise(self, st, use_graph=False, trace=None, overlap=False):
        """The loop.  Four execution forms with bit-identical results: {serial, two-stream overlap} x {eager, hipGraph replay}."""
        if use_graph:
            return self._denoise_graph(st, overlap, trace)
        return self._denoise_overlap_eager(st) if overlap else self._denoise_serial_eager(st, trace)

    @torch.no_grad()
    def decode(self, latents):
        img = self.vae.decode(latents / self.vae.cfg.scaling_factor)                       # :1876
        return (img / 2 + 0.5).clamp(0, 1)                                                 # postprocess

    @torch.no_grad()
    def __call__(self, *, return_latents=False, use_graph=False, overlap=False, **kw):
        st = self.prepare(**kw)
        lat = self.denoise(st, use_graph=use_graph, overlap=overlap)
        return lat if return_latents else self.decode(lat)
