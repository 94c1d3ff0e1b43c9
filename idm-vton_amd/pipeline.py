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
that GarmentNet batch plus the TryonNet steps that consume its features.  Blocks RAMP UP -- 1, 2, 4, then `garment_steps`
timesteps -- because only the first block's GarmentNet batch cannot hide behind TryonNet work (nothing runs before it): with a
one-timestep first block 7.5 ms of a call are exposed instead of the 45 ms of a six-timestep batch, and each later batch is
shorter than the TryonNet steps of the block before it.  Every batch runs at its exact size (no padded timesteps).
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
        self._set_shapes = {}
        self._side = None
        self.garment_steps = 6                               # timesteps per GarmentNet batch (see the module docstring)
        self.ramp = True                                     # first blocks of 1, 2, 4 timesteps

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
        # any H x W divisible by 8 (the reference's own check, tryon_pipeline.py check_inputs): odd latent levels go through `upsample_size`
        # (unet.py: _conv3 out_hw), token counts that are not a multiple of 16 are padded inside each Transformer2DModel (unet.py: _transformer)
        if H % 8 or W % 8:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {H} and {W}.")
        start_is_given = bool(noise.get("latents_given"))    # the reference's `latents=` argument: used as the start as they are (:880-882)
        if strength < 1.0 and noise.get("image") is None and not start_is_given:
            raise ValueError("strength < 1 starts from add_noise(encode(image)): pass the posterior draw of the init-image encode as "
                             "noise['image'] (the reference's first random draw, tryon_pipeline.py:883-889), or set noise['latents_given'] "
                             "when noise['latents'] already is the start (the reference's `latents=` argument)")
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
        if strength == 1.0 or start_is_given:
            latents = f32(noise["latents"]) * sched.init_noise_sigma                       # :889-893
        else:                                                                              # image + noise start (:883-891)
            src = init_image if image_dtype is None else init_image.to(image_dtype).float()    # prepare_latents casts the image (:884)
            ab = float(sched.alphas_cumprod[int(timesteps[0])])
            latents = ab ** 0.5 * self.vae.encode_sample(src, f32(noise["image"])) + (1.0 - ab) ** 0.5 * f32(noise["latents"])
        mask_l = torch.nn.functional.interpolate(mask, size=(h, w))                        # :939-941
        # the three VAE encodes of the call (:964 masked image, :1644-1647 pose, :1654 cloth) as ONE encoder pass over 3B images: per
        # image the arithmetic is unchanged (GroupNorm / attention are per image), the convolution GEMMs see 3x the rows and the
        # launch count of the encoder is paid once
        enc = self.vae.encode_sample(torch.cat([masked_image, pose_img, cloth]),
                                     torch.cat([f32(noise["masked"]), f32(noise["pose"]), f32(noise["cloth"])]))
        masked_lat, pose_lat, cloth_lat = enc[:B], enc[B:2 * B], enc[2 * B:]
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
        # GarmentNet over consecutive timesteps per batch: batch index = j*B + b (timestep-major).  Block sizes ramp 1, 2, 4, k, k, ...
        # (module docstring); temb_gk rows beyond a block's own c*B are never read
        n = len(timesteps)
        k = max(1, min(self.garment_steps, n))
        sizes, s0 = [], 0
        for c in (1, 2, 4):
            if c < k and s0 + c <= n and self.ramp:
                sizes.append(c); s0 += c
        while s0 < n:
            sizes.append(min(k, n - s0)); s0 += sizes[-1]
        blocks, s0 = [], 0
        for c in sizes:
            blocks.append((s0, c)); s0 += c
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
    def _garment_side(self, st, temb_gk, fset, c=None):
        B, h, w = st["B"], st["h"], st["w"]
        c = st["k"] if c is None else c                      # timesteps in this batch (the set's buffers hold up to st["k"])
        _, feats = self.unet_encoder.forward(st["cloth_k"][:c * B], temb_gk[:c * B], st["ctx_gk"], c * B, h, w, feats_buf=fset["feats"])   # :1787
        self.unet.project_garment_kv(feats, out=fset["kv"])

    def _tryon_main(self, st, temb_t, coef, noise, kv_j):
        B, h, w = st["B"], st["h"], st["w"]
        ops.pack_input(st["latents"], st["cond"], st["x_in"])                              # :1769,1777
        eps, _ = self.unet.forward(st["x_in"], temb_t, st["ctx_t"], 2 * B, h, w, garment_kv=kv_j)        # :1796-1808
        ops.cfg_step(eps, st["latents"], noise, coef)                                      # :1814-1823
        return eps

    def _new_set(self, st, like=None, run=True):
        """A persistent {70 features, 70 (K, V^T)} set for up to k timesteps + per-timestep views of its K / V^T.  The tensor shapes
        are discovered once per (B, h, w, k) by running a GarmentNet batch (cached); after that a set is a plain allocation."""
        B, h, w, k = st["B"], st["h"], st["w"], st["k"]
        key = (B, h, w, k)
        if like is None and key not in self._set_shapes:
            _, feats = self.unet_encoder.forward(st["cloth_k"], st["temb_gk"][0], st["ctx_gk"], k * B, h, w)
            kv = self.unet.project_garment_kv(feats)
            self._set_shapes[key] = ([tuple(f.shape) for f in feats], [(tuple(kk.shape), tuple(vv.shape), kk.dtype) for kk, vv in kv])
        elif like is None:
            fs, ks = self._set_shapes[key]
            feats = [torch.empty(sh, dtype=self.dtype, device=self.device) for sh in fs]
            kv = [(torch.empty(a, dtype=d, device=self.device), torch.empty(b, dtype=d, device=self.device)) for a, b, d in ks]   # d: uint8 = e4m3 (attn_fp8)
        else:
            feats = [torch.empty_like(f) for f in like["feats"]]
            kv = [(torch.empty_like(kk), torch.empty_like(vv)) for kk, vv in like["kv"]]
        per_step = []
        for j in range(k):
            per_step.append([(kk[j * B * (kk.shape[0] // (k * B)):(j + 1) * B * (kk.shape[0] // (k * B))], vv[j * B:(j + 1) * B]) for kk, vv in kv])
        return dict(feats=feats, kv=kv, step=per_step)

    def _noise(self, st, i):
        return st["steps_noise"][i] if st["steps_noise"] is not None else None

    def _denoise_serial_eager(self, st, trace=None, on_step=None):
        """on_step(i, t, latents) runs after step i on the live latents (it may rewrite them in place); a true return value ends the loop
        (the reference's per-step callbacks and `interrupt`, tryon_pipeline.py:1766-1767,1840-1863: host code between two steps, which
        only this un-captured form can run)."""
        fset = None
        for bi, (s0, c) in enumerate(st["blocks"]):
            if fset is None:
                fset = self._new_set(st)
            self._garment_side(st, st["temb_gk"][bi], fset, c)
            for j in range(c):
                i = s0 + j
                self._tryon_main(st, st["temb_t"][i], st["coef"][i], self._noise(st, i), fset["step"][j])
                if trace is not None:
                    trace.setdefault("step_latents", []).append(st["latents"].clone())
                if on_step is not None and on_step(i, int(st["timesteps"][i]), st["latents"]):
                    return st["latents"]
        return st["latents"]

    def _denoise_overlap_eager(self, st, trace=None):
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        s0set = self._new_set(st)
        sets = [s0set, self._new_set(st, like=s0set)]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        self._garment_side(st, st["temb_gk"][0], sets[0], st["blocks"][0][1])      # block 0's batch: nothing to hide behind
        side.wait_stream(main)                                       # prepare()'s tensors and set 0 are complete
        nb = len(st["blocks"])
        for bi, (s0, c) in enumerate(st["blocks"]):
            cur, nxt = bi & 1, (bi + 1) & 1
            if bi + 1 < nb:
                with torch.cuda.stream(side):
                    if bi >= 1:
                        side.wait_event(free[nxt])                   # TryonNet block bi-1 is done reading set nxt
                    self._garment_side(st, st["temb_gk"][bi + 1], sets[nxt], st["blocks"][bi + 1][1])
                    ready[nxt].record(side)
            if bi >= 1:
                main.wait_event(ready[cur])
            for j in range(c):
                i = s0 + j
                self._tryon_main(st, st["temb_t"][i], st["coef"][i], self._noise(st, i), sets[cur]["step"][j])
                if trace is not None:
                    trace.setdefault("step_latents", []).append(st["latents"].clone())
            free[cur].record(main)
        main.wait_stream(side)
        return st["latents"]

    def _graph_state(self, st):
        """Persistent buffers + captured graphs for one shape.  The graphs are SMALL: ('garm', p, c) = the GarmentNet batch into feature
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
            if kind == "garm":                                   # j = timesteps in the batch
                self._garment_side(st, G["tgk"], G["sets"][par], j)
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
            for j in range(k):
                self._graph(G, "tryon", p, j)
        for bi, (_, c) in enumerate(blocks):                          # one GarmentNet graph per (set, batch size)
            self._graph(G, "garm", (bi & 1) if overlap else 0, c)
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
                G["graphs"][("garm", 0, c)].replay()
                tryon_block(s0, c, 0)
            return sst["latents"]
        G["tgk"].copy_(st["temb_gk"][0])
        G["graphs"][("garm", 0, blocks[0][1])].replay()               # block 0's features, on the main stream
        side.wait_stream(main)
        for bi, (s0, c) in enumerate(blocks):
            cur, nxt = bi & 1, (bi + 1) & 1
            if bi + 1 < nb:
                with torch.cuda.stream(side):
                    if bi >= 1:
                        side.wait_event(free[nxt])                   # TryonNet block bi-1 is done reading set nxt
                    G["tgk"].copy_(st["temb_gk"][bi + 1])
                    G["graphs"][("garm", nxt, blocks[bi + 1][1])].replay()
                    ready[nxt].record(side)
            if bi >= 1:
                main.wait_event(ready[cur])
            tryon_block(s0, c, cur)
            free[cur].record(main)
        main.wait_stream(side)
        return sst["latents"]

    @torch.no_grad()
    def denoise(self, st, use_graph=False, trace=None, overlap=False, on_step=None):
        """The loop.  Four execution forms with bit-identical results: {serial, two-stream overlap} x {eager, hipGraph replay}; a per-step
        host hook (`on_step`) selects the serial eager form."""
        if on_step is not None:
            return self._denoise_serial_eager(st, trace, on_step)
        if use_graph:
            return self._denoise_graph(st, overlap, trace)
        return self._denoise_overlap_eager(st, trace) if overlap else self._denoise_serial_eager(st, trace)

    @torch.no_grad()
    def decode(self, latents):
        img = self.vae.decode(latents / self.vae.cfg.scaling_factor)                       # :1876
        return (img / 2 + 0.5).clamp(0, 1)                                                 # postprocess

    @torch.no_grad()
    def __call__(self, *, return_latents=False, use_graph=False, overlap=False, timing=None, on_step=None, **kw):
        """timing: a list that receives one (start, prepared, denoised, decoded) tuple of HIP events recorded on the current stream
        (bench.py: the loop's share of a timed call without a second, instrumented run)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timing is not None else None
        if ev:
            ev[0].record()
        st = self.prepare(**kw)
        if ev:
            ev[1].record()
        lat = self.denoise(st, use_graph=use_graph, overlap=overlap, on_step=on_step)
        if ev:
            ev[2].record()
        out = lat if return_latents else self.decode(lat)
        if ev:
            ev[3].record()
            timing.append(tuple(ev))
        return out
