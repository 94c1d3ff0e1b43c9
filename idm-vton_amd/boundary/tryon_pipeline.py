"""`src.tryon_pipeline.StableDiffusionXLInpaintPipeline` on the MI355X engine.

Mirror of the reference class (/root/reference/src/tryon_pipeline.py:309; ctor :387-422, encode_image :460-482,
prepare_ip_adapter_image_embeds :485-507, encode_prompt :511-743, check_inputs :763-848, __call__ :1254-1894) for the path
inference.py:316-414 / gradio_demo/app.py:111-234 take: same constructor components, `from_pretrained(path, unet=, vae=, ...)`,
`.to(device)`, `.device`, `encode_prompt(...)` -> 4-tuple, `__call__(...)` keyword surface -> `(list[PIL.Image],)`.

What runs where: the two CLIP text encoders and the CLIP-H image encoder are the transformers modules the caller passes; when
they are transformers CLIP towers living on the GPU their forward is executed by idm_vton_amd.clip on the HIP kernels (SURVEY.md
8a row a17 / 8f-1; once per call, outside the loop), any other module is called as is.  Everything from the VAE encodes to the VAE decode runs on the HIP kernels through idm_vton_amd.pipeline.TryonEngine:
3 VAE encodes, Resampler, hoisted K/V + embedding tables, the denoising loop (GarmentNet || TryonNet on two streams,
hipGraph replay) and the decode.  RNG draws follow the reference's order with the caller's generator.

guidance_scale <= 1 (no CFG) runs the batched step with guidance 1 (the conditional prediction to one fp32 rounding, at the cost of
the unused half); strength < 1 starts from the noised image latents as the reference does.
Loud differences: `padding_mask_crop`, `masked_image_latents=`, `timesteps=`,
`denoising_start/end`, `guidance_rescale`, `cross_attention_kwargs`, step callbacks, `num_images_per_prompt != 1` and
schedulers other than DDPM/DDIM raise NotImplementedError: the reference scripts use none of them.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from .. import ffi
from ..clip import HipCLIPText, HipCLIPVision
from ..pipeline import TryonEngine
from .modules import params_version
from .scheduler import DDIMScheduler, DDPMScheduler
from .unet import GarmentUNet2DConditionModel, TryonUNet2DConditionModel
from .vae import AutoencoderKL


def _to_tensor_image(x, name, lo_hi=None, size=None):
    """PIL / list of PIL / ndarray / tensor -> float32 tensor [B,C,H,W] (VaeImageProcessor.preprocess semantics, B.6).
    size = (height, width): `preprocess(image, height=, width=)` of the reference (:1588-1595) RESIZES to it -- PIL images with PIL's
    lanczos filter (VaeImageProcessor.resize, config.resample default), tensors with F.interpolate's default (nearest)."""
    import PIL.Image
    if isinstance(x, PIL.Image.Image):
        x = [x]
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], PIL.Image.Image):
        if size is not None:
            x = [im if (im.height, im.width) == tuple(size) else im.resize((size[1], size[0]), resample=PIL.Image.LANCZOS) for im in x]
        arr = np.stack([np.asarray(im.convert("RGB") if im.mode != "L" else im, dtype=np.float32) / 255.0 for im in x])
        if arr.ndim == 3:
            arr = arr[..., None]
        x = torch.from_numpy(arr).permute(0, 3, 1, 2)
    elif isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    elif isinstance(x, (list, tuple)):
        x = torch.stack([torch.as_tensor(t) for t in x])
    if not torch.is_tensor(x):
        raise ValueError(f"`{name}` has to be a tensor, PIL image, ndarray or list of those, got {type(x)}")
    if x.ndim == 3:
        x = x.unsqueeze(0)
    x = x.float()
    if size is not None and tuple(x.shape[-2:]) != tuple(size):
        x = torch.nn.functional.interpolate(x, size=tuple(size))
    return x


def _randn(shape, generator, device, dtype):
    """diffusers.utils.torch_utils.randn_tensor semantics, result as fp32 on `device`."""
    dev = torch.device(device)
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size "
                             f"of {shape[0]}. Make sure the batch size matches the length of the generators.")
        return torch.cat([_randn((1,) + tuple(shape[1:]), g, device, dtype) for g in generator], dim=0)
    rand_device = dev
    if generator is not None and generator.device.type != dev.type and generator.device.type == "cpu":
        rand_device = torch.device("cpu")
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(dev).float()


class StableDiffusionXLInpaintPipeline:
    _optional_components = ["tokenizer", "tokenizer_2", "text_encoder", "text_encoder_2", "image_encoder",
                            "feature_extractor", "unet_encoder"]

    def __init__(self, vae, text_encoder, text_encoder_2, tokenizer, tokenizer_2, unet, unet_encoder, scheduler,
                 image_encoder=None, feature_extractor=None, requires_aesthetics_score=False,
                 force_zeros_for_empty_prompt=True):
        self.vae, self.text_encoder, self.text_encoder_2 = vae, text_encoder, text_encoder_2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.unet, self.unet_encoder, self.scheduler = unet, unet_encoder, scheduler
        self.image_encoder, self.feature_extractor = image_encoder, feature_extractor
        self.config = SimpleNamespace(force_zeros_for_empty_prompt=force_zeros_for_empty_prompt,
                                      requires_aesthetics_score=requires_aesthetics_score)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        self._device = torch.device("cpu")
        self._engine, self._engine_key = None, None
        self._clip = {}                                     # name -> (params_version, HipCLIPText | HipCLIPVision)
        self._guidance_scale = 7.5
        self.use_graph, self.overlap = True, True          # engine execution mode (hipGraph replay, two-stream loop)

    # ------------------------------------------------------------------------------------------ construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=torch.float32, **components):
        """Components passed as keyword arguments override loading (inference.py:316-329).  unet / unet_encoder / vae /
        scheduler are loaded from `<path>/<name>/` when not given; the transformers components (text encoders, tokenizers,
        image encoder, feature extractor) are loaded with transformers' own from_pretrained when their folder exists."""
        root = pretrained_model_name_or_path
        c = dict(components)

        def sub(name):
            return os.path.isdir(os.path.join(root, name))

        if "unet" not in c:
            c["unet"] = TryonUNet2DConditionModel.from_pretrained(root, subfolder="unet", torch_dtype=torch_dtype)
        if "unet_encoder" not in c:
            c["unet_encoder"] = (GarmentUNet2DConditionModel.from_pretrained(root, subfolder="unet_encoder", torch_dtype=torch_dtype)
                                 if sub("unet_encoder") else None)
        if "vae" not in c:
            c["vae"] = AutoencoderKL.from_pretrained(root, subfolder="vae", torch_dtype=torch_dtype)
        if "scheduler" not in c:
            c["scheduler"] = DDPMScheduler.from_pretrained(root, subfolder="scheduler")
        tf_loaders = dict(text_encoder="CLIPTextModel", text_encoder_2="CLIPTextModelWithProjection",
                          image_encoder="CLIPVisionModelWithProjection", tokenizer="AutoTokenizer", tokenizer_2="AutoTokenizer",
                          feature_extractor="CLIPImageProcessor")
        for name, cls_name in tf_loaders.items():
            if name in c:
                continue
            c[name] = None
            if sub(name):
                import transformers
                kw = dict(torch_dtype=torch_dtype) if "Model" in cls_name else {}
                c[name] = getattr(transformers, cls_name).from_pretrained(root, subfolder=name, **kw)
        extra = {}
        mi = os.path.join(root, "model_index.json")
        if os.path.isfile(mi):
            raw = json.load(open(mi))
            extra = {k: raw[k] for k in ("requires_aesthetics_score", "force_zeros_for_empty_prompt") if k in raw}
        return cls(**c, **extra)

    def to(self, device=None, dtype=None):
        for name in ("vae", "text_encoder", "text_encoder_2", "unet", "unet_encoder", "image_encoder"):
            m = getattr(self, name)
            if m is not None and hasattr(m, "to"):
                m.to(device) if dtype is None else m.to(device, dtype)
        if device is not None:
            self._device = torch.device(device)
        return self

    @property
    def device(self):
        return self._device

    # ------------------------------------------------------------------------------------------ memory toggles of the call surface
    # /root/reference/src/tryon_pipeline.py:427-457 forward these to the VAE.  The HIP VAE has nothing to switch: it already walks its batch in
    # chunks below the 2 GiB a buffer descriptor addresses (idm_vton_amd/vae.py) and 288 GB of HBM need no tiling; the methods exist so that a
    # caller that toggles them keeps working, and they record the request on the VAE like diffusers does (`use_slicing` / `use_tiling`).
    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def enable_vae_tiling(self):
        self.vae.enable_tiling()

    def disable_vae_tiling(self):
        self.vae.disable_tiling()

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None

    # ------------------------------------------------------------------------------------------ conditioning encoders
    def _hip_clip(self, name):
        """The HIP executor (idm_vton_amd.clip) of a transformers CLIP tower that lives on the GPU: prepared weights are cached and
        rebuilt when the module's parameters change.  None for a module on the CPU or one that is not a transformers CLIP tower
        (a caller's own encoder is called as is).  An fp32 tower is executed with bf16 STORAGE and fp32 accumulation (the kernels take
        16-bit operands; bf16 keeps fp32's exponent range): hidden states differ from an fp32 torch run by bf16 rounding, ~1e-2
        relative (tests/test_clip_gpu.py), where the reference would have run that module in fp32 -- pass it in fp16 (as inference.py
        does, :262-274) for fp16 rounding instead."""
        mod = getattr(self, name)
        kind = getattr(getattr(mod, "config", None), "model_type", None)
        if kind not in ("clip_text_model", "clip_vision_model") or not hasattr(mod, "parameters"):
            return None                                   # a caller's own encoder (any callable): called as is, never inspected
        p0 = next(mod.parameters(), None)
        if p0 is None or not p0.is_cuda:
            return None
        ffi.lib()
        key = params_version(mod)
        hit = self._clip.get(name)
        if hit is None or hit[0] != key:
            dt = p0.dtype if p0.dtype in (torch.float16, torch.bfloat16) else torch.bfloat16     # fp32 module: bf16 storage, as the VAE
            cls = HipCLIPVision if kind == "clip_vision_model" else HipCLIPText
            hit = self._clip[name] = (key, cls(mod.state_dict(), mod.config, dt, p0.device))
        return hit[1]

    def encode_image(self, image, device, num_images_per_prompt, output_hidden_states=None):
        """CLIP-H penultimate hidden states of the garment image and of an all-zero image (reference :460-482)."""
        dtype = next(self.image_encoder.parameters()).dtype
        if not isinstance(image, torch.Tensor):
            image = self.feature_extractor(image, return_tensors="pt").pixel_values
        image = image.to(device=device, dtype=dtype)
        hip = self._hip_clip("image_encoder")
        if output_hidden_states:
            if hip is not None:                                                  # both images in one pass of the tower
                hs = hip(torch.cat([image, torch.zeros_like(image)]), penultimate_only=True).hidden_states[-2].to(dtype)
                pos, neg = hs[:image.shape[0]], hs[image.shape[0]:]
            else:
                pos = self.image_encoder(image, output_hidden_states=True).hidden_states[-2]
                neg = self.image_encoder(torch.zeros_like(image), output_hidden_states=True).hidden_states[-2]
            return (pos.repeat_interleave(num_images_per_prompt, dim=0), neg.repeat_interleave(num_images_per_prompt, dim=0))
        emb = (hip(image).image_embeds.to(dtype) if hip is not None else self.image_encoder(image).image_embeds)
        emb = emb.repeat_interleave(num_images_per_prompt, dim=0)
        return emb, torch.zeros_like(emb)

    def prepare_ip_adapter_image_embeds(self, ip_adapter_image, device, num_images_per_prompt):
        """cat([uncond, cond]) CLIP states [2B,257,1280] for the Resampler (reference :485-507); the Resampler is not an
        `ImageProjection`, so hidden states (not pooled embeds) are used."""
        pos, neg = self.encode_image(ip_adapter_image, device, 1, True)
        return torch.cat([neg, pos]).to(device) if self.do_classifier_free_guidance else pos

    def _encode_text(self, texts, device, max_length=None):
        tokenizers = [self.tokenizer, self.tokenizer_2] if self.tokenizer is not None else [self.tokenizer_2]
        encoders = [self.text_encoder, self.text_encoder_2] if self.text_encoder is not None else [self.text_encoder_2]
        if len(texts) == 2 and len(tokenizers) == 1:
            texts = texts[1:]
        hidden, pooled = [], None
        names = ["text_encoder", "text_encoder_2"][-len(encoders):]
        for i, (text, tok, enc, name) in enumerate(zip(texts, tokenizers, encoders, names)):
            ids = tok(text, padding="max_length", max_length=max_length or tok.model_max_length, truncation=True,
                      return_tensors="pt").input_ids
            hip = self._hip_clip(name)
            if hip is not None:
                last_tower = i == len(encoders) - 1           # earlier towers: only hidden_states[-2] is consumed
                out = hip(ids.to(device), penultimate_only=not last_tower)
                pooled = out.first.to(enc.dtype) if last_tower else None
                hidden.append(out.hidden_states[-2].to(enc.dtype))
                continue
            out = enc(ids.to(device), output_hidden_states=True)
            pooled = out[0]                               # the LAST encoder's first output: pooled text_embeds of encoder 2
            hidden.append(out.hidden_states[-2])
        return torch.cat(hidden, dim=-1), pooled

    def encode_prompt(self, prompt, prompt_2=None, device=None, num_images_per_prompt=1, do_classifier_free_guidance=True,
                      negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, negative_prompt_embeds=None,
                      pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, lora_scale=None, clip_skip=None):
        """-> (prompt_embeds [B,77,2048], negative_prompt_embeds, pooled_prompt_embeds [B,1280], negative_pooled) as the
        reference (:511-743): penultimate hidden states of both CLIP text encoders concatenated, pooled output of the second;
        an absent negative prompt gives zeros when force_zeros_for_empty_prompt, else the encoding of ""."""
        if lora_scale is not None or clip_skip is not None:
            raise NotImplementedError("lora_scale / clip_skip are not used by the try-on scripts")
        device = device or self._device
        if prompt is not None:
            prompt = [prompt] if isinstance(prompt, str) else list(prompt)
            batch = len(prompt)
        else:
            batch = prompt_embeds.shape[0]
        if prompt_embeds is None:
            p2 = prompt_2 or prompt
            p2 = [p2] if isinstance(p2, str) else list(p2)
            prompt_embeds, pooled_prompt_embeds = self._encode_text([prompt, p2], device)
        zero_neg = negative_prompt is None and self.config.force_zeros_for_empty_prompt
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if zero_neg:
                negative_prompt_embeds = torch.zeros_like(prompt_embeds)
                negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)
            else:
                neg = negative_prompt or ""
                neg2 = negative_prompt_2 or neg
                neg = batch * [neg] if isinstance(neg, str) else list(neg)
                neg2 = batch * [neg2] if isinstance(neg2, str) else list(neg2)
                if prompt is not None and type(prompt) is not type(neg):
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(neg)} != {type(prompt)}.")
                if batch != len(neg):
                    raise ValueError(f"`negative_prompt`: {neg} has batch size {len(neg)}, but `prompt`: {prompt} has batch size "
                                     f"{batch}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
                negative_prompt_embeds, negative_pooled_prompt_embeds = self._encode_text([neg, neg2], device,
                                                                                          max_length=prompt_embeds.shape[1])
        dt = self.text_encoder_2.dtype if self.text_encoder_2 is not None else self.unet.dtype
        rep = lambda t: t.to(dtype=dt, device=device).repeat_interleave(num_images_per_prompt, dim=0)
        prompt_embeds, pooled_prompt_embeds = rep(prompt_embeds), rep(pooled_prompt_embeds)
        if do_classifier_free_guidance:
            negative_prompt_embeds, negative_pooled_prompt_embeds = rep(negative_prompt_embeds), rep(negative_pooled_prompt_embeds)
        return prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds

    # ------------------------------------------------------------------------------------------ argument checks
    def check_inputs(self, prompt, prompt_2, image, mask_image, height, width, strength, callback_steps, output_type,
                     negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, negative_prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, padding_mask_crop=None):
        """Same conditions and ValueErrors as the reference (:763-848)."""
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to only "
                             "forward one of the two.")
        if prompt_2 is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt_2`: {prompt_2} and `prompt_embeds`: {prompt_embeds}. Please make sure to "
                             "only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt_2 is not None and not isinstance(prompt_2, (str, list)):
            raise ValueError(f"`prompt_2` has to be of type `str` or `list` but is {type(prompt_2)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt`: {negative_prompt} and `negative_prompt_embeds`: "
                             f"{negative_prompt_embeds}. Please make sure to only forward one of the two.")
        if negative_prompt_2 is not None and negative_prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `negative_prompt_2`: {negative_prompt_2} and `negative_prompt_embeds`: "
                             f"{negative_prompt_embeds}. Please make sure to only forward one of the two.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but got: "
                             f"`prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")

    # ------------------------------------------------------------------------------------------ engine
    def hip_engine(self):
        """TryonEngine over the components' prepared HIP weights (rebuilt only when a component's parameters change)."""
        ffi.lib()
        if self.unet_encoder is None:
            raise ValueError("`unet_encoder` (GarmentNet) is required: pass it to from_pretrained or assign pipe.unet_encoder "
                             "(gradio_demo/app.py:125)")
        t, g, v = self.unet.hip_engine(), self.unet_encoder.hip_engine(), self.vae.hip_engine()
        if t.dtype != g.dtype:
            raise TypeError(f"unet ({t.dtype}) and unet_encoder ({g.dtype}) must share one storage dtype")
        proj = self.unet.encoder_hid_proj
        res = proj._engine(t.dtype, t.device) if proj is not None else None
        key = (id(t), id(g), id(v), id(res))
        if key != self._engine_key:
            self._engine = TryonEngine(t, g, v, res, t.dtype, t.device)
            self._engine_key = key
        return self._engine

    # ------------------------------------------------------------------------------------------ the call
    @torch.no_grad()
    def __call__(self, prompt=None, prompt_2=None, image=None, mask_image=None, masked_image_latents=None, height=None,
                 width=None, padding_mask_crop=None, strength=0.9999, num_inference_steps=50, timesteps=None,
                 denoising_start=None, denoising_end=None, guidance_scale=7.5, negative_prompt=None, negative_prompt_2=None,
                 num_images_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 ip_adapter_image=None, output_type="pil", cloth=None, pose_img=None, text_embeds_cloth=None, return_dict=True,
                 cross_attention_kwargs=None, guidance_rescale=0.0, original_size=None, crops_coords_top_left=(0, 0),
                 target_size=None, negative_original_size=None, negative_crops_coords_top_left=(0, 0),
                 negative_target_size=None, aesthetic_score=6.0, negative_aesthetic_score=2.5, clip_skip=None,
                 pooled_prompt_embeds_c=None, callback_on_step_end=None, callback_on_step_end_tensor_inputs=["latents"],
                 **kwargs):
        callback_steps = kwargs.pop("callback_steps", None)
        callback = kwargs.pop("callback", None)              # deprecated form, still honoured by the reference (:1855-1863)
        self._interrupt = False                              # :1519
        if callback_on_step_end is not None and any(k != "latents" for k in (callback_on_step_end_tensor_inputs or [])):
            raise NotImplementedError("callback_on_step_end_tensor_inputs other than 'latents': the text / image conditioning is projected once "
                                      "per call (step-invariant K/V tables), a callback cannot replace it between two steps")
        for name, val, default in (("masked_image_latents", masked_image_latents, None), ("padding_mask_crop", padding_mask_crop, None),
                                   ("timesteps", timesteps, None), ("denoising_start", denoising_start, None),
                                   ("denoising_end", denoising_end, None), ("cross_attention_kwargs", cross_attention_kwargs, None),
                                   ("clip_skip", clip_skip, None)):
            if val is not default:
                raise NotImplementedError(f"`{name}` is not used by the try-on scripts and not supported by the HIP engine")
        if guidance_rescale != 0.0:
            raise NotImplementedError("guidance_rescale != 0 is not supported")
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt != 1 is not supported")
        if self.config.requires_aesthetics_score:
            raise NotImplementedError("requires_aesthetics_score (refiner-style conditioning) is not on the try-on path")
        if image is None:
            raise ValueError("`image` input cannot be undefined.")
        if mask_image is None:
            raise ValueError("`mask_image` input cannot be undefined.")
        for name, val in (("cloth", cloth), ("pose_img", pose_img), ("text_embeds_cloth", text_embeds_cloth),
                          ("ip_adapter_image", ip_adapter_image)):
            if val is None:
                raise ValueError(f"`{name}` is required by the try-on pipeline (inference.py:397-414)")

        height = height or self.unet.config.sample_size * self.vae_scale_factor                  # :1486-1487
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, prompt_2, image, mask_image, height, width, strength, callback_steps, output_type,
                          negative_prompt, negative_prompt_2, prompt_embeds, negative_prompt_embeds,
                          callback_on_step_end_tensor_inputs, padding_mask_crop)
        self._guidance_scale = guidance_scale
        cfg = self.do_classifier_free_guidance              # guidance_scale <= 1: the engine runs its batched step with guidance 1
        # strength < 1 (reference :987-995, 883-893): the last int(n*strength) timesteps, from the noised image latents (the signature
        # default 0.9999 already drops one step; inference.py:404 and gradio_demo/app.py:225 pass 1.0)
        n_exec = min(int(num_inference_steps * strength), num_inference_steps)
        if n_exec < 1:                                      # :1568-1572
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline"
                             f"steps is {n_exec} which is < 1 and not appropriate for this pipeline.")
        kind = {DDPMScheduler: "ddpm", DDIMScheduler: "ddim"}.get(type(self.scheduler))
        if kind is None:
            kind = {"DDPMScheduler": "ddpm", "DDIMScheduler": "ddim"}.get(type(self.scheduler).__name__)
        if kind is None:
            raise NotImplementedError(f"scheduler {type(self.scheduler).__name__}: the fused step kernel implements DDPM and DDIM(eta=0)")
        if kind == "ddim" and eta != 0.0:
            raise NotImplementedError("DDIM with eta != 0")

        eng = self.hip_engine()
        device = eng.device
        (prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds) = self.encode_prompt(
            prompt=prompt, prompt_2=prompt_2, device=device, num_images_per_prompt=1, do_classifier_free_guidance=cfg,
            negative_prompt=negative_prompt, negative_prompt_2=negative_prompt_2, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
            negative_pooled_prompt_embeds=negative_pooled_prompt_embeds)                         # :1541

        if not cfg:                                         # :1710-1711 never read them without CFG
            negative_prompt_embeds = negative_pooled_prompt_embeds = None
        img = _to_tensor_image(image, "image", size=(height, width))                             # :1588-1591: resized to (height, width)
        msk = _to_tensor_image(mask_image, "mask_image", size=(height, width))                   # :1593-1595
        if msk.shape[1] != 1:
            msk = msk.mean(dim=1, keepdim=True)                                                  # do_convert_grayscale
        pose = _to_tensor_image(pose_img, "pose_img")
        clo = _to_tensor_image(cloth, "cloth")
        B = img.shape[0]
        for name, t in (("pose_img", pose), ("cloth", clo)):                                     # encoded as they are (:1644-1654): no resize there
            if tuple(t.shape[-2:]) != (height, width):
                raise ValueError(f"`{name}` is {tuple(t.shape[-2:])} but (height, width) = {(height, width)}: the reference encodes it as it "
                                 "is and fails when its latents are concatenated with the image's")
        if img.min() < 0:
            raise ValueError("`image` is expected in [0, 1] (inference.py:408 passes (image + 1) / 2)")
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor

        # RNG consumption order of the reference: latents, masked-image posterior, pose posterior (GLOBAL
        # generator: tryon_pipeline.py:1646 passes none), cloth posterior, then one draw per DDPM step with t > 0
        # Each draw mirrors diffusers' randn_tensor as the reference calls it: on the GENERATOR's device (a CPU generator draws
        # on the CPU and the result is moved; a list of generators draws per sample) and in the dtype the reference draws in --
        # initial latents: prompt_embeds.dtype (:1599-1610 -> prepare_latents :889); VAE posterior samples: fp32, the VAE being
        # upcast (:913-915, DiagonalGaussianDistribution.sample); DDPM variance noise: the model output's dtype (DDPMScheduler.step).
        shape = (B, 4, h, w)
        # strength < 1 and no `latents=`: the init image is encoded first (prepare_latents :883-886 -> posterior draw), then the noise
        n_img = _randn(shape, generator, device, torch.float32) if (strength != 1.0 and latents is None) else None
        n_lat = latents.to(device).float() if latents is not None else _randn(shape, generator, device, prompt_embeds.dtype)
        n_masked, n_pose, n_cloth = (_randn(shape, generator, device, torch.float32), _randn(shape, None, device, torch.float32),
                                     _randn(shape, generator, device, torch.float32))
        steps_noise = None
        if kind == "ddpm":
            steps_noise = torch.stack([_randn(shape, generator, device, eng.dtype) for _ in range(n_exec)])
        image_states = self.prepare_ip_adapter_image_embeds(ip_adapter_image, device, 1)        # :1720-1723

        call = dict(image=img, mask_image=msk, pose_img=pose, cloth=clo, prompt_embeds=prompt_embeds,
                    negative_prompt_embeds=negative_prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                    negative_pooled_prompt_embeds=negative_pooled_prompt_embeds, text_embeds_cloth=text_embeds_cloth,
                    noise=dict(latents=n_lat, masked=n_masked, pose=n_pose, cloth=n_cloth, steps=steps_noise, image=n_img,
                               latents_given=latents is not None),
                    num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, ip_hidden_states=image_states,
                    strength=strength, scheduler=kind, height=height, width=width)
        if isinstance(getattr(self, "trace_call", None), dict):    # test hook: what crossed the engine boundary (an oracle can replay it)
            self.trace_call.update(call)
        on_step = None
        if callback_on_step_end is not None or callback is not None:
            # host code between two steps (tryon_pipeline.py:1840-1863) and `interrupt` (:1766-1767): the engine runs its serial un-captured loop
            # (same kernels, bit-identical latents: tests/test_parity_gpu.py) and hands the live fp32 latents over after every step
            def on_step(i, t, lat, _dt=prompt_embeds.dtype):
                tt = torch.tensor(t, device=lat.device)
                if callback_on_step_end is not None:
                    view = lat.to(_dt)
                    outs = callback_on_step_end(self, i, tt, {"latents": view} if callback_on_step_end_tensor_inputs else {}) or {}
                    new = outs.pop("latents", view)
                    if new is not view:
                        lat.copy_(new.to(lat.device, lat.dtype))
                    if outs:
                        raise NotImplementedError(f"callback_on_step_end returned {sorted(outs)}: only 'latents' can be replaced between steps")
                if callback is not None and i % (callback_steps or 1) == 0:
                    callback(i, tt, lat.to(_dt))                                                 # :1859-1863 (scheduler order 1)
                return self._interrupt                                                            # :1766-1767: the remaining steps are skipped
        lat = eng(image_dtype=prompt_embeds.dtype, return_latents=True, use_graph=self.use_graph, overlap=self.overlap, on_step=on_step, **call)
        if output_type == "latent":
            return (lat.clone(),)
        out = eng.decode(lat)                                                                    # :1876 + postprocess
        if output_type == "pt":
            return (out,)
        arr = (out.permute(0, 2, 3, 1).float().cpu().numpy() * 255.0).round().astype("uint8")
        if output_type == "np":
            return (arr.astype("float32") / 255.0,)
        if output_type != "pil":
            raise ValueError(f"unknown output_type {output_type!r}")
        import PIL.Image
        return ([PIL.Image.fromarray(a) for a in arr],)                                          # :1885,1894 -> (list[PIL],)
