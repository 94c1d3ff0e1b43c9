"""`src.unet_hacked_tryon.UNet2DConditionModel` and `src.unet_hacked_garmnet.UNet2DConditionModel` on the HIP engine.

Mirror of the reference classes' call surface:
  * parameters under the diffusers state-dict names (Appendix C), incl. `...attn2.processor.to_k_ip.weight` (the IP
    processors are nn.Modules registered on the attention layers: src/unet_hacked_tryon.py:773-791) and
    `encoder_hid_proj.*` (the Resampler: :474-485);
  * `.config.{in_channels, sample_size, time_cond_proj_dim, addition_time_embed_dim, encoder_hid_dim_type, ...}`, `.dtype`,
    `.device`, `.encoder_hid_proj` (callable), `.add_embedding.linear_1.in_features` (src/tryon_pipeline.py:1049),
    `attn_processors` / `set_attn_processor` (:794-852), `from_pretrained(path, subfolder=, torch_dtype=)`;
  * forward signatures: tryon src/unet_hacked_tryon.py:1006-1022 (incl. `garment_features=`), returns `(sample,)` or an
    object with `.sample`; garmnet src/unet_hacked_garmnet.py:917-932, returns `((sample,), garment_features)` :1281-1284.

The forward runs idm_vton_amd.unet.HipUNet (NHWC, fused kernels).  HipUNet implements the arithmetic of the stock
processors in fused form, so forward() requires the stock processors (AttnProcessor2_0 on attn1, IPAttnProcessor2_0 /
AttnProcessor2_0 on attn2) and raises NotImplementedError for anything else; the processors stay individually callable
(plugin API).  GarmentNet's `sample` return value is None: the reference computes it and every caller discards it
(src/tryon_pipeline.py:1787), and the layers that produce it are dead compute here.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ffi, ops
from ..config import UNetConfig, unet_param_shapes, unet_topology
from ..unet import HipUNet
from .attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
from .modules import Attention, _Holder, params_version
from .resampler import Resampler


class UNet2DConditionOutput(SimpleNamespace):
    """`.sample` holder (diffusers BaseOutput stand-in)."""


def _ensure(node, path):
    for p in path:
        if p not in node._modules:
            node.add_module(p, _Holder())
        node = node._modules[p]
    return node


def _build_tree(root, shapes, dtype, device):
    """Nested modules whose state_dict() keys are exactly `shapes`' names.  2-D weights become nn.Linear (so attributes
    like `.in_features` exist); parameters already created by pre-installed modules (Attention, Resampler) are kept."""
    shapes = list(shapes)
    for name, shp in shapes:                              # pass 1: every 2-D weight becomes an nn.Linear
        parts = name.split(".")
        if len(shp) == 2 and parts[-1] == "weight":
            parent = _ensure(root, parts[:-2])
            if parts[-2] not in parent._modules:
                parent.add_module(parts[-2], nn.Linear(shp[1], shp[0], bias=False, device=device, dtype=dtype))
    for name, shp in shapes:                              # pass 2: everything else (biases land on the Linears above)
        parts = name.split(".")
        node = _ensure(root, parts[:-1])
        if node._parameters.get(parts[-1]) is not None:
            continue
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(*shp, dtype=dtype, device=device), requires_grad=False))


class _UNetBase(nn.Module):
    _mode = "tryon"

    def __init__(self, config: UNetConfig = None, torch_dtype=torch.float32, device="cpu", **overrides):
        super().__init__()
        cfg = config or (UNetConfig.sdxl_tryon() if self._mode == "tryon" else UNetConfig.sdxl_garmnet())
        for k, v in overrides.items():
            if k == "attention_head_dim":             # diffusers' misnomer: number of heads (unet_hacked_tryon.py:366-372)
                k = "num_attention_heads"
            if not hasattr(cfg, k):
                raise TypeError(f"unknown UNet config field {k!r}")
            setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        if cfg.mode != self._mode:
            raise ValueError(f"{type(self).__module__} expects a {self._mode!r} config, got {cfg.mode!r}")
        self.cfg = cfg
        self.config = SimpleNamespace(
            in_channels=cfg.in_channels, out_channels=cfg.out_channels, sample_size=cfg.sample_size,
            block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
            attention_head_dim=cfg.num_attention_heads, cross_attention_dim=cfg.cross_attention_dim,
            transformer_layers_per_block=cfg.transformer_layers_per_block, time_cond_proj_dim=cfg.time_cond_proj_dim,
            addition_embed_type=cfg.addition_embed_type, addition_time_embed_dim=cfg.addition_time_embed_dim,
            projection_class_embeddings_input_dim=cfg.projection_class_embeddings_input_dim,
            encoder_hid_dim_type=cfg.encoder_hid_dim_type, encoder_hid_dim=cfg.encoder_hid_dim,
            norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps, down_block_types=cfg.down_block_types,
            up_block_types=cfg.up_block_types)
        with torch.device(device):
            if cfg.encoder_hid_dim_type == "ip_image_proj":                     # unet_hacked_tryon.py:474-485
                r = cfg.resampler
                self.encoder_hid_proj = Resampler(dim=r["dim"], depth=r["depth"], dim_head=r["dim_head"], heads=r["heads"],
                                                  num_queries=r["num_queries"], embedding_dim=cfg.encoder_hid_dim,
                                                  output_dim=cfg.cross_attention_dim, ff_mult=r["ff_mult"]).to(torch_dtype)
            else:
                self.encoder_hid_proj = None
            topo = unet_topology(cfg)
            sites = [(f"down_blocks.{i}.attentions.{j}", b) for i, b in enumerate(topo["down"]) if b["attn"] for j in range(len(b["resnets"]))]
            sites += [("mid_block.attentions.0", topo["mid"])]
            sites += [(f"up_blocks.{i}.attentions.{j}", b) for i, b in enumerate(topo["up"]) if b["attn"] for j in range(len(b["resnets"]))]
            for p, b in sites:
                for k in range(b["n_tf"]):
                    blk = _ensure(self, f"{p}.transformer_blocks.{k}".split("."))
                    a1 = Attention(b["ch"], None, b["heads"], b["ch"] // b["heads"], processor=AttnProcessor2_0())
                    if self._mode == "tryon":                                    # unet_hacked_tryon.py:773-791
                        proc = IPAttnProcessor2_0(hidden_size=b["ch"], cross_attention_dim=cfg.cross_attention_dim,
                                                  num_tokens=cfg.ip_num_tokens)
                    else:
                        proc = AttnProcessor2_0()
                    a2 = Attention(b["ch"], cfg.cross_attention_dim, b["heads"], b["ch"] // b["heads"], processor=proc)
                    blk.add_module("attn1", a1.to(torch_dtype))
                    blk.add_module("attn2", a2.to(torch_dtype))
        _build_tree(self, unet_param_shapes(cfg), torch_dtype, device)
        for p in self.parameters():
            p.requires_grad_(False)
        self._hip, self._hip_key = None, None

    # ------------------------------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, torch_dtype=torch.float32, **kw):
        """Local directory in the diffusers layout: <path>/<subfolder>/config.json + diffusion_pytorch_model.safetensors.
        (Hub ids need network access and are not supported: inference.py is run with --pretrained_model_name_or_path <dir>.)"""
        from safetensors.torch import load_file
        d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        cj = os.path.join(d, "config.json")
        if not os.path.isfile(cj):
            raise EnvironmentError(f"{cj} not found: from_pretrained needs a local diffusers-layout directory")
        raw = json.load(open(cj))
        fields = {k: v for k, v in raw.items() if not k.startswith("_") and (k == "attention_head_dim" or hasattr(UNetConfig, k) or k in UNetConfig.__dataclass_fields__)}
        fields.pop("mode", None)
        m = cls(torch_dtype=torch_dtype, device="meta", **fields)
        sd = load_file(os.path.join(d, "diffusion_pytorch_model.safetensors"))
        m.load_state_dict({k: v.to(torch_dtype) for k, v in sd.items()}, strict=True, assign=True)
        return m

    def save_pretrained(self, save_directory):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        c = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()}
        # not diffusers config keys: the Resampler geometry (train_xl.py:341-352 hard-codes it; absent => those defaults)
        c["resampler"], c["ip_num_tokens"] = dict(self.cfg.resampler), self.cfg.ip_num_tokens
        c["_class_name"] = "UNet2DConditionModel"
        json.dump(c, open(os.path.join(save_directory, "config.json"), "w"), indent=1)
        save_file({k: v.contiguous().cpu() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))

    # ------------------------------------------------------------------------------------------ attributes
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def attn_processors(self):
        """{'<path>.processor': processor} for every attention layer (reference :794-817)."""
        return {f"{n}.processor": m.processor for n, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor, _remove_lora=False):
        """One processor for all layers, or a dict keyed like `attn_processors` (reference :819-852)."""
        layers = {f"{n}.processor": m for n, m in self.named_modules() if isinstance(m, Attention)}
        if isinstance(processor, dict):
            if len(processor) != len(layers):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not "
                                 f"match the number of attention layers: {len(layers)}. Please make sure to pass "
                                 f"{len(layers)} processor classes.")
            for k, m in layers.items():
                m.set_processor(processor[k])
        else:
            for m in layers.values():
                m.set_processor(processor)
        self._hip_key = None

    def set_default_attn_processor(self):
        """Reference :854-867: only when every processor is a stock cross-attention processor; anything else -- TryonNet's IPAttnProcessor2_0
        layers, which are in neither of diffusers' processor sets -- raises ValueError there, and here (silently replacing them would drop
        the to_k_ip / to_v_ip parameters from the state dict and leave a model whose forward cannot run)."""
        procs = self.attn_processors
        if not all(type(p) is AttnProcessor2_0 for p in procs.values()):
            raise ValueError(f"Cannot call `set_default_attn_processor` when attention processors are of type {next(iter(procs.values()))}")
        self.set_attn_processor(AttnProcessor2_0(), _remove_lora=True)

    def set_attention_slice(self, slice_size):
        """Reference :869-932 (sliced attention to save memory).  No-op by construction: the HIP attention kernel is blockwise (online softmax
        over 64-key tiles, csrc/attention.hip) and never materialises a [queries x keys] score matrix, so there is nothing to slice; the
        argument is validated like the reference's (one entry per sliceable layer, none larger than the layer's head count) and recorded."""
        dims = [m.heads for _, m in self.named_modules() if isinstance(m, Attention)]            # sliceable_head_dim of every attention layer
        if slice_size == "auto":
            sizes = [d // 2 for d in dims]
        elif slice_size == "max":
            sizes = len(dims) * [1]
        elif isinstance(slice_size, (list, tuple)):
            sizes = list(slice_size)
        elif isinstance(slice_size, int) or slice_size is None:
            sizes = len(dims) * [slice_size]
        else:
            raise ValueError(f"slice_size {slice_size!r} must be 'auto', 'max', an int or a list of ints")
        if len(sizes) != len(dims):
            raise ValueError(f"You have provided {len(sizes)}, but {self.config} has {len(dims)} different"
                             f" attention layers. Make sure to match `len(slice_size)` to be {len(dims)}.")
        for size, dim in zip(sizes, dims):
            if size is not None and size > dim:
                raise ValueError(f"size {size} has to be smaller or equal to {dim}.")
        self.attention_slice = slice_size

    def fuse_qkv_projections(self):
        """Reference :970-991.  The HIP engine ALWAYS runs attn1's q|k|v as one GEMM and attn2's k|v as one (prepared weight layouts,
        idm_vton_amd/unet.py:_prep), whatever this flag says: same arithmetic per output channel, so the call only checks what the reference
        checks and records the request."""
        for proc in self.attn_processors.values():
            if "Added" in proc.__class__.__name__:
                raise ValueError("`fuse_qkv_projections()` is not supported for models having added KV projections.")
        self.original_attn_processors = self.attn_processors

    def unfuse_qkv_projections(self):
        """Reference :993-1004."""
        if getattr(self, "original_attn_processors", None) is not None:
            self.set_attn_processor(self.original_attn_processors)

    def enable_freeu(self, s1, s2, b1, b2):
        """Reference :938-960 rescales skip / backbone features in the up blocks -- different arithmetic that no IDM-VTON script enables."""
        raise NotImplementedError("FreeU is not used on the try-on path and not supported by the HIP forward")

    def disable_freeu(self):
        return None

    # ------------------------------------------------------------------------------------------ engine
    def _check_fusable(self):
        scale = None
        for n, m in self.named_modules():
            if not isinstance(m, Attention):
                continue
            want = IPAttnProcessor2_0 if (self._mode == "tryon" and n.endswith("attn2")) else AttnProcessor2_0
            if type(m.processor) is not want:
                raise NotImplementedError(f"{n}.processor is {type(m.processor).__name__}; the fused HIP forward implements "
                                          f"the stock {want.__name__} only")
            if want is IPAttnProcessor2_0:
                if scale is not None and m.processor.scale != scale:
                    raise NotImplementedError("per-layer IP scales differ; the fused HIP forward takes one scale")
                scale = m.processor.scale
        return 1.0 if scale is None else float(scale)

    def hip_engine(self):
        """The HipUNet executing this module's weights (prepared GEMM layouts cached until a parameter changes)."""
        ffi.lib()
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("UNet2DConditionModel.forward runs on the GPU only (HIP kernels); call .to('cuda') first")
        if p0.dtype not in (torch.float16, torch.bfloat16):
            raise TypeError(f"HIP kernels compute in float16/bfloat16 storage, model dtype is {p0.dtype}")
        ip_scale = self._check_fusable()
        key = (params_version(self), ip_scale)
        if key != self._hip_key:
            # engine options a drop-in user selects without touching the reference script: IDMVTON_ATTN_FP8=1 (BASELINE configs[4]: self-
            # attention on e4m3 operands), IDMVTON_STREAM_F32=1 (fp32 residual stream inside each Transformer2DModel)
            import os
            flag = lambda k: os.environ.get(k, "0") == "1"
            self._hip = HipUNet(self.cfg, self.state_dict(), p0.dtype, p0.device, stream_f32=flag("IDMVTON_STREAM_F32"),
                                attn_fp8=flag("IDMVTON_ATTN_FP8"))
            self._hip.ip_scale = ip_scale
            self._hip_key = key
        return self._hip

    def _unsupported(self, **kw):
        for k, v in kw.items():
            if v is not None:
                raise NotImplementedError(f"`{k}` is not used on the try-on path and not supported by the HIP forward")

    def _run(self, sample, timestep, text, added_cond_kwargs, ip_tokens, garment_features):
        eng = self.hip_engine()
        dev, dt = eng.device, eng.dtype
        B, _, h, w = sample.shape
        t = timestep.reshape(-1)[:1].tolist() if torch.is_tensor(timestep) else [timestep]   # timesteps.expand(B) :1131
        temb = eng.time_embeddings(t, B, added_cond_kwargs)[0]
        ctx = eng.encode_context(text.to(dev), None if ip_tokens is None else ip_tokens.to(dev))
        x = ops.to_nhwc(sample.to(dev, torch.float32).contiguous(), dt, cpad=eng.cin_pad)
        feats = None
        if garment_features is not None:
            feats = [f.to(dev, dt).contiguous() for f in garment_features]
        return eng.forward(x, temb, ctx, B, h, w, garment_feats=feats), (B, h, w)


class TryonUNet2DConditionModel(_UNetBase):
    """Drop-in for src.unet_hacked_tryon.UNet2DConditionModel (reference :204)."""
    _mode = "tryon"

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                encoder_attention_mask=None, return_dict=True, garment_features=None):
        self._unsupported(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                          down_block_additional_residuals=down_block_additional_residuals,
                          mid_block_additional_residual=mid_block_additional_residual,
                          down_intrablock_additional_residuals=down_intrablock_additional_residuals,
                          encoder_attention_mask=encoder_attention_mask)
        cfg = self.cfg
        added_cond_kwargs = added_cond_kwargs or {}
        if cfg.addition_embed_type == "text_time":                                       # reference :1174-1186
            if "text_embeds" not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `addition_embed_type` set to 'text_time' which "
                                 "requires the keyword argument `text_embeds` to be passed in `added_cond_kwargs`")
            if "time_ids" not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `addition_embed_type` set to 'text_time' which "
                                 "requires the keyword argument `time_ids` to be passed in `added_cond_kwargs`")
        ip = None
        if cfg.encoder_hid_dim_type == "ip_image_proj":                                  # reference :1234-1242
            if "image_embeds" not in added_cond_kwargs:
                raise ValueError(f"{self.__class__} has the config param `encoder_hid_dim_type` set to 'ip_image_proj' "
                                 "which requires the keyword argument `image_embeds` to be passed in  `added_conditions`")
            ip = added_cond_kwargs["image_embeds"]
        if garment_features is None:
            raise ValueError("TryonNet needs `garment_features` (the 70 GarmentNet norm1 outputs: tryon_pipeline.py:1787-1808)")
        n_feat = self.hip_engine().num_features()
        if len(garment_features) != n_feat:
            raise ValueError(f"expected {n_feat} garment features, got {len(garment_features)}")
        (eps, _), (B, h, w) = self._run(sample, timestep, encoder_hidden_states, added_cond_kwargs, ip, garment_features)
        out = ops.to_nchw(eps, cfg.out_channels, (h, w)).to(sample.dtype)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)


class GarmentUNet2DConditionModel(_UNetBase):
    """Drop-in for src.unet_hacked_garmnet.UNet2DConditionModel (reference :80)."""
    _mode = "garmnet"

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs=None, added_cond_kwargs=None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, down_intrablock_additional_residuals=None,
                encoder_attention_mask=None, return_dict=True):
        self._unsupported(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                          down_block_additional_residuals=down_block_additional_residuals,
                          mid_block_additional_residual=mid_block_additional_residual,
                          down_intrablock_additional_residuals=down_intrablock_additional_residuals,
                          encoder_attention_mask=encoder_attention_mask)
        (_, feats), (_, h, w) = self._run(sample, timestep, encoder_hidden_states, None, None, None)
        # the engine keeps round16(H*W) token rows per image; the reference's features have exactly H*W (views, no copy)
        feats = [f if f.shape[1] == n else f[:, :n] for f, n in zip(feats, self.hip_engine().feature_tokens(h, w))]
        if not return_dict:
            return (None,), feats
        return UNet2DConditionOutput(sample=None), feats
