"""Model hyper-parameters (product side) and the parameter inventory (names/shapes) of the two UNets, the Resampler and
the VAE, in diffusers state-dict naming so real checkpoints load unchanged.

Values for SDXL come from SURVEY.md A.1 (the checkpoints' config.json files are not in the reference repo; the class
defaults at /root/reference/src/unet_hacked_tryon.py:301-356 are overridden by them; train_xl.py:323-373 documents the
TryonNet surgery: 13 input channels, ip_image_proj, text_time).
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple


@dataclass
class UNetConfig:
    mode: str = "tryon"                      # "tryon" | "garmnet"
    in_channels: int = 13
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 2, 10)
    num_attention_heads: Tuple[int, ...] = (5, 10, 20)     # diffusers' `attention_head_dim` (unet_hacked_tryon.py:366-372)
    cross_attention_dim: int = 2048
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_embed_type: Optional[str] = "text_time"
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    encoder_hid_dim_type: Optional[str] = "ip_image_proj"
    encoder_hid_dim: int = 1280
    ip_num_tokens: int = 16
    sample_size: int = 128
    time_cond_proj_dim: Optional[int] = None
    resampler: dict = field(default_factory=lambda: dict(dim=1280, depth=4, dim_head=64, heads=20, num_queries=16,
                                                         ff_mult=4))

    @staticmethod
    def sdxl_tryon():
        return UNetConfig()

    @staticmethod
    def sdxl_garmnet():
        return UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None)

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.13025
    force_upcast: bool = True


# ---------------------------------------------------------------------------------------------------- topology walk
def unet_topology(cfg: UNetConfig):
    """The block structure both the parameter inventory and the executor walk (mirrors the ctor at
    src/unet_hacked_tryon.py:589-744).  Returns dict(down=[...], mid=..., up=[...]) of plain dicts."""
    boc = cfg.block_out_channels
    nb = len(boc)
    down, out_ch = [], boc[0]
    for i, t in enumerate(cfg.down_block_types):
        in_ch, out_ch = out_ch, boc[i]
        down.append(dict(type=t, resnets=[(in_ch if j == 0 else out_ch, out_ch) for j in range(cfg.layers_per_block)],
                         attn=t != "DownBlock2D", n_tf=cfg.transformer_layers_per_block[i],
                         heads=cfg.num_attention_heads[i], ch=out_ch, down=i != nb - 1))
    mid = dict(ch=boc[-1], n_tf=cfg.transformer_layers_per_block[-1], heads=cfg.num_attention_heads[-1])
    rboc = list(reversed(boc))
    rheads = list(reversed(cfg.num_attention_heads))
    rtf = list(reversed(cfg.transformer_layers_per_block))
    up, out_ch = [], rboc[0]
    for i, t in enumerate(cfg.up_block_types):
        prev, out_ch = out_ch, rboc[i]
        in_ch = rboc[min(i + 1, nb - 1)]
        n = cfg.layers_per_block + 1
        res = []
        for j in range(n):
            skip = in_ch if j == n - 1 else out_ch
            rin = prev if j == 0 else out_ch
            res.append((rin, skip, out_ch))
        up.append(dict(type=t, resnets=res, attn=t != "UpBlock2D", n_tf=rtf[i], heads=rheads[i], ch=out_ch,
                       up=i != nb - 1))
    return dict(down=down, mid=mid, up=up)


def _resnet_params(p, cin, cout, temb):
    yield f"{p}.norm1.weight", (cin,)
    yield f"{p}.norm1.bias", (cin,)
    yield f"{p}.conv1.weight", (cout, cin, 3, 3)
    yield f"{p}.conv1.bias", (cout,)
    if temb is not None:
        yield f"{p}.time_emb_proj.weight", (cout, temb)
        yield f"{p}.time_emb_proj.bias", (cout,)
    yield f"{p}.norm2.weight", (cout,)
    yield f"{p}.norm2.bias", (cout,)
    yield f"{p}.conv2.weight", (cout, cout, 3, 3)
    yield f"{p}.conv2.bias", (cout,)
    if cin != cout:
        yield f"{p}.conv_shortcut.weight", (cout, cin, 1, 1)
        yield f"{p}.conv_shortcut.bias", (cout,)


def _transformer_params(p, ch, n_tf, cfg):
    xd = cfg.cross_attention_dim
    for n in ("norm.weight", "norm.bias", "proj_in.bias", "proj_out.bias"):
        yield f"{p}.{n}", (ch,)
    yield f"{p}.proj_in.weight", (ch, ch)
    yield f"{p}.proj_out.weight", (ch, ch)
    for k in range(n_tf):
        b = f"{p}.transformer_blocks.{k}"
        for n in ("norm1", "norm2", "norm3"):
            yield f"{b}.{n}.weight", (ch,)
            yield f"{b}.{n}.bias", (ch,)
        for a, kd in (("attn1", ch), ("attn2", xd)):
            yield f"{b}.{a}.to_q.weight", (ch, ch)
            yield f"{b}.{a}.to_k.weight", (ch, kd)
            yield f"{b}.{a}.to_v.weight", (ch, kd)
            yield f"{b}.{a}.to_out.0.weight", (ch, ch)
            yield f"{b}.{a}.to_out.0.bias", (ch,)
        if cfg.mode == "tryon":
            yield f"{b}.attn2.processor.to_k_ip.weight", (ch, xd)
            yield f"{b}.attn2.processor.to_v_ip.weight", (ch, xd)
        yield f"{b}.ff.net.0.proj.weight", (8 * ch, ch)
        yield f"{b}.ff.net.0.proj.bias", (8 * ch,)
        yield f"{b}.ff.net.2.weight", (ch, 4 * ch)
        yield f"{b}.ff.net.2.bias", (ch,)


def resampler_param_shapes(prefix, embedding_dim, output_dim, dim, depth, dim_head, heads, num_queries, ff_mult):
    inner = dim_head * heads
    yield f"{prefix}latents", (1, num_queries, dim)
    yield f"{prefix}proj_in.weight", (dim, embedding_dim)
    yield f"{prefix}proj_in.bias", (dim,)
    yield f"{prefix}proj_out.weight", (output_dim, dim)
    yield f"{prefix}proj_out.bias", (output_dim,)
    yield f"{prefix}norm_out.weight", (output_dim,)
    yield f"{prefix}norm_out.bias", (output_dim,)
    for l in range(depth):
        a = f"{prefix}layers.{l}.0"
        for n in ("norm1", "norm2"):
            yield f"{a}.{n}.weight", (dim,)
            yield f"{a}.{n}.bias", (dim,)
        yield f"{a}.to_q.weight", (inner, dim)
        yield f"{a}.to_kv.weight", (2 * inner, dim)
        yield f"{a}.to_out.weight", (dim, inner)
        f = f"{prefix}layers.{l}.1"
        yield f"{f}.0.weight", (dim,)
        yield f"{f}.0.bias", (dim,)
        yield f"{f}.1.weight", (dim * ff_mult, dim)
        yield f"{f}.3.weight", (dim, dim * ff_mult)


def unet_param_shapes(cfg: UNetConfig):
    """(name, shape) for every parameter, in module order -- SURVEY.md Appendix C."""
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    topo = unet_topology(cfg)
    yield "conv_in.weight", (boc[0], cfg.in_channels, 3, 3)
    yield "conv_in.bias", (boc[0],)
    yield "time_embedding.linear_1.weight", (temb, boc[0])
    yield "time_embedding.linear_1.bias", (temb,)
    yield "time_embedding.linear_2.weight", (temb, temb)
    yield "time_embedding.linear_2.bias", (temb,)
    if cfg.encoder_hid_dim_type == "ip_image_proj":
        yield from resampler_param_shapes("encoder_hid_proj.", cfg.encoder_hid_dim, cfg.cross_attention_dim, **cfg.resampler)
    if cfg.addition_embed_type == "text_time":
        yield "add_embedding.linear_1.weight", (temb, cfg.projection_class_embeddings_input_dim)
        yield "add_embedding.linear_1.bias", (temb,)
        yield "add_embedding.linear_2.weight", (temb, temb)
        yield "add_embedding.linear_2.bias", (temb,)
    for i, blk in enumerate(topo["down"]):
        for j, (ci, co) in enumerate(blk["resnets"]):
            yield from _resnet_params(f"down_blocks.{i}.resnets.{j}", ci, co, temb)
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                yield from _transformer_params(f"down_blocks.{i}.attentions.{j}", blk["ch"], blk["n_tf"], cfg)
        if blk["down"]:
            yield f"down_blocks.{i}.downsamplers.0.conv.weight", (blk["ch"], blk["ch"], 3, 3)
            yield f"down_blocks.{i}.downsamplers.0.conv.bias", (blk["ch"],)
    m = topo["mid"]
    yield from _resnet_params("mid_block.resnets.0", m["ch"], m["ch"], temb)
    yield from _transformer_params("mid_block.attentions.0", m["ch"], m["n_tf"], cfg)
    yield from _resnet_params("mid_block.resnets.1", m["ch"], m["ch"], temb)
    for i, blk in enumerate(topo["up"]):
        for j, (rin, skip, co) in enumerate(blk["resnets"]):
            yield from _resnet_params(f"up_blocks.{i}.resnets.{j}", rin + skip, co, temb)
        if blk["attn"]:
            for j in range(len(blk["resnets"])):
                yield from _transformer_params(f"up_blocks.{i}.attentions.{j}", blk["ch"], blk["n_tf"], cfg)
        if blk["up"]:
            yield f"up_blocks.{i}.upsamplers.0.conv.weight", (blk["ch"], blk["ch"], 3, 3)
            yield f"up_blocks.{i}.upsamplers.0.conv.bias", (blk["ch"],)
    yield "conv_norm_out.weight", (boc[0],)
    yield "conv_norm_out.bias", (boc[0],)
    yield "conv_out.weight", (cfg.out_channels, boc[0], 3, 3)
    yield "conv_out.bias", (cfg.out_channels,)


def vae_param_shapes(cfg: VAEConfig):
    boc, L = cfg.block_out_channels, cfg.layers_per_block

    def attn(p, ch):
        yield f"{p}.group_norm.weight", (ch,)
        yield f"{p}.group_norm.bias", (ch,)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            yield f"{p}.{n}.weight", (ch, ch)
            yield f"{p}.{n}.bias", (ch,)

    def mid(p, ch):
        yield from _resnet_params(f"{p}.resnets.0", ch, ch, None)
        yield from attn(f"{p}.attentions.0", ch)
        yield from _resnet_params(f"{p}.resnets.1", ch, ch, None)

    yield "encoder.conv_in.weight", (boc[0], cfg.in_channels, 3, 3)
    yield "encoder.conv_in.bias", (boc[0],)
    out = boc[0]
    for i, c in enumerate(boc):
        cin, out = out, c
        for j in range(L):
            yield from _resnet_params(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, None)
        if i != len(boc) - 1:
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (out,)
    yield from mid("encoder.mid_block", boc[-1])
    yield "encoder.conv_norm_out.weight", (boc[-1],)
    yield "encoder.conv_norm_out.bias", (boc[-1],)
    yield "encoder.conv_out.weight", (2 * cfg.latent_channels, boc[-1], 3, 3)
    yield "encoder.conv_out.bias", (2 * cfg.latent_channels,)
    yield "decoder.conv_in.weight", (boc[-1], cfg.latent_channels, 3, 3)
    yield "decoder.conv_in.bias", (boc[-1],)
    yield from mid("decoder.mid_block", boc[-1])
    rboc = list(reversed(boc))
    out = rboc[0]
    for i, c in enumerate(rboc):
        cin, out = out, c
        for j in range(L + 1):
            yield from _resnet_params(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, None)
        if i != len(boc) - 1:
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (out, out, 3, 3)
            yield f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (out,)
    yield "decoder.conv_norm_out.weight", (boc[0],)
    yield "decoder.conv_norm_out.bias", (boc[0],)
    yield "decoder.conv_out.weight", (cfg.out_channels, boc[0], 3, 3)
    yield "decoder.conv_out.bias", (cfg.out_channels,)
    yield "quant_conv.weight", (2 * cfg.latent_channels, 2 * cfg.latent_channels, 1, 1)
    yield "quant_conv.bias", (2 * cfg.latent_channels,)
    yield "post_quant_conv.weight", (cfg.latent_channels, cfg.latent_channels, 1, 1)
    yield "post_quant_conv.bias", (cfg.latent_channels,)


def random_state_dict(shapes, seed, dtype, device, std=0.02):
    """Seeded random-init weights (there are no trained weights anywhere: SURVEY.md 0.2): N(0, std) for matrices / conv
    kernels, N(0, std) biases, norm gains 1 + N(0, std), so activations stay bounded through 70 blocks and every
    parameter influences the output.  Generated per tensor on the CPU generator so oracle and HIP see identical bytes."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in shapes:
        t = torch.randn(*shape, generator=g) * std
        if name.endswith(".weight") and len(shape) == 1:       # every 1-D weight is a LayerNorm / GroupNorm gain
            t = t + 1.0
        sd[name] = t.to(dtype).to(device)
    return sd


def fill_random_(views, seed, std=0.02):
    """In-place seeded init of an arena's parameter views ON THEIR DEVICE (bench / multi-GPU path: weights are generated
    by rank 0 directly in HBM and broadcast; same distribution as random_state_dict, different byte stream)."""
    import torch
    dev = next(iter(views.values())).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, v in views.items():
        v.normal_(0.0, std, generator=g)
        if name.endswith(".weight") and v.dim() == 1:
            v.add_(1.0)
    return views
