"""diffusers 0.25.0 `Attention` (the module the reference's BasicTransformerBlock builds, src/attentionhacked_tryon.py:201-240)
and its default `AttnProcessor2_0` (GarmentNet keeps it: SURVEY.md 8a row a9).  Written from the published semantics of that
release; implemented: the configuration the SDXL UNets use, plus the VAE mid-block form the reference's own `UNetMidBlock2D`
constructs (src/unet_block_hacked_tryon.py:585-597: `norm_num_groups` -> GroupNorm(query_dim, eps, affine), bias=True,
residual_connection=True, rescale_output_factor, upcast_softmax, _from_deprecated_attn_block).  Not implemented: spatial_norm /
added_kv / norm_cross."""
from typing import Union

import torch
import torch.nn.functional as F
from torch import nn

from ..utils import USE_PEFT_BACKEND
from .lora import LoRACompatibleLinear


class AttnProcessor2_0:
    def __init__(self):
        if not hasattr(F, "scaled_dot_product_attention"):
            raise ImportError("AttnProcessor2_0 requires PyTorch 2.0, to use it, please upgrade PyTorch to 2.0.")

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale: float = 1.0):
        residual = hidden_states
        if attn.spatial_norm is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size, sequence_length, _ = (hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape)
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
            attention_mask = attention_mask.view(batch_size, attn.heads, -1, attention_mask.shape[-1])
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        args = () if USE_PEFT_BACKEND else (scale,)
        query = attn.to_q(hidden_states, *args)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif attn.norm_cross:
            encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
        key = attn.to_k(encoder_hidden_states, *args)
        value = attn.to_v(encoder_hidden_states, *args)
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states, *args)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        hidden_states = hidden_states / attn.rescale_output_factor
        return hidden_states


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None, cross_attention_norm_num_groups=32,
                 added_kv_proj_dim=None, norm_num_groups=None, spatial_norm_dim=None, out_bias=True, scale_qk=True,
                 only_cross_attention=False, eps=1e-5, rescale_output_factor=1.0, residual_connection=False,
                 _from_deprecated_attn_block=False, processor=None):
        super().__init__()
        self.inner_dim = dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.dropout = dropout
        self._from_deprecated_attn_block = _from_deprecated_attn_block
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if self.scale_qk else 1.0
        self.heads = heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        if spatial_norm_dim is not None or cross_attention_norm is not None or added_kv_proj_dim is not None:
            raise NotImplementedError("not constructed on the IDM-VTON UNet / VAE path")
        # diffusers 0.25 Attention.__init__: GroupNorm over the query channels, applied by the processor on [b, c, tokens]
        self.group_norm = (nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.spatial_norm = None
        self.norm_cross = None
        linear_cls = nn.Linear if USE_PEFT_BACKEND else LoRACompatibleLinear
        self.to_q = linear_cls(query_dim, self.inner_dim, bias=bias)
        if not self.only_cross_attention:
            self.to_k = linear_cls(self.cross_attention_dim, self.inner_dim, bias=bias)
            self.to_v = linear_cls(self.cross_attention_dim, self.inner_dim, bias=bias)
        else:
            self.to_k = None
            self.to_v = None
        self.to_out = nn.ModuleList([])
        self.to_out.append(linear_cls(self.inner_dim, query_dim, bias=out_bias))
        self.to_out.append(nn.Dropout(dropout))
        if processor is None:
            processor = AttnProcessor2_0() if hasattr(F, "scaled_dot_product_attention") and self.scale_qk else AttnProcessor()
        self.set_processor(processor)

    def set_processor(self, processor, _remove_lora=False):
        if hasattr(self, "processor") and isinstance(self.processor, torch.nn.Module) and not isinstance(processor, torch.nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask,
                              **cross_attention_kwargs)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        raise NotImplementedError("attention masks are not used on the IDM-VTON path")


class AttnProcessor:
    def __init__(self, *a, **k):
        raise NotImplementedError("pre-PyTorch-2 processor: not on the IDM-VTON path")


class AttnAddedKVProcessor(AttnProcessor):
    pass


class AttnAddedKVProcessor2_0(AttnProcessor):
    pass


ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor, AttnAddedKVProcessor2_0)
CROSS_ATTENTION_PROCESSORS = (AttnProcessor, AttnProcessor2_0)
AttentionProcessor = Union[AttnProcessor, AttnProcessor2_0, AttnAddedKVProcessor, AttnAddedKVProcessor2_0]


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    from .. import _placeholder
    return _placeholder(name)
