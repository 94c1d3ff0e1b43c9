"""Parameter-holding nn.Module pieces shared by the boundary classes.

`Attention` carries the attributes the reference's processors read from diffusers' Attention
(ip_adapter/attention_processor.py:213-276: heads, to_q/to_k/to_v, to_out[0]/[1], spatial_norm, group_norm,
norm_cross, residual_connection, rescale_output_factor) with diffusers' parameter names, so state-dict keys match a
real checkpoint (`...attn2.to_k.weight`, `...attn2.processor.to_k_ip.weight`: SURVEY.md Appendix C).
"""
import torch
import torch.nn as nn


class Attention(nn.Module):
    """diffusers-Attention-like container as built at src/attentionhacked_tryon.py:201-210,231-240: q/k/v without
    bias, to_out[0] with bias, dropout 0, softmax scale dim_head^-0.5.  forward() delegates to `self.processor`."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False, out_bias=True,
                 processor=None):
        super().__init__()
        self.inner_dim = heads * dim_head
        self.query_dim = query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])
        if processor is None:
            from .attention_processor import AttnProcessor2_0
            processor = AttnProcessor2_0()
        self.set_processor(processor)

    def set_processor(self, processor):
        # processors are nn.Modules registered as `processor` (=> keys `...attn2.processor.to_k_ip.weight`)
        if "processor" in self._modules:
            del self._modules["processor"]
        self.processor = processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        raise NotImplementedError("attention masks are not supported by the HIP attention kernels "
                                  "(the try-on path never passes one: tryon_pipeline.py:1799-1808)")

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class _Holder(nn.Module):
    """Structure-only node of a parameter tree (no forward)."""


def build_param_tree(root, shapes, dtype=torch.float32, device="meta"):
    """Register every (dotted name, shape) of `shapes` under `root` as nested modules/parameters so that
    root.state_dict() has exactly those keys (diffusers naming).  Parameters are created on `device` ("meta" by default:
    the real storage arrives with load_state_dict(assign=True) / from_arena)."""
    for name, shp in shapes:
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Holder())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(*shp, dtype=dtype, device=device),
                                                        requires_grad=False))
    return root


def params_version(module):
    """Cheap fingerprint of a module's parameter storage: changes when weights are re-assigned, moved or edited in
    place (tensor._version), so prepared HIP weight layouts can be cached safely."""
    return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in module.parameters())
