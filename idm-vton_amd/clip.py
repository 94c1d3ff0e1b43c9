"""CLIP text / vision towers on the HIP kernels -- SURVEY.md 8a row a17 (8f-1): the one-off conditioning encoders the pipeline runs
before the denoising loop (/root/reference/src/tryon_pipeline.py:511-743 `encode_prompt`: CLIP-L `CLIPTextModel` + OpenCLIP-bigG
`CLIPTextModelWithProjection`, penultimate hidden states and the second tower's projected EOS embedding; :460-482 `encode_image`:
CLIP-H `CLIPVisionModelWithProjection`, penultimate hidden states for the Resampler).

The architecture executed is the one of `transformers.models.clip` (the dependency the reference imports; absent from
/root/reference): token+position embeddings -> [pre-LN -> causal MHA -> +res -> pre-LN -> fc1 -> act -> fc2 -> +res] x L ->
final LN -> EOS pooling -> text_projection; vision: patch conv (stride = patch) + class token + positions -> pre_layrnorm -> the
same encoder (no mask) -> post_layernorm(CLS) -> visual_projection.  State-dict keys are transformers' (with or without the
`text_model.` / `vision_model.` prefix, both spellings exist across transformers versions).

Kernels: every linear is `idmvton_gemm_conv` (QKV fused into one GEMM, bias / erf-GELU / quick-GELU / residual in the epilogue),
LayerNorm is `idmvton_layernorm`, attention is `idmvton_attn_small` (causal flag, head_dim 64 or 80).  The embedding gathers and
the patch unfold are index plumbing done with torch on the device.  A few hundred tokens, run once per call: latency, not roofline.
"""
import types

import torch

from . import ops

_ACTS = {"quick_gelu": "quick_gelu", "gelu": "gelu"}


def _strip(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


class _Tower:
    """Shared pre-LN encoder stack."""

    def _init_layers(self, sd, cfg, dtype, device):
        self.dtype, self.device = dtype, torch.device(device)
        self.hidden, self.heads, self.nlayers = cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers
        self.d = self.hidden // self.heads
        self.eps = cfg.layer_norm_eps
        if cfg.hidden_act not in _ACTS:
            raise NotImplementedError(f"CLIP hidden_act {cfg.hidden_act!r}: the HIP GEMM epilogue has gelu (erf) and quick_gelu")
        self.act = _ACTS[cfg.hidden_act]
        if self.hidden % 64 or cfg.intermediate_size % 64 or self.d % 2 or self.d > 128:
            raise ValueError(f"CLIP hidden={self.hidden} intermediate={cfg.intermediate_size} head_dim={self.d}: the HIP GEMM needs "
                             "K % 64 == 0 and the attention kernel an even head_dim <= 128")
        cv = lambda t: t.detach().to(device=self.device, dtype=dtype).contiguous()
        self.layers = []
        for l in range(self.nlayers):
            p = f"encoder.layers.{l}."
            g = lambda n: sd[p + n]
            self.layers.append(dict(
                ln1=(cv(g("layer_norm1.weight")), cv(g("layer_norm1.bias"))), ln2=(cv(g("layer_norm2.weight")), cv(g("layer_norm2.bias"))),
                wqkv=cv(torch.cat([g("self_attn.q_proj.weight"), g("self_attn.k_proj.weight"), g("self_attn.v_proj.weight")])),
                bqkv=cv(torch.cat([g("self_attn.q_proj.bias"), g("self_attn.k_proj.bias"), g("self_attn.v_proj.bias")])),
                wo=cv(g("self_attn.out_proj.weight")), bo=cv(g("self_attn.out_proj.bias")),
                w1=cv(g("mlp.fc1.weight")), b1=cv(g("mlp.fc1.bias")), w2=cv(g("mlp.fc2.weight")), b2=cv(g("mlp.fc2.bias"))))
        self._cv = cv

    def _encode(self, x, B, L, causal, n_layers=None):
        """x: [B*L][hidden] -> list of hidden states (input first), each [B][L][hidden], after each of the first n_layers layers."""
        H, hs = self.hidden, [x.view(B, L, self.hidden)]
        for lay in self.layers[:self.nlayers if n_layers is None else n_layers]:
            h = ops.layernorm(x, *lay["ln1"], self.eps)
            qkv = ops.linear(h, lay["wqkv"], bias=lay["bqkv"])                                   # [B*L][3H]
            o = torch.empty(B * L, H, dtype=self.dtype, device=self.device)
            q3 = qkv.view(B, L, 3 * H)
            ops.attention_small(q3[:, :, :H], q3[:, :, H:2 * H], q3[:, :, 2 * H:], o.view(B, L, H), self.heads, self.d,
                                scale=self.d ** -0.5, causal=causal, B=B, Lq=L, Lk=L, ldq=3 * H, ldk=3 * H, ldv=3 * H, ldo=H)
            x = ops.linear(o, lay["wo"], bias=lay["bo"], res=x)
            h = ops.layernorm(x, *lay["ln2"], self.eps)
            h = ops.linear(h, lay["w1"], bias=lay["b1"], gelu=self.act == "gelu", quick_gelu=self.act == "quick_gelu")
            x = ops.linear(h, lay["w2"], bias=lay["b2"], res=x)
            hs.append(x.view(B, L, H))
        return hs


class HipCLIPText(_Tower):
    """`CLIPTextModel` / `CLIPTextModelWithProjection` forward.  __call__(input_ids, output_hidden_states=True) returns an object
    with .hidden_states (tuple of L+1), .last_hidden_state, .pooler_output and, with a projection, .text_embeds; out[0] is
    text_embeds for the projected tower and last_hidden_state otherwise (what `encode_prompt` indexes, reference :601-606)."""

    def __init__(self, state_dict, config, dtype=torch.bfloat16, device="cuda"):
        sd = _strip(dict(state_dict), "text_model.")
        self._init_layers(sd, config, dtype, device)
        self.tok = self._cv(sd["embeddings.token_embedding.weight"])
        self.pos = self._cv(sd["embeddings.position_embedding.weight"])
        self.lnf = (self._cv(sd["final_layer_norm.weight"]), self._cv(sd["final_layer_norm.bias"]))
        self.proj = self._cv(sd["text_projection.weight"]) if "text_projection.weight" in sd else None
        self.eos = config.eos_token_id

    def __call__(self, input_ids, output_hidden_states=True, penultimate_only=False):
        ids = input_ids.to(self.device)
        B, L = ids.shape
        x = (self.tok[ids] + self.pos[:L]).reshape(B * L, self.hidden).contiguous()
        hs = self._encode(x, B, L, True, self.nlayers - 1 if penultimate_only else None)
        out = types.SimpleNamespace(hidden_states=tuple(hs) if output_hidden_states else None)
        if penultimate_only:                                      # tower 1 of encode_prompt: only hidden_states[-2] is consumed
            out.hidden_states = tuple(hs) + (None,)
            out.first = None
            return out
        last = ops.layernorm(hs[-1].reshape(B * L, self.hidden), *self.lnf, self.eps).view(B, L, self.hidden)
        idx = ids.int().argmax(-1) if self.eos == 2 else (ids.int() == self.eos).int().argmax(-1)
        pooled = last[torch.arange(B, device=self.device), idx]
        out.last_hidden_state, out.pooler_output = last, pooled
        if self.proj is not None:
            out.text_embeds = ops.linear(pooled.contiguous(), self.proj)
            out.first = out.text_embeds
        else:
            out.first = last
        return out


class HipCLIPVision(_Tower):
    """`CLIPVisionModelWithProjection` forward.  __call__(pixel_values) -> .hidden_states (tuple of L+1, [0] = after pre_layrnorm),
    .last_hidden_state, .image_embeds."""

    def __init__(self, state_dict, config, dtype=torch.bfloat16, device="cuda"):
        sd = _strip(dict(state_dict), "vision_model.")
        self._init_layers(sd, config, dtype, device)
        self.patch, self.nch = config.patch_size, config.num_channels
        kp = self.nch * self.patch * self.patch
        self.kpad = (kp + 63) // 64 * 64
        w = torch.zeros(self.hidden, self.kpad, dtype=torch.float32)
        w[:, :kp] = sd["embeddings.patch_embedding.weight"].detach().float().reshape(self.hidden, kp)
        self.wpatch = self._cv(w)
        self.cls = self._cv(sd["embeddings.class_embedding"])
        self.pos = self._cv(sd["embeddings.position_embedding.weight"])
        self.pre = (self._cv(sd["pre_layrnorm.weight"]), self._cv(sd["pre_layrnorm.bias"]))
        self.post = (self._cv(sd["post_layernorm.weight"]), self._cv(sd["post_layernorm.bias"]))
        self.proj = self._cv(sd["visual_projection.weight"]) if "visual_projection.weight" in sd else None

    def __call__(self, pixel_values, output_hidden_states=True, penultimate_only=False):
        px = pixel_values.to(device=self.device, dtype=self.dtype)
        B, Cc, Hh, Ww = px.shape
        P, g = self.patch, Hh // self.patch
        assert Cc == self.nch and Hh == Ww and Hh % P == 0, f"pixel_values {tuple(px.shape)} vs patch {P}"
        kp = Cc * P * P
        cols = torch.zeros(B * g * g, self.kpad, dtype=self.dtype, device=self.device)            # im2col of a stride=patch conv
        cols[:, :kp] = px.view(B, Cc, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, kp)
        L = g * g + 1
        x = torch.empty(B, L, self.hidden, dtype=self.dtype, device=self.device)
        x[:, 0] = self.cls
        x[:, 1:] = ops.linear(cols, self.wpatch).view(B, g * g, self.hidden)
        x += self.pos[:L]
        x = ops.layernorm(x.view(B * L, self.hidden), *self.pre, self.eps)
        hs = self._encode(x, B, L, False, self.nlayers - 1 if penultimate_only else None)
        out = types.SimpleNamespace(hidden_states=tuple(hs) + ((None,) if penultimate_only else ()))
        if penultimate_only:
            return out
        out.last_hidden_state = hs[-1]
        out.pooler_output = ops.layernorm(hs[-1][:, 0].contiguous(), *self.post, self.eps)
        if self.proj is not None:
            out.image_embeds = ops.linear(out.pooler_output, self.proj)
        return out
