"""IP-Adapter Resampler (Perceiver) on the HIP kernels -- host mirror of /root/reference/ip_adapter/resampler.py:129-176.

Runs once per pipeline call (src/tryon_pipeline.py:1726), outside the denoising loop.  The 16 latent queries attend to
cat(x, latents) (273 keys): the attention kernel's two key segments under one softmax, so the cat is never materialised.
The reference splits the softmax scale as d^-0.25 on q and on k (:71-72); the kernel applies the same total d^-0.5.
"""
import torch

from . import ops


class HipResampler:
    def __init__(self, state_dict, prefix="", dim=1280, depth=4, dim_head=64, heads=20, num_queries=16, ff_mult=4,
                 dtype=torch.bfloat16, device="cuda"):
        assert dim_head == 64, "HIP attention kernels are specialised for head_dim 64"
        assert num_queries % 16 == 0, "num_queries must be a multiple of 16 (key-order V^T)"
        self.dtype, self.device = dtype, torch.device(device)
        self.dim, self.depth, self.heads, self.nq = dim, depth, heads, num_queries
        self.inner = dim_head * heads
        n = len(prefix)
        self.sd = {k[n:]: v.to(device=self.device, dtype=dtype).contiguous() for k, v in state_dict.items() if k.startswith(prefix)}

    def __call__(self, x):
        """x: [B][n1][embedding_dim] -> [B][num_queries][output_dim]"""
        sd, dt, dev = self.sd, self.dtype, self.device
        B, n1, E = x.shape
        r1 = ops.round16(n1)                                         # token rows padded: key-order V^T needs multiples of 16
        xp = torch.zeros(B, r1, E, dtype=dt, device=dev)
        xp[:, :n1] = x.to(dev, dt)
        inner, nq, D = self.inner, self.nq, self.dim
        lat = sd["latents"].repeat(B, 1, 1).reshape(B * nq, D).contiguous()                       # :166
        xs = ops.linear(xp.reshape(B * r1, E), sd["proj_in.weight"], bias=sd["proj_in.bias"])      # :168
        for l in range(self.depth):
            a, f = f"layers.{l}.0", f"layers.{l}.1"
            xn = ops.layernorm(xs, sd[a + ".norm1.weight"], sd[a + ".norm1.bias"], 1e-5)          # :57
            ln = ops.layernorm(lat, sd[a + ".norm2.weight"], sd[a + ".norm2.bias"], 1e-5)         # :58
            q = ops.linear(ln, sd[a + ".to_q.weight"])                                            # :62
            kx = torch.empty(B * r1, inner, dtype=dt, device=dev)
            vx = torch.empty(B, inner, r1, dtype=dt, device=dev)
            ops.linear(xn, sd[a + ".to_kv.weight"], out=kx, vt=vx, vt_n0=inner, vt_tokens=r1)     # :63-64 (x rows)
            kl = torch.empty(B * nq, inner, dtype=dt, device=dev)
            vl = torch.empty(B, inner, nq, dtype=dt, device=dev)
            ops.linear(ln, sd[a + ".to_kv.weight"], out=kl, vt=vl, vt_n0=inner, vt_tokens=nq)     # (latent rows)
            o = torch.empty(B * nq, inner, dtype=dt, device=dev)
            ops.attention(q, o, [dict(k=kx, vt=vx, nk=n1, ldk=inner, ldvt=r1, k_rows=r1),
                                 dict(k=kl, vt=vl, nk=nq, ldk=inner, ldvt=nq, k_rows=nq)], self.heads,
                          B=B, Nq=nq, ldq=inner, ldo=inner)                                       # :71-76
            lat = ops.linear(o, sd[a + ".to_out.weight"], res=lat)                                 # :78 + residual :172
            h = ops.layernorm(lat, sd[f + ".0.weight"], sd[f + ".0.bias"], 1e-5)                   # FeedForward :13-20
            h = ops.linear(h, sd[f + ".1.weight"], gelu=True)
            lat = ops.linear(h, sd[f + ".3.weight"], res=lat)                                      # :173
        out = ops.linear(lat, sd["proj_out.weight"], bias=sd["proj_out.bias"])                     # :175
        out = ops.layernorm(out, sd["norm_out.weight"], sd["norm_out.bias"], 1e-5)                 # :176
        return out.view(B, nq, -1)
