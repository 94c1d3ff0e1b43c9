"""SDXL AutoencoderKL encode/decode on the HIP kernels (NHWC).

Host-side mirror of the diffusers AutoencoderKL the reference calls at src/tryon_pipeline.py:924,1646,1876 (block
structure: src/unet_block_hacked_tryon.py:505-627,1292-1349,2511-2568).  Same state-dict keys.
The single-head 512-wide mid-block attention runs as gemm_conv (QK^T) -> softmax_rows -> gemm_conv (PV).
"""
import torch

from . import ops
from .config import VAEConfig
from .unet import _Conv, _pad64
from .weights import pad_k


class HipVAE:
    def __init__(self, cfg: VAEConfig, state_dict, dtype=torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        sd = {k: v.to(device=self.device, dtype=dtype) for k, v in state_dict.items()}
        self.sd = sd
        boc, L = cfg.block_out_channels, cfg.layers_per_block
        self.convs = {}
        self._gn_stats = torch.empty(ops.GN_STATS_DOUBLES, dtype=torch.float64, device=self.device)

        def res(p, cin, cout):
            self.convs[p + ".conv1"] = _Conv(sd, p + ".conv1")
            self.convs[p + ".conv2"] = _Conv(sd, p + ".conv2", shortcut=(p + ".conv_shortcut") if cin != cout else None)

        self.enc_plan, out = [], boc[0]
        self.convs["encoder.conv_in"] = _Conv(sd, "encoder.conv_in", cin_pad=_pad64(cfg.in_channels))
        for i, c in enumerate(boc):
            cin, out = out, c
            rs = []
            for j in range(L):
                p = f"encoder.down_blocks.{i}.resnets.{j}"
                res(p, cin if j == 0 else out, out)
                rs.append((p, cin if j == 0 else out, out))
            down = None
            if i != len(boc) - 1:
                down = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                self.convs[down] = _Conv(sd, down)
            self.enc_plan.append((rs, down))
        for side in ("encoder", "decoder"):
            res(f"{side}.mid_block.resnets.0", boc[-1], boc[-1])
            res(f"{side}.mid_block.resnets.1", boc[-1], boc[-1])
        # conv_out (-> 2*latent) padded to 64 outputs so that quant_conv (1x1, K=8) becomes a K=64 GEMM
        self.convs["encoder.conv_out"] = _Conv(sd, "encoder.conv_out", pad_out_to=64)
        lc = cfg.latent_channels
        self.quant_w = pad_k(sd["quant_conv.weight"].reshape(2 * lc, 2 * lc), 64)
        self.quant_b = sd["quant_conv.bias"].contiguous()
        pq = torch.zeros(64, 64, dtype=dtype, device=self.device)
        pq[:lc, :lc] = sd["post_quant_conv.weight"].reshape(lc, lc)
        self.pq_w = pq
        self.pq_b = torch.zeros(64, dtype=dtype, device=self.device)
        self.pq_b[:lc] = sd["post_quant_conv.bias"]
        self.convs["decoder.conv_in"] = _Conv(sd, "decoder.conv_in", cin_pad=64)
        rboc = list(reversed(boc))
        self.dec_plan, out = [], rboc[0]
        for i, c in enumerate(rboc):
            cin, out = out, c
            rs = []
            for j in range(L + 1):
                p = f"decoder.up_blocks.{i}.resnets.{j}"
                res(p, cin if j == 0 else out, out)
                rs.append((p, cin if j == 0 else out, out))
            up = None
            if i != len(boc) - 1:
                up = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                self.convs[up] = _Conv(sd, up)
            self.dec_plan.append((rs, up))
        self.convs["decoder.conv_out"] = _Conv(sd, "decoder.conv_out", pad_out_to=4)

    # ---- blocks ----
    def _gn(self, x, name, silu):
        return ops.groupnorm(x, self.sd[name + ".weight"], self.sd[name + ".bias"], self.cfg.norm_num_groups, 1e-6, silu,
                             self._gn_stats)

    def _conv(self, x, name, B, H, W, stride=1, ups=False, pad=1, extra=(), **kw):
        cv = self.convs[name]
        if ups:
            Ho, Wo = 2 * H, 2 * W
        else:
            Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
            if pad == 0:                                   # VAE Downsample2D: F.pad(0,1,0,1) then conv s2 p0
                Ho, Wo = (H + 1 - 3) // stride + 1, (W + 1 - 3) // stride + 1
        segs = ops.conv_segs(x, 3, pad, length=cv.cin_pad) + list(extra)
        out = ops.gemm_conv(segs, cv.w, B * Ho * Wo, Ho=Ho, Wo=Wo, Hi=H, Wi=W, stride=stride, ups=ups, bias=cv.b, **kw)
        return out.view(B, Ho * Wo, cv.n), Ho, Wo

    def _resnet(self, p, x, cin, cout, B, H, W):
        g1 = self._gn(x, p + ".norm1", True)
        h, _, _ = self._conv(g1, p + ".conv1", B, H, W)
        g2 = self._gn(h, p + ".norm2", True)
        if cin != cout:
            out, _, _ = self._conv(g2, p + ".conv2", B, H, W, extra=[ops.SegSpec(x, 0, cin)])
        else:
            out, _, _ = self._conv(g2, p + ".conv2", B, H, W, res=x.reshape(B * H * W, cin))
        return out

    def _attn(self, p, x, B, N, C):
        sd, dt, dev = self.sd, self.dtype, self.device
        t = self._gn(x, p + ".group_norm", False).reshape(B * N, C)
        # q leaves its projection multiplied by C^-0.5 (fp32, before the one rounding): the N x N scores stored between the two
        # GEMMs are then softmax logits of O(1) magnitude instead of raw 512-term dot products
        q = ops.linear(t, sd[p + ".to_q.weight"], bias=sd[p + ".to_q.bias"], colscale_n=C, colscale=C ** -0.5)
        k = ops.linear(t, sd[p + ".to_k.weight"], bias=sd[p + ".to_k.bias"])
        vt = torch.empty(B, C, N, dtype=dt, device=dev)
        ops.linear(t, sd[p + ".to_v.weight"], bias=sd[p + ".to_v.bias"], vt=vt, vt_n0=0, vt_tokens=N, vt_perm=False)
        o = torch.empty(B * N, C, dtype=dt, device=dev)
        for b in range(B):
            s = ops.linear(q[b * N:(b + 1) * N], k[b * N:(b + 1) * N])          # [N][N] scores
            ops.softmax_rows(s, 1.0)
            ops.linear(s, vt[b], out=o[b * N:(b + 1) * N])
        out = ops.linear(o, sd[p + ".to_out.0.weight"], bias=sd[p + ".to_out.0.bias"], res=x.reshape(B * N, C))
        return out.view(B, N, C)

    def _mid(self, side, x, B, H, W):
        C = self.cfg.block_out_channels[-1]
        x = self._resnet(f"{side}.mid_block.resnets.0", x, C, C, B, H, W)
        x = self._attn(f"{side}.mid_block.attentions.0", x, B, H * W, C)
        return self._resnet(f"{side}.mid_block.resnets.1", x, C, C, B, H, W)

    # ---- public ----
    def encode_moments(self, image_nchw):
        """image: fp32 NCHW in [-1, 1] -> moments NHWC [B][h*w][8] (mean 0..3, logvar 4..7)."""
        B, _, H, W = image_nchw.shape
        x = ops.to_nhwc(image_nchw.contiguous().float(), self.dtype, cpad=_pad64(self.cfg.in_channels))
        x, H, W = self._conv(x, "encoder.conv_in", B, H, W)
        for rs, down in self.enc_plan:
            for p, ci, co in rs:
                x = self._resnet(p, x, ci, co, B, H, W)
            if down is not None:
                x, H, W = self._conv(x, down, B, H, W, stride=2, pad=0)
        x = self._mid("encoder", x, B, H, W)
        x = self._gn(x, "encoder.conv_norm_out", True)
        x, _, _ = self._conv(x, "encoder.conv_out", B, H, W)                      # [B][hw][64] (8 real)
        mom = ops.linear(x.reshape(B * H * W, 64), self.quant_w, bias=self.quant_b)
        return mom.view(B, H * W, 2 * self.cfg.latent_channels), H, W

    def encode_sample(self, image_nchw, noise, scale=None):
        """`vae.encode(x).latent_dist.sample() * scaling_factor` with caller-supplied N(0,1) noise [B][4][h][w] fp32."""
        B, _, H, W = image_nchw.shape
        # the kernels address every tensor through 32-bit buffer descriptors (< 2 GiB): the largest activation of the encoder is
        # B x H x W x 128 channels x 2 bytes, so large batches run in chunks (per image the arithmetic does not depend on the chunking)
        bmax = max(1, (2 ** 31 - 1) // (H * W * self.cfg.block_out_channels[0] * 2))
        sc = self.cfg.scaling_factor if scale is None else scale
        if B <= bmax:
            mom, h, w = self.encode_moments(image_nchw)
            return ops.vae_sample(mom, noise.contiguous(), sc)
        out = []
        for b0 in range(0, B, bmax):
            mom, h, w = self.encode_moments(image_nchw[b0:b0 + bmax])
            out.append(ops.vae_sample(mom, noise[b0:b0 + bmax].contiguous(), sc))
        return torch.cat(out)

    def decode(self, z_nchw):
        """z: fp32 NCHW latents (already divided by scaling_factor) -> image fp32 NCHW (pre-postprocess)."""
        B, _, H, W = z_nchw.shape
        z = ops.to_nhwc(z_nchw.contiguous().float(), self.dtype, cpad=64)
        x = ops.linear(z.reshape(B * H * W, 64), self.pq_w, bias=self.pq_b).view(B, H * W, 64)   # post_quant_conv
        x, _, _ = self._conv(x, "decoder.conv_in", B, H, W)
        x = self._mid("decoder", x, B, H, W)
        for rs, up in self.dec_plan:
            for p, ci, co in rs:
                x = self._resnet(p, x, ci, co, B, H, W)
            if up is not None:
                x, H, W = self._conv(x, up, B, H, W, ups=True)
        x = self._gn(x, "decoder.conv_norm_out", True)
        x, _, _ = self._conv(x, "decoder.conv_out", B, H, W)
        return ops.to_nchw(x, self.cfg.out_channels, (H, W))
