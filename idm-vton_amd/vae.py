"""SDXL AutoencoderKL encode/decode on the HIP kernels (NHWC).

Host-side mirror of the diffusers AutoencoderKL the reference calls at src/tryon_pipeline.py:924,1646,1876 (block
structure: src/unet_block_hacked_tryon.py:505-627,1292-1349,2511-2568).  Same state-dict keys.
The single-head 512-wide mid-block attention runs as gemm_conv (QK^T) -> softmax_rows -> gemm_conv (PV).

DECODE RUNS AT FP32-EQUIVALENT PRECISION (`precise_decode=True`, the default).  The reference decodes with the VAE upcast to
fp32 (src/tryon_pipeline.py:1076-1093 upcast_vae, :1868-1880 needs_upcasting / force_upcast); 16-bit storage of the ~60 chained
feature maps costs 2.5e-3 (fp16) / 2.0e-2 (bf16) of the image range (profiles/r03_v7_fullsize_parity.json).  The MFMA needs 16-bit
operands, so a value x travels as the bf16 pair hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits, fp32's exponent range -- fp16
pairs would overflow where the reference's fp32 does not) and a product is

    x . w = hi . w_hi + lo . w_hi + hi . w_lo   (+ O(2^-17))

= extra K-segments of ONE gemm_conv launch: activations [hi | lo] (segments (coff 0, len 2C) and (coff 0, len C)) against weights laid
out [w_hi | w_hi] [w_lo]; the w_lo block is dropped where the weights are exactly bf16 (a bf16 checkpoint: 2x the MFMA work instead of
3x).  Everything between two GEMMs stays fp32: accumulators -> fp32 outputs (+ fp32 bias, fp32 residual) -> GroupNorm reads fp32 and
writes the next operand pair.  The single-head attention's two activation x activation products use the 3-term form.
"""
import torch

from . import ffi, ops
from .config import VAEConfig
from .unet import _Conv, _pad64
from .weights import pad_k


def _hi_lo(w32):
    """fp32 -> (bf16 hi, bf16 lo, lo is nonzero anywhere)."""
    hi = w32.to(torch.bfloat16)
    lo = (w32 - hi.float()).to(torch.bfloat16)
    return hi, lo, bool((lo != 0).any().item())


class _PConv:
    """Split-precision conv / linear weights: [N][taps * 2 * Cin] = per tap [w_hi | w_hi] (against activation pairs [hi | lo]), then --
    unless every w_lo is zero -- [N][taps * Cin] of w_lo (against hi again); a fused 1x1 shortcut the same way on its own input pair.
    `segs(x_pair, k, pad)` builds the matching K-segments.  bias fp32."""

    def __init__(self, w32, b32, cin_pad=None, shortcut=None, pad_out_to=None, hi_only=False):
        """hi_only: the one-term product hi . w_hi (the activation's bf16 rounding only: 1x the MFMA work) -- for the layers where the decode
        bar does not need the low halves (HipVAE.HI_ONLY)."""
        dev = w32.device
        self.hi_only = hi_only
        if w32.dim() == 2:
            w32 = w32[:, :, None, None]
        co, ci, kh, kw = w32.shape
        self.k, self.cin = kh, cin_pad or ci
        wk = torch.zeros(co, kh * kw, self.cin, dtype=torch.float32, device=dev)
        wk[..., :ci] = w32.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
        hi, lo, self.three = _hi_lo(wk)
        if hi_only:
            self.three = False
            parts = [hi.reshape(co, -1)]
        else:
            parts = [torch.cat([hi, hi], dim=2).reshape(co, -1)]
        if self.three:
            parts.append(lo.reshape(co, -1))
        b = b32.float().clone() if b32 is not None else torch.zeros(co, dtype=torch.float32, device=dev)
        self.sc_cin, self.sc_three = 0, False
        if shortcut is not None:
            ws32, bs32 = shortcut
            ws32 = ws32.reshape(ws32.shape[0], ws32.shape[1]).float()
            shi, slo, self.sc_three = _hi_lo(ws32)
            self.sc_cin = ws32.shape[1]
            if hi_only:
                self.sc_three = False
                parts.append(shi)
            else:
                parts.append(torch.cat([shi, shi], dim=1))
            if self.sc_three:
                parts.append(slo)
            b = b + bs32.float()
        w = torch.cat(parts, dim=1)
        if pad_out_to and pad_out_to > co:
            w = torch.cat([w, torch.zeros(pad_out_to - co, w.shape[1], dtype=w.dtype, device=dev)])
            b = torch.cat([b, torch.zeros(pad_out_to - co, dtype=b.dtype, device=dev)])
        self.w, self.b, self.n = w.contiguous(), b.contiguous(), w.shape[0]

    def segs(self, xp, pad=1, extra=None):
        """xp: activation pair [..., 2*cin]; extra: the shortcut's input pair [..., 2*sc_cin]."""
        k, c = self.k, self.cin
        taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)] if k > 1 else [(0, 0)]
        if self.hi_only:                                     # the hi half of each pair only
            out = [ops.SegSpec(xp, 0, c, dy, dx) for dy, dx in taps]
            if self.sc_cin:
                out.append(ops.SegSpec(extra, 0, self.sc_cin))
            return out
        out = [ops.SegSpec(xp, 0, 2 * c, dy, dx) for dy, dx in taps]
        if self.three:
            out += [ops.SegSpec(xp, 0, c, dy, dx) for dy, dx in taps]
        if self.sc_cin:
            out.append(ops.SegSpec(extra, 0, 2 * self.sc_cin))
            if self.sc_three:
                out.append(ops.SegSpec(extra, 0, self.sc_cin))
        return out


class HipVAE:
    @property
    def precise_decode(self):
        """Read-only after __init__ (the 16-bit decoder's weights are not built for precise_decode=True and vice versa): construct another
        HipVAE for an A/B."""
        return self._precise_decode

    def __init__(self, cfg: VAEConfig, state_dict, dtype=torch.bfloat16, device="cuda", precise_decode=True):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self._precise_decode = bool(precise_decode)         # fixed at construction: only the chosen decoder's weights are prepared
        if self.precise_decode:
            self._prep_precise({k: v.to(device=self.device, dtype=torch.float32) for k, v in state_dict.items()
                                if k.startswith(("decoder.", "post_quant_conv."))})
        # the split-precision decoder holds its own (hi, lo) weight pairs: the 16-bit copies of the decoder's weights and their padded conv
        # objects would be dead memory on every rank, so they are only built for precise_decode=False
        dec16 = not self.precise_decode
        sd = {k: v.to(device=self.device, dtype=dtype) for k, v in state_dict.items()
              if dec16 or not k.startswith(("decoder.", "post_quant_conv."))}
        self.sd = sd
        boc, L = cfg.block_out_channels, cfg.layers_per_block
        self.convs = {}
        self._gn_stats = torch.empty(ops.GN_STATS_DOUBLES, dtype=torch.float64, device=self.device)

        def res(p, cin, cout):
            if p.startswith("decoder.") and not dec16:
                return
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
        if dec16:
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
                if dec16:
                    self.convs[up] = _Conv(sd, up)
            self.dec_plan.append((rs, up))
        if dec16:
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
        o = torch.empty(B * N, C, dtype=dt, device=dev)
        if N % 64:
            # a token count that is not a multiple of the GEMM's K granule (latent H*W of an odd size, e.g. 33 x 25): keys padded to Np rows of
            # zeros, the softmax runs over the first N columns and writes 0 into the rest, V^T zero-padded the same way (torch glue: rare path)
            Np = (N + 63) // 64 * 64
            v = ops.linear(t, sd[p + ".to_v.weight"], bias=sd[p + ".to_v.bias"])
            for b in range(B):
                kp = torch.zeros(Np, C, dtype=dt, device=dev)
                kp[:N] = k[b * N:(b + 1) * N]
                vtp = torch.zeros(C, Np, dtype=dt, device=dev)
                vtp[:, :N] = v[b * N:(b + 1) * N].t()
                s = ops.linear(q[b * N:(b + 1) * N], kp)                         # [N][Np]
                ops.softmax_rows(s, 1.0, n_valid=N)
                ops.linear(s, vtp, out=o[b * N:(b + 1) * N])
            out = ops.linear(o, sd[p + ".to_out.0.weight"], bias=sd[p + ".to_out.0.bias"], res=x.reshape(B * N, C))
            return out.view(B, N, C)
        vt = torch.empty(B, C, N, dtype=dt, device=dev)
        ops.linear(t, sd[p + ".to_v.weight"], bias=sd[p + ".to_v.bias"], vt=vt, vt_n0=0, vt_tokens=N, vt_perm=False)
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

    # ---- split-precision decode (module docstring) ----
    PDT = torch.bfloat16                                     # operand pairs are bf16 whatever the engine's storage type: fp32's exponent range

    def _prep_precise(self, sd32, hi_only=()):
        cfg = self.cfg
        self._sd32_dec = sd32 if getattr(self, "_keep_sd32", False) else None
        boc, L = cfg.block_out_channels, cfg.layers_per_block
        self.p32 = {k: v.contiguous() for k, v in sd32.items() if ".norm" in k or ".group_norm" in k or k.endswith("conv_norm_out.weight")
                    or k.endswith("conv_norm_out.bias")}
        pc = self.pconvs = {}

        def res(p, cin, cout):
            pc[p + ".conv1"] = _PConv(sd32[p + ".conv1.weight"], sd32[p + ".conv1.bias"], hi_only=(p + ".conv1") in hi_only)
            sc = (sd32[p + ".conv_shortcut.weight"], sd32[p + ".conv_shortcut.bias"]) if cin != cout else None
            pc[p + ".conv2"] = _PConv(sd32[p + ".conv2.weight"], sd32[p + ".conv2.bias"], shortcut=sc, hi_only=(p + ".conv2") in hi_only)

        lc = cfg.latent_channels
        pq = torch.zeros(64, 64, dtype=torch.float32, device=self.device)
        pq[:lc, :lc] = sd32["post_quant_conv.weight"].reshape(lc, lc)
        pqb = torch.zeros(64, dtype=torch.float32, device=self.device)
        pqb[:lc] = sd32["post_quant_conv.bias"]
        pc["post_quant_conv"] = _PConv(pq, pqb)
        pc["decoder.conv_in"] = _PConv(sd32["decoder.conv_in.weight"], sd32["decoder.conv_in.bias"], cin_pad=64)
        res("decoder.mid_block.resnets.0", boc[-1], boc[-1])
        res("decoder.mid_block.resnets.1", boc[-1], boc[-1])
        a = "decoder.mid_block.attentions.0"
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            pc[f"{a}.{nm}"] = _PConv(sd32[f"{a}.{nm}.weight"], sd32[f"{a}.{nm}.bias"])
        rboc, out = list(reversed(boc)), boc[-1]
        for i, c in enumerate(rboc):
            cin, out = out, c
            for j in range(L + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else out, out)
            if i != len(boc) - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                pc[u] = _PConv(sd32[u + ".weight"], sd32[u + ".bias"], hi_only=u in hi_only)
        pc["decoder.conv_out"] = _PConv(sd32["decoder.conv_out.weight"], sd32["decoder.conv_out.bias"], pad_out_to=8)

    def _pgn(self, x32, name, silu):
        return ops.groupnorm(x32, self.p32[name + ".weight"], self.p32[name + ".bias"], self.cfg.norm_num_groups, 1e-6, silu,
                             self._gn_stats, split_dtype=self.PDT)

    def _pconv(self, xp, name, B, H, W, ups=False, extra=None, res=None):
        """xp: operand pair [B][H*W][2*Cin] -> fp32 [B][Ho*Wo][N]."""
        cv = self.pconvs[name]
        Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
        out = ops.gemm_conv(cv.segs(xp, 1, extra), cv.w, B * Ho * Wo, Ho=Ho, Wo=Wo, Hi=H, Wi=W, ups=ups, bias=cv.b, res=res, out_f32=True)
        return out.view(B, Ho * Wo, cv.n), Ho, Wo

    def _plin(self, xp2d, name, **kw):
        cv = self.pconvs[name]
        return ops.gemm_conv(cv.segs(xp2d, 0), cv.w, xp2d.shape[0], bias=cv.b, out_f32=True, **kw)

    def _pair(self, x32_2d):
        return ops.split(x32_2d, self.PDT, ffi.SPLIT_ACT)

    def _presnet(self, p, x, cin, cout, B, H, W):
        g1 = self._pgn(x, p + ".norm1", True)
        h, _, _ = self._pconv(g1, p + ".conv1", B, H, W)
        g2 = self._pgn(h, p + ".norm2", True)
        if cin != cout:
            xs = self._pair(x.reshape(B * H * W, cin)).view(B, H * W, 2 * cin)
            out, _, _ = self._pconv(g2, p + ".conv2", B, H, W, extra=xs)
        else:
            out, _, _ = self._pconv(g2, p + ".conv2", B, H, W, res=x.reshape(B * H * W, cin))
        return out

    def _pattn(self, p, x, B, N, C):
        t = self._pgn(x, p + ".group_norm", False).reshape(B * N, 2 * C)
        q = self._plin(t, p + ".to_q", colscale_n=C, colscale=C ** -0.5)          # scaled in fp32 before the operand split
        k = self._plin(t, p + ".to_k")
        v = self._plin(t, p + ".to_v")
        o = torch.empty(B * N, C, dtype=torch.float32, device=self.device)
        # query rows in chunks: the probability pair [rows][2N] is addressed through a 32-bit buffer descriptor (< 2 GiB): all 12288 rows at
        # once at 768x1024, 16384 of the 24576 at 1024x1536
        Np = (N + 63) // 64 * 64                             # keys padded (zero rows) to the K granule of the P.V product; == N at the usual sizes
        qc = max(256, min(N, (2 ** 31 - 1) // (4 * Np) // 256 * 256))
        for b in range(B):
            kb, vb = k[b * N:(b + 1) * N], v[b * N:(b + 1) * N]
            if Np != N:
                kb, vb = torch.cat([kb, kb.new_zeros(Np - N, C)]), torch.cat([vb, vb.new_zeros(Np - N, C)])
            k3 = ops.split(kb, self.PDT, ffi.SPLIT_W3)                                                          # [Np][3C]
            vt3 = ops.split(vb, self.PDT, ffi.SPLIT_W3T)                                                        # [C][3Np]
            for r0 in range(0, N, qc):
                sl = slice(b * N + r0, b * N + min(r0 + qc, N))
                n = sl.stop - sl.start
                qp = self._pair(q[sl])
                s = ops.gemm_conv([ops.SegSpec(qp, 0, 2 * C), ops.SegSpec(qp, 0, C)], k3, n, out_f32=True)      # [n][Np] logits, 3-term product
                pp = ops.softmax_rows_split(s, 1.0, self.PDT, n_valid=0 if Np == N else N)
                del s
                ops.gemm_conv([ops.SegSpec(pp, 0, 2 * Np), ops.SegSpec(pp, 0, Np)], vt3, n, out=o[sl])
                del pp
            del k3, vt3
        out = self._plin(self._pair(o), p + ".to_out.0", res=x.reshape(B * N, C))
        return out.view(B, N, C)

    def _decode_precise(self, z_nchw):
        B, _, H, W = z_nchw.shape
        C = self.cfg.block_out_channels[-1]
        zp = ops.to_nhwc(z_nchw.contiguous().float(), self.PDT, cpad=64, split=True)                            # [B][hw][128]
        x = self._plin(zp.reshape(B * H * W, 128), "post_quant_conv")                                           # fp32 [B*hw][64]
        x, _, _ = self._pconv(self._pair(x).view(B, H * W, 128), "decoder.conv_in", B, H, W)
        x = self._presnet("decoder.mid_block.resnets.0", x, C, C, B, H, W)
        x = self._pattn("decoder.mid_block.attentions.0", x, B, H * W, C)
        x = self._presnet("decoder.mid_block.resnets.1", x, C, C, B, H, W)
        for rs, up in self.dec_plan:
            for p, ci, co in rs:
                x = self._presnet(p, x, ci, co, B, H, W)
            if up is not None:
                c = x.shape[-1]
                x, H, W = self._pconv(self._pair(x.reshape(B * H * W, c)).view(B, H * W, 2 * c), up, B, H, W, ups=True)
        x = self._pgn(x, "decoder.conv_norm_out", True)
        x, _, _ = self._pconv(x, "decoder.conv_out", B, H, W)                                                   # fp32 [B][HW][8]
        return ops.to_nchw(x, self.cfg.out_channels, (H, W))

    def decode(self, z_nchw):
        """z: fp32 NCHW latents (already divided by scaling_factor) -> image fp32 NCHW (pre-postprocess)."""
        B, _, H, W = z_nchw.shape
        if self.precise_decode:
            # the largest tensor addressed through a 32-bit buffer descriptor is the operand pair of the full-resolution 256-channel
            # resnet: 64 h w pixels x 2 x 256 channels x 2 bytes per image
            c_hi = self.cfg.block_out_channels[1] if len(self.cfg.block_out_channels) > 1 else self.cfg.block_out_channels[0]
            bmax = max(1, (2 ** 31 - 1) // (64 * H * W * 2 * c_hi * 2))
            if B <= bmax:
                return self._decode_precise(z_nchw)
            return torch.cat([self._decode_precise(z_nchw[b0:b0 + bmax]) for b0 in range(0, B, bmax)])
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
