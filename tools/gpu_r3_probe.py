"""Round-3 probes (timing only, HIP events, median of 7 behind a cache flush):
  pitch   the M = 3072 projection GEMMs with the activation operand's row pitch padded by 0 / 64 / 128 / 192 elements -- does the
          row stride (K*2 bytes = 20 or 80 cache lines) concentrate a tile's loads on a few L2 channels?
  stream  to_out-shaped GEMM (+bias +residual) with 16-bit vs fp32 residual stream in / out, and LayerNorm of a 16-bit vs fp32 row
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


def main():
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    res = {}

    def timed(fn, n=7):
        fn(); fn()
        ts = []
        for _ in range(n):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[n // 2]

    for (M, N, K) in ((3072, 1280, 1280), (3072, 1280, 5120), (3072, 3840, 1280), (12288, 640, 640), (12288, 640, 2560)):
        w = r(N, K)
        rs = r(M, N)
        for pad in (0, 64, 128, 192, 320):
            xb = r(M, K + pad)
            x = xb[:, :K]
            for tag, h in (("auto", 0), ("r128x64", hint(1, 128, 64)), ("r128x128", hint(1, 128, 128)), ("r128x256", hint(1, 128, 256))):
                try:
                    t = timed(lambda: ops.linear(x, w, res=rs, tile_hint=h))
                except Exception as e:                       # a tile the shape does not admit
                    t = None
                res[f"pitch {M}x{N}x{K} pad{pad} {tag}"] = t
                print(f"pitch {M}x{N}x{K} pad{pad:3d} {tag:9s} {t if t is None else round(t, 1)} us" + ("" if t is None else f"  {2.0 * M * N * K / t / 1e6:.0f} TF"), flush=True)
    for (M, N, K) in ((3072, 1280, 1280), (12288, 640, 640), (3072, 1280, 5120)):
        x, w, b = r(M, K), r(N, K), r(N)
        r16, r32 = r(M, N), torch.randn(M, N, device=dev)
        o32 = torch.empty(M, N, device=dev)
        g, bt = r(N), r(N)
        t_a = timed(lambda: ops.linear(x, w, bias=b, res=r16))
        t_b = timed(lambda: ops.linear(x, w, bias=b, res=r32, out=o32))
        t_c = timed(lambda: ops.linear(x, w, bias=b, res=r32))
        t_l16 = timed(lambda: ops.layernorm(r16, g, bt))
        t_l32 = timed(lambda: ops.layernorm(r32, g, bt))
        res[f"stream {M}x{N}x{K}"] = dict(gemm_16=t_a, gemm_f32_in_out=t_b, gemm_f32_in_16_out=t_c, ln_16=t_l16, ln_f32=t_l32)
        print(f"stream {M}x{N}x{K}: gemm 16-bit res/out {t_a:.1f} us, fp32 res+out {t_b:.1f}, fp32 res 16-bit out {t_c:.1f}; layernorm 16-bit in {t_l16:.1f}, fp32 in {t_l32:.1f}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r3_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
