"""Quick A/B of GEMM tile variants on the real config-2 shapes (cold caches, HIP events, median of 7)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402
from idm_vton_amd.weights import interleave_geglu  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


def main():
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)

    def timed(fn):
        fn(); fn()
        ts = []
        for _ in range(7):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[3]

    cases = []
    x, w, b = r(3072, 1280), r(10240, 1280), r(10240)
    wi, bi = interleave_geglu(w, b)
    cases.append(("geglu 3072x10240x1280", 2 * 3072 * 10240 * 1280, lambda h: ops.linear(x, wi, bias=bi, geglu=True, tile_hint=h),
                  [("r256x256", hint(1, 256, 256)), ("p256x256", hint(2, 256, 256)), ("r128x256", hint(1, 128, 256)), ("p128x256", hint(2, 128, 256))]))
    x2, w2, rs2 = r(3072, 1280), r(1280, 1280), r(3072, 1280)
    cases.append(("proj 3072x1280x1280", 2 * 3072 * 1280 * 1280, lambda h: ops.linear(x2, w2, res=rs2, tile_hint=h),
                  [("r128x128", hint(1, 128, 128)), ("r128x64", hint(1, 128, 64)), ("p128x64", hint(2, 128, 64)), ("p64x64", hint(2, 64, 64))]))
    x3, rs3 = r(1536, 1280), r(1536, 1280)
    cases.append(("proj 1536x1280x1280", 2 * 1536 * 1280 * 1280, lambda h: ops.linear(x3, w2, res=rs3, tile_hint=h),
                  [("r128x64", hint(1, 128, 64)), ("p128x64", hint(2, 128, 64)), ("r64x64", hint(1, 64, 64)), ("p64x64", hint(2, 64, 64))]))
    x4, w4 = r(3072, 5120), r(1280, 5120)
    cases.append(("ff2 3072x1280x5120", 2 * 3072 * 1280 * 5120, lambda h: ops.linear(x4, w4, res=rs2, tile_hint=h),
                  [("r128x128", hint(1, 128, 128)), ("p128x64", hint(2, 128, 64)), ("p128x256", hint(2, 128, 256))]))
    w5 = r(3840, 1280)
    cases.append(("qkv 3072x3840x1280", 2 * 3072 * 3840 * 1280, lambda h: ops.linear(x2, w5, tile_hint=h),
                  [("r128x128", hint(1, 128, 128)), ("r128x256", hint(1, 128, 256)), ("p128x256", hint(2, 128, 256)), ("p256x256", hint(2, 256, 256))]))
    for name, fl, fn, vs in cases:
        print(name, " ".join(f"{tag}={(t := timed(lambda: fn(h))):.1f}us/{fl / t / 1e6:.0f}TF" for tag, h in vs), flush=True)


if __name__ == "__main__":
    main()
