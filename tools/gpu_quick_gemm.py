"""Quick A/B of GEMM tile variants on the real config-2 shapes (warm: 20 back-to-back launches; cold: behind a cache flush, median of 7)."""
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

    def cold(fn):
        fn(); fn()
        ts = []
        for _ in range(7):
            flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[3]

    def warm(fn, n=20):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    cases = []
    big = [("q256x256", hint(3, 256, 256)), ("r256x256", hint(1, 256, 256)), ("p256x256", hint(2, 256, 256)), ("r128x256", hint(1, 128, 256))]
    for (M, C, tag) in ((3072, 1280, "TryonNet L2"), (12288, 640, "TryonNet L1"), (9216, 1280, "GarmentNet L2 x6"), (36864, 640, "GarmentNet L1 x6")):
        x, w, b = r(M, C), r(8 * C, C), r(8 * C)
        wi, bi = interleave_geglu(w, b)
        cases.append((f"geglu {M}x{8 * C}x{C} ({tag})", 2 * M * 8 * C * C, lambda h, x=x, wi=wi, bi=bi: ops.linear(x, wi, bias=bi, geglu=True, tile_hint=h), big))
        x4, w4, rs = r(M, 4 * C), r(C, 4 * C), r(M, C)
        cases.append((f"ff2 {M}x{C}x{4 * C} ({tag})", 2 * M * C * 4 * C, lambda h, x4=x4, w4=w4, rs=rs: ops.linear(x4, w4, res=rs, tile_hint=h),
                      big + [("r128x128", hint(1, 128, 128))]))
    x2, w5 = r(3072, 1280), r(3840, 1280)
    cases.append(("qkv-like plain 3072x3840x1280", 2 * 3072 * 3840 * 1280, lambda h: ops.linear(x2, w5, tile_hint=h), big))
    for name, fl, fn, vs in cases:
        out = []
        for tag, h in vs:
            try:
                tc, tw = cold(lambda: fn(h)), warm(lambda: fn(h))
                out.append(f"{tag}={tc:.1f}/{tw:.1f}us ({fl / tw / 1e6:.0f}TF)")
            except Exception as e:
                out.append(f"{tag}=ERR")
        print(name, " ".join(out), flush=True)


if __name__ == "__main__":
    main()
