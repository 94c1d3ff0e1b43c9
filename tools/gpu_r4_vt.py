"""Round 4 probe: the fused QKV projection's V^T part, LDS-transposed stores vs stores straight from the accumulators (tile_hint bit 14), per
tile, timed three ways: warm (back to back), cold (behind a 640 MB flush), and the tuner's way (flush, then operands touched back into cache)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ffi, ops  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


def main():
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s, scale=0.5: (torch.randn(*s, device=dev) * scale).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    L = ffi.lib()
    st = torch.cuda.current_stream().cuda_stream
    touch = lambda t: L.idmvton_prefetch(C.c_void_p(t.data_ptr()), C.c_uint64(t.numel() * t.element_size()), 0, C.c_void_p(st))
    for (M, Cc, tok) in ((3072, 1280, 768), (9216, 1280, 768), (12288, 640, 3072)):
        x, w = r(M, Cc), r(3 * Cc, Cc, scale=0.03)
        qk, vt = torch.empty(M, 2 * Cc, dtype=dt, device=dev), torch.empty(M // tok, Cc, tok, dtype=dt, device=dev)
        fn = lambda h: ops.linear(x, w, out=qk, vt=vt, vt_n0=2 * Cc, vt_tokens=tok, colscale_n=Cc, colscale=ops.QSCALE, tile_hint=h)
        fn(hint(1, 128, 64) | 0x4000)
        ref = vt.clone()
        print(f"QKV {M}x{3 * Cc}x{Cc}")
        for tag, h in (("p128x64", hint(2, 128, 64)), ("r128x64", hint(1, 128, 64)), ("r128x128", hint(1, 128, 128)), ("r128x256", hint(1, 128, 256)),
                       ("r256x256", hint(1, 256, 256)), ("h256x256", hint(5, 256, 257))):
            row = []
            for direct in (0, 1):
                hh = h | (0x4000 if direct else 0)
                vt.fill_(float("nan")); fn(hh); ok = torch.equal(vt, ref)
                ts = {}
                for mode in ("warm", "cold", "tuner"):
                    v = []
                    for _ in range(7):
                        if mode != "warm":
                            flush.zero_()
                        if mode == "tuner":
                            touch(w); touch(x)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        n = 10 if mode == "warm" else 1
                        e0.record()
                        for _ in range(n):
                            fn(hh)
                        e1.record(); e1.synchronize()
                        v.append(e0.elapsed_time(e1) * 1e3 / n)
                    ts[mode] = sorted(v)[3]
                row.append(("direct" if direct else "lds   ") + f" ok={ok} " + " ".join(f"{k}={v:.1f}" for k, v in ts.items()))
            print(f"  {tag:9s} " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
