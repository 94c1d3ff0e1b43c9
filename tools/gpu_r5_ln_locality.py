"""Round 5 probe: what does a LayerNorm launch cost by WHERE its input lives?  (a) in the chain: right behind the GEMM (+ residual) that wrote
its input from other CUs / XCDs (the state of every LayerNorm of the loop); (b) the same input read a second time by the same launch geometry
(lines resident in the reading XCD's own L2: what XCD-affine token ownership would give it); (c) behind a 640 MB flush (HBM).  HIP events around
the LayerNorm launch only, median of 25.  python tools/gpu_r5_ln_locality.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402


def med(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


def main():
    dev, dt = "cuda", torch.bfloat16
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    for M, C in ((3072, 1280), (12288, 640), (9216, 1280)):
        r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
        a, w, b, res, g, bt = r(M, C), r(C, C) * 0.05, r(C), r(M, C), r(C), r(C)
        hs = torch.empty(M, C, dtype=dt, device=dev)
        out = {}
        for mode in ("chain", "same_xcd_l2", "hbm"):
            ts = []
            for _ in range(25):
                flush.zero_()
                ops.linear(a, w, bias=b, res=res, out=hs)                # the producer: attn1.to_out + residual
                if mode == "same_xcd_l2":
                    ops.layernorm(hs, g, bt, 1e-5)                       # first pass pulls the rows into the reading XCDs' L2s
                if mode == "hbm":
                    flush.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.layernorm(hs, g, bt, 1e-5)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            out[mode] = round(med(ts), 2)
        print(f"LayerNorm M={M} C={C} ({M * C * 4 / 1e6:.1f} MB in + out): {out} us", flush=True)


if __name__ == "__main__":
    main()
