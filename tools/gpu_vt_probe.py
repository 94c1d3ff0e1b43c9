"""What the V^T epilogue costs: the fused q|k|v projection (M x 3C x C, last C columns written transposed in key order) vs the same GEMM with a
plain epilogue, cold operands, tuned tile and a few others."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idm_vton_amd  # noqa
from idm_vton_amd import ops
dev, dt = torch.device("cuda"), torch.bfloat16
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev, dt)
flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
def timed(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
hint = lambda v, bn, bm: (v << 28) | (bn << 16) | bm
for (B, N, C) in ((4, 768, 1280), (12, 768, 1280), (4, 3072, 640)):
    M = B * N
    x, w = r(M, C), r(3 * C, C)
    out = torch.empty(M, 2 * C, dtype=dt, device=dev)
    vt = torch.empty(B, C, N, dtype=dt, device=dev)
    full = torch.empty(M, 3 * C, dtype=dt, device=dev)
    for name, h in (("table/auto", 0), ("w16_128x256", hint(6, 128, 256)), ("w12_256x192", hint(6, 256, 192)), ("h_256x192", hint(5, 256, 192)), ("r_128x256", hint(1, 128, 256))):
        try:
            t_vt = timed(lambda: ops.linear(x, w, out=out, vt=vt, vt_n0=2 * C, vt_tokens=N, colscale_n=C, colscale=ops.QSCALE, tile_hint=h))
            t_pl = timed(lambda: ops.linear(x, w, out=full, colscale_n=C, colscale=ops.QSCALE, tile_hint=h))
            print(f"M={M} N={3 * C} K={C} {name:12s} with V^T {t_vt:6.1f} us   plain {t_pl:6.1f} us   diff {t_vt - t_pl:+5.1f}", flush=True)
        except Exception as e:
            print(name, "n/a", str(e)[:80])
