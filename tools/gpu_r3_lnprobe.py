"""LN-fold cost probe: producer GEMM with / without rowstats_out; consumer GEMMs with / without the folded LayerNorm (warm, 50 back-to-back launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from idm_vton_amd import ops
from idm_vton_amd.weights import interleave_geglu

dt, dev = torch.bfloat16, "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (M, C) in ((3072, 1280), (12288, 640), (9216, 1280)):
    a, wp, bp, rsd = r(M, C), r(C, C), r(C), r(M, C)
    stats = ops.RowStats(M, C, dev)
    gam, bet = r(C) + 1.0, r(C)
    t0 = timed(lambda: ops.linear(a, wp, bias=bp, res=rsd))
    t1 = timed(lambda: ops.linear(a, wp, bias=bp, res=rsd, rowstats_out=stats))
    hs = ops.linear(a, wp, bias=bp, res=rsd, rowstats_out=stats)
    n1 = ops.layernorm(hs, gam, bet)
    tl = timed(lambda: ops.layernorm(hs, gam, bet))
    print(f"M={M} C={C}: producer plain {t0:.1f} us, +rowstats {t1:.1f} us; layernorm kernel {tl:.1f} us", flush=True)
    wq = r(C, C); wqs, cvq = ops.ln_fold_weights(wq, gam, bet)
    t2 = timed(lambda: ops.linear(n1, wq)); t3 = timed(lambda: ops.linear(hs, wqs, ln=(stats, cvq)))
    print(f"   to_q   N={C}: plain {t2:.1f} us, folded {t3:.1f} us", flush=True)
    w3 = r(3 * C, C); w3s, cv3 = ops.ln_fold_weights(w3, gam, bet)
    B = M // 768 if C == 1280 else M // 3072
    N = M // B
    qk = torch.empty(M, 2 * C, dtype=dt, device=dev); vt = torch.empty(B, C, N, dtype=dt, device=dev)
    t4 = timed(lambda: ops.linear(n1, w3, out=qk, vt=vt, vt_n0=2 * C, vt_tokens=N, colscale_n=C, colscale=ops.QSCALE))
    t5 = timed(lambda: ops.linear(hs, w3s, out=qk, vt=vt, vt_n0=2 * C, vt_tokens=N, colscale_n=C, colscale=ops.QSCALE, ln=(stats, cv3)))
    print(f"   qkv    N={3 * C}: plain {t4:.1f} us, folded {t5:.1f} us", flush=True)
    wg, bg = r(8 * C, C), r(8 * C)
    wi, bi = interleave_geglu(wg, bg); wis, cvg = ops.ln_fold_weights(wi, gam, bet)
    t6 = timed(lambda: ops.linear(n1, wi, bias=bi, geglu=True)); t7 = timed(lambda: ops.linear(hs, wis, bias=bi, geglu=True, ln=(stats, cvg)))
    print(f"   geglu  N={8 * C}: plain {t6:.1f} us, folded {t7:.1f} us", flush=True)
