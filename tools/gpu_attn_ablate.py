"""Ablation table of attn_sp_kernel (dbg build: tools/build_variant.sh dbg attention.hip "-DATTN_DBG -DSP_SGB=0"; IDMVTON_HIP_LIB=.ab_r06/libdbg.so).
Mask bits: 1 no softmax VALU, 2 no MFMA, 4 no LDS fragment reads, 8 no in-loop DMA, 16 no workgroup barrier.  Timing only (results are wrong)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idm_vton_amd  # noqa
from idm_vton_amd import ops
dev, dt = torch.device("cuda"), torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, dt)
def timed(fn, n=60):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
MASKS = [0, 32, 1, 33, 0, 32]
names = {1: "softmax", 2: "mfma", 4: "ldsread", 8: "dma", 16: "barrier", 32: "rotation", 64: "(all waves group 0 else 1)"}
for (B, heads, N, b0) in ((4, 10, 3072, 2), (4, 20, 768, 2)):
    C = heads * 64
    qk = r(B * N, 2 * C, sc=0.35); vt = r(B, C, N)
    segs = [dict(k=qk[:, C:], vt=vt, nk=N, ldk=2 * C, ldvt=N), dict(k=r((B - b0) * N, C, sc=0.35), vt=r(B - b0, C, N), nk=N, ldk=C, ldvt=N, b0=b0)]
    out = torch.empty(B * N, C, dtype=dt, device=dev)
    for rep in range(2):
        for m in MASKS:
            t = timed(lambda: ops.attention(qk, out, segs, heads, B=B, Nq=N, ldq=2 * C, ldo=C, q_prescaled=True, tune=((32 + m) << 16) | (3 << 8) | 8))
            if rep == 1:
                has = [v for k, v in names.items() if not (m & k)]
                print(f"N={N} mask {m:2d}  {t:7.1f} us   keeps: {' '.join(has) if has else '(loop skeleton only)'}", flush=True)
