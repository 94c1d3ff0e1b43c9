"""Quick timing of the self-attention launches of the denoising loop (config 2: level 1 = 4 x 10 heads x 3072 + 3072 keys, level 2 = 4 x 20 x
768 + 768; CFG: the first two batch elements have no garment segment), 40 back-to-back launches each.  IDMVTON_HIP_LIB selects the build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import idm_vton_amd  # noqa
from idm_vton_amd import ops

dev, dt = torch.device("cuda"), torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev, dt)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, heads, N, b0, sc) in ((4, 10, 3072, 2, 1.0), (4, 20, 768, 2, 1.0), (4, 10, 3072, 2, 3.0), (12, 10, 3072, 12, 1.0)):
    C = heads * 64
    qk = r(B * N, 2 * C, sc=sc)
    vt = r(B, C, N)
    segs = [dict(k=qk[:, C:], vt=vt, nk=N, ldk=2 * C, ldvt=N)]
    if b0 < B:
        Bg = B - b0
        segs.append(dict(k=r(Bg * N, C, sc=sc), vt=r(Bg, C, N), nk=N, ldk=C, ldvt=N, b0=b0))
    out = torch.empty(B * N, C, dtype=dt, device=dev)
    keys = sum(N * (B - s.get("b0", 0)) for s in segs)
    fl = 4.0 * N * 64 * heads * keys
    # softmax form of the experimental ping-pong build of profiles/r03_attn_lazy_max_experiment.patch (tune bits 28-29: 0 row max per tile,
    # 1 lazy per half, 2 lazy single test); the committed kernel ignores the bits -- three timings of the same kernel then show the ORDER
    # effect of this loop (the first is 5-15 % slower than the third: clocks still ramping), so compare builds position by position
    for smode in (0, 1, 2):
        tune = (2 << 16) | (2 << 8) | 8 | (smode << 28)
        try:
            t = timed(lambda: ops.attention(qk, out, segs, heads, B=B, Nq=N, ldq=2 * C, ldo=C, q_prescaled=True, tune=tune))
        except Exception as e:
            print("   smode", smode, "failed:", str(e)[:80]); continue
        print(f"B={B} heads={heads} N={N} garment from b={b0} logit scale {sc} smode {smode}: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s", flush=True)
