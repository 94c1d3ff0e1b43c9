"""attn_pf_kernel (tune kernel 7 / 8) against fp32 SDPA on the edge cases of tests/kernel_checks.py, then timed beside the committed
ping-pong builds at the two CFG launches of the denoising loop (level 1: 4 x 10 heads x 3072 + 3072 keys; level 2: 4 x 20 x 768 + 768)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import idm_vton_amd  # noqa
from idm_vton_amd import ops
from tests import kernel_checks as kc

dev = torch.device("cuda")
pf = lambda k, thr=0: (thr << 26) | (k << 16) | (3 << 8) | 8
VARIANTS = [("pp_s2d0", kc.pp_tune(2, 0)), ("pp_s2d1", kc.pp_tune(2, 1)), ("pf_vsum", pf(8)), ("sp8", pf(16)), ("sp4", (16 << 16) | (3 << 8) | 4)]
TIMING_ONLY = []
CHECK = os.environ.get("CHECK", "sp")
if len(sys.argv) > 1:
    VARIANTS += [(f"k{k}", pf(int(k))) for k in sys.argv[1].split(",")]

bad = 0
for dt in (torch.bfloat16, torch.float16):
    for name, tn in VARIANTS:
        if not name.startswith((CHECK, "k")):
            continue
        for thr in (0, 1, 2):
            t = tn | (thr << 26)
            checks = [
                ("2seg_cfg_N768", lambda: kc.check_attn_self(4, 4, 768, dt, dev, n_garm=768, b0=2, tune=t, prescaled=True)),
                ("ragged_N200", lambda: kc.check_attn_self(2, 2, 200, dt, dev, n_garm=200, b0=1, tune=t, prescaled=True)),
                ("odd_tiles_N320", lambda: kc.check_attn_self(2, 2, 320, dt, dev, n_garm=192, b0=1, tune=t, prescaled=True)),
                ("big_logits", lambda: kc.check_attn_self(2, 2, 256, dt, dev, n_garm=256, b0=1, scale=4.0, tune=t, prescaled=True)),
                ("N16", lambda: kc.check_attn_self(2, 1, 16, dt, dev, n_garm=16, b0=1, tune=t, prescaled=True)),
                ("N64_1tile", lambda: kc.check_attn_self(1, 1, 64, dt, dev, tune=t, prescaled=True)),
                ("N128_2tiles", lambda: kc.check_attn_self(1, 2, 128, dt, dev, tune=t, prescaled=True)),
                ("1seg_N1000", lambda: kc.check_attn_self(1, 3, 1000, dt, dev, tune=t, prescaled=True)),
                ("spike", lambda: kc.check_attn_spike(dt, dev, tune=t, prescaled=True)),
                ("neg", lambda: kc.check_attn_neg(dt, dev, tune=t, prescaled=True)),
                ("N3072_h10", lambda: kc.check_attn_self(4, 10, 3072, dt, dev, n_garm=3072, b0=2, tune=t, prescaled=True)),
            ]
            for cn, fn in checks:
                try:
                    e = fn()
                except Exception as ex:
                    e = float("nan"); print("   EXC", str(ex)[:120])
                tol = 2e-2 if dt == torch.bfloat16 else 4e-3
                ok = e == e and e < tol
                bad += not ok
                print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]:9s} {name} thr{thr} {cn:14s} {e:.2e}", flush=True)
print("FAILURES:", bad, flush=True)

g = torch.Generator(device="cpu").manual_seed(0)
def timed(fn, n=60):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for dt in (torch.bfloat16, torch.float16):
    r = lambda *s, sc=float(os.environ.get('SC', '0.35')): (torch.randn(*s, generator=g) * sc).to(dev, dt)
    for (B, heads, N, b0) in ((4, 10, 3072, 2), (4, 20, 768, 2), (12, 10, 3072, 12), (12, 20, 768, 12), (2, 10, 6144, 1), (2, 20, 1536, 1)):
        C = heads * 64
        qk = r(B * N, 2 * C)
        vt = r(B, C, N, sc=1.0)
        segs = [dict(k=qk[:, C:], vt=vt, nk=N, ldk=2 * C, ldvt=N)]
        if b0 < B:
            Bg = B - b0
            segs.append(dict(k=r(Bg * N, C), vt=r(Bg, C, N, sc=1.0), nk=N, ldk=C, ldvt=N, b0=b0))
        out = torch.empty(B * N, C, dtype=dt, device=dev)
        keys = sum(N * (B - s.get("b0", 0)) for s in segs)
        fl = 4.0 * N * 64 * heads * keys
        line = []
        for rep in range(2):
            for name, tn in VARIANTS + TIMING_ONLY:
                t = timed(lambda: ops.attention(qk, out, segs, heads, B=B, Nq=N, ldq=2 * C, ldo=C, q_prescaled=True, tune=tn))
                if rep == 1:
                    line.append(f"{name}={t:.1f}us/{fl / t / 1e6:.0f}TF")
        print(f"{str(dt)[6:]} B={B} h={heads} N={N} b0={b0}: " + "  ".join(line), flush=True)
