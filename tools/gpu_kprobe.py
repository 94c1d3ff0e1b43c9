"""Kernel probe for rocprofv3 PMC passes: launches a fixed list of (kernel configuration x real config-2 shape) a few times
each, so per-dispatch counters (MFMA busy, wait cycles, LDS conflicts, L2 hit/miss, FETCH/WRITE bytes) can be read per
kernel + grid from the counter CSV.  python tools/gpu_kprobe.py [reps]   (run under rocprofv3 --pmc ... --kernel-trace)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ffi, ops  # noqa: E402
from idm_vton_amd.weights import interleave_geglu  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)
    jobs = []
    # GEGLU ff1 L2 (TryonNet): 3072 x 10240 x 1280
    x, w, b = r(3072, 1280), r(10240, 1280), r(10240)
    wi, bi = interleave_geglu(w, b)
    for h in (hint(1, 256, 256), hint(1, 128, 256), hint(0, 128, 128)):
        jobs.append(lambda h=h, x=x, wi=wi, bi=bi: ops.linear(x, wi, bias=bi, geglu=True, tile_hint=h))
    # proj L2: 3072 x 1280 x 1280 ; 1536 x 1280 x 1280
    for M in (3072, 1536):
        x2, w2, b2, rs = r(M, 1280), r(1280, 1280), r(1280), r(M, 1280)
        for h in (hint(1, 128, 128), hint(1, 128, 64), hint(0, 128, 64)):
            jobs.append(lambda h=h, x2=x2, w2=w2, b2=b2, rs=rs: ops.linear(x2, w2, bias=b2, res=rs, tile_hint=h))
    # ff2 L2: 3072 x 1280 x 5120
    x3, w3, b3, rs3 = r(3072, 5120), r(1280, 5120), r(1280), r(3072, 1280)
    for h in (hint(1, 128, 128), hint(1, 128, 256)):
        jobs.append(lambda h=h: ops.linear(x3, w3, bias=b3, res=rs3, tile_hint=h))
    # self attention L2 (TryonNet: B4 h20 N768 + garment 768 for the cond half) and L1
    for (B, hd, N) in ((4, 20, 768), (4, 10, 3072)):
        C = hd * 64
        q, k, v = r(B, N, C), r(B, N, C), r(B, C, N)
        kg, vg = r(B // 2, N, C), r(B // 2, C, N)
        out = torch.empty(B, N, C, dtype=dt, device=dev)
        segs = [dict(k=k, vt=v, nk=N, ldk=C, ldvt=N), dict(k=kg, vt=vg, nk=N, ldk=C, ldvt=N, b0=B // 2)]
        for tn in ((2 << 8) | 8, (3 << 8) | 4, (2 << 24) | (2 << 8) | 8, (2 << 24) | (2 << 8) | 4):
            jobs.append(lambda tn=tn, q=q, out=out, segs=segs, hd=hd: ops.attention(q, out, segs, hd, tune=tn))
    for j in jobs:
        for _ in range(reps):
            flush.zero_()
            j()
    torch.cuda.synchronize()
    print("kprobe done:", len(jobs), "jobs x", reps)


if __name__ == "__main__":
    main()
