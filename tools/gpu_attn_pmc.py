"""Attention kernel variants on the TryonNet L1 shape, a few launches each, for rocprofv3 --pmc passes (tools/pmc_by_kernel.py
turns the counter CSV into a per-kernel table).  python tools/gpu_attn_pmc.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402
from tests.kernel_checks import pp_tune  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: torch.randn(*s, device=dev).to(dt)
    B, heads, N, b0 = 4, 10, 3072, 2
    C = heads * 64
    q, k1, v1, k2, v2 = r(B, N, C), r(B, N, C), r(B, C, N), r(B - b0, N, C), r(B - b0, C, N)
    out = torch.empty(B, N, C, dtype=dt, device=dev)
    segs = [dict(k=k1, vt=v1, nk=N, ldk=C, ldvt=N), dict(k=k2, vt=v2, nk=N, ldk=C, ldvt=N, b0=b0)]
    tunes = [(2 << 8) | 8, pp_tune(2, 0), pp_tune(2, 1), pp_tune(3, 1), (4 << 16) | (2 << 8) | 8, (5 << 16) | (2 << 8) | 8, (6 << 16) | (2 << 8) | 8]
    for tn in tunes:
        for _ in range(reps):
            ops.attention(q, out, segs, heads, tune=tn)
    torch.cuda.synchronize()
    print("done", len(tunes), "variants x", reps)


if __name__ == "__main__":
    main()
