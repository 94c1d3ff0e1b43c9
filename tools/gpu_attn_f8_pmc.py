"""The bf16 ping-pong attention kernel and the fp8 attention kernel on the TryonNet L1 shape (B4 h10, 3072 q x 3072 + 3072 garment keys,
CFG), a few launches each, for one rocprofv3 --pmc pass (tools/pmc_db_by_kernel.py makes the per-kernel table)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402


def main():
    reps = 4
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s: torch.randn(*s, device=dev).to(dt)
    B, heads, N, b0 = 4, 10, 3072, 2
    C = heads * 64
    q, k1, v1, k2, v2 = r(B, N, C), r(B, N, C), r(B, C, N), r(B - b0, N, C), r(B - b0, C, N)
    out = torch.empty(B, N, C, dtype=dt, device=dev)
    segs = [dict(k=k1, vt=v1, nk=N, ldk=C, ldvt=N), dict(k=k2, vt=v2, nk=N, ldk=C, ldvt=N, b0=b0)]
    q8 = ops.quant_f8(q.view(B * N, C), 4.0)
    segs8 = []
    for (k, v, bb) in ((k1, v1, 0), (k2, v2, b0)):
        Bs = k.shape[0]
        segs8.append(dict(k8=ops.quant_f8(k.view(Bs * N, C), 4.0), vt8=ops.quant_f8(v.view(Bs * C, N), 4.0, mode=1), nk=N, ldk=C, ldvt=N, b0=bb))
    for _ in range(reps):
        ops.attention(q, out, segs, heads, tune=(2 << 16) | (2 << 8) | 8, q_prescaled=True)
        ops.attention_f8(q8, out, segs8, heads, qk_scale_exp=-4, v_scale_exp=-2, B=B, Nq=N, ldq=C, ldo=C)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
