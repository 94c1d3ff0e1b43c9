"""Round 4 probe: attn2 as two launches (to_q GEMM + cross-attention kernel) vs the fused form (csrc/xattn.cuh) on the loop's real shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ffi, ops  # noqa: E402


def hint(v, bn, bm):
    return (v << 28) | (bn << 16) | bm


def main():
    ops.load_tune()
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s, scale=0.5: (torch.randn(*s, device=dev) * scale).to(dt)
    flush = torch.empty(640 << 20, dtype=torch.uint8, device=dev)

    def timeit(fn):
        fn(); fn()
        out = {}
        for mode in ("warm", "cold"):
            v = []
            for _ in range(7):
                if mode == "cold":
                    flush.zero_()
                n = 10 if mode == "warm" else 1
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record(); e1.synchronize()
                v.append(e0.elapsed_time(e1) * 1e3 / n)
            out[mode] = sorted(v)[3]
        return out

    for name, B, heads, N, nip in (("TryonNet L2 (M=3072, C=1280)", 4, 20, 768, 16), ("TryonNet L1 (M=12288, C=640)", 4, 10, 3072, 16),
                                   ("GarmentNet L2 x6 (M=9216, C=1280)", 12, 20, 768, 0), ("GarmentNet L1 x6 (M=36864, C=640)", 12, 10, 3072, 0)):
        C = heads * 64
        M = B * N
        x, wq = r(M, C), r(C, C, scale=C ** -0.5)
        segs = []
        for nk in [77] + ([nip] if nip else []):
            rows = (nk + 31) // 32 * 32
            segs.append(dict(k=r(B, rows, C), vt=r(B, C, rows), nk=nk, ldk=C, ldvt=rows, k_rows=rows))
        wx = ops.xattn_q_weight(wq)
        q2, att = torch.empty(M, C, dtype=dt, device=dev), torch.empty(M, C, dtype=dt, device=dev)

        def two():
            ops.linear(x, wq, out=q2)
            if nip:
                ops.attention(q2, att, segs, heads, mode=ffi.ATTN_CROSS, B=B, Nq=N, ldq=C, ldo=C)
            else:
                ops.attention(q2, att, segs, heads, B=B, Nq=N, ldq=C, ldo=C)
        row = [f"two launches {timeit(two)}", f"to_q alone {timeit(lambda: ops.linear(x, wq, out=q2))}"]
        for tag, h in (("f128x64", hint(1, 128, 64)), ("f128x128", hint(1, 128, 128)), ("f128x256", hint(1, 128, 256))):
            row.append(f"{tag} {timeit(lambda: ops.linear(x, wx, out=att, xattn=dict(segs=segs, tokens=N), tile_hint=h))}")
        print(name + "\n    " + "\n    ".join(row), flush=True)


if __name__ == "__main__":
    main()
