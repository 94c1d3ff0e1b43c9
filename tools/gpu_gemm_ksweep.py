"""K sweep (one gpurun call): time(K) for fixed M x N splits a GEMM launch into its fixed part (launch, pipeline fill, epilogue) and
its per-K-tile part.  ours (several tiles) vs hipBLASLt (torch.matmul, yardstick only).  -> gpurun_out/r2_ksweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402

DT, DEV = torch.bfloat16, "cuda"
rnd = lambda *s, scale=1.0: (torch.randn(*s, device=DEV) * scale).to(DT)
H = lambda v, bn, bm: (v << 28) | (bn << 16) | bm


def timeit(fn, rounds=9, inner=20):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    res = {}
    NARROW = 0x8000
    for M, N, tiles in ((3072, 10240, (("r256x256", H(1, 256, 256)))),
                        (8192, 8192, (("r256x256", H(1, 256, 256))))):
        for K in (64, 640, 1280, 2560, 8192):
            x, w = rnd(M, K, scale=0.5), rnd(N, K, scale=0.03)
            o = torch.empty(M, N, dtype=DT, device=DEV)
            wt = w.t()
            r = {tag: round(timeit(lambda: ops.linear(x, w, out=o, tile_hint=h)), 1) for tag, h in tiles}
            r["hipblaslt"] = round(timeit(lambda: torch.matmul(x, wt, out=o)), 1)
            res[f"{M}x{N}x{K}"] = r
            print(f"{M}x{N}x{K:5d}  " + "  ".join(f"{k} {v:7.1f}" for k, v in r.items()), flush=True)
        # a pure store kernel of the same output volume, as a floor for the epilogue
        o = torch.empty(M, N, dtype=DT, device=DEV)
        res[f"{M}x{N} fill"] = round(timeit(lambda: o.fill_(1.0)), 1)
        print(f"{M}x{N} fill_ {res[f'{M}x{N} fill']} us", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r2_ksweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
