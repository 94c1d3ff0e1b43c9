"""Large-GEMM launches for rocprofv3 --pmc passes (tools/pmc_by_kernel.py tabulates per kernel + grid): the 8192^3 and ff1 shapes on
the 256x256 kernels, plus hipBLASLt (yardstick) on the same operands.  python tools/gpu_gemm_pmc.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402

H = lambda v, bn, bm: (v << 28) | (bn << 16) | bm


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ops.load_tune(None)
    dt, dev = torch.bfloat16, "cuda"
    r = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).to(dt)
    for M, N, K in ((8192, 8192, 8192), (3072, 10240, 1280)):
        x, w = r(M, K, scale=0.5), r(N, K, scale=0.03)
        o = torch.empty(M, N, dtype=dt, device=dev)
        wt = w.t()
        for _ in range(reps):
            ops.linear(x, w, out=o, tile_hint=H(1, 256, 256))
            ops.linear(x, w, out=o, tile_hint=H(1, 128, 256))
            ops.linear(x, w, out=o, tile_hint=H(5, 256, 257))       # round 4: the hand-scheduled persistent loop (gemm_lin_kernel)
            torch.matmul(x, wt, out=o)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
