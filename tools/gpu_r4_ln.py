"""Round 4 probe: LayerNorm launch time on the loop's shapes (rows per wave selected by IDMVTON_LN_RPW = 1 | 2, read once per process).
warm = 20 back-to-back launches on the same operands (L2 / Infinity Cache resident, as behind the GEMM that wrote them)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from idm_vton_amd import ops  # noqa: E402


def main():
    dt, dev = torch.bfloat16, "cuda"
    out = [f"IDMVTON_LN_RPW={os.environ.get('IDMVTON_LN_RPW', 'auto')}"]
    for rows, C in ((3072, 1280), (9216, 1280), (12288, 640), (36864, 640), (1536, 1280)):
        x = (torch.randn(rows, C, device=dev) * 2).to(dt)
        g, b = torch.randn(C, device=dev).to(dt), torch.randn(C, device=dev).to(dt)
        y = torch.empty_like(x)
        fn = lambda: ops.layernorm(x, g, b, 1e-5, out=y)
        fn(); fn()
        v = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            v.append(e0.elapsed_time(e1) * 1e3 / 20)
        us = sorted(v)[4]
        out.append(f"{rows}x{C}: {us:.2f} us = {2 * rows * C * 2 / us / 1e6:.2f} TB/s")
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
