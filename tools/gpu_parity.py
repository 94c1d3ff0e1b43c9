"""Model-level parity report (HIP engine vs oracle) -> gpurun_out/parity.json.  python tools/gpu_parity.py [tiny|mid|all]"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import parity_checks  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    runs = [("tiny", torch.float16, False), ("tiny", torch.bfloat16, False), ("tiny", torch.float16, True)]
    if which in ("mid", "all"):
        runs += [("mid", torch.float16, False), ("mid", torch.bfloat16, False)]
    if which == "tiny":
        runs = runs[:3]
    out = {}
    for kind, dt, graph in runs:
        tag = f"{kind}/{'f16' if dt == torch.float16 else 'bf16'}{'/graph' if graph else ''}"
        t0 = time.time()
        try:
            r = parity_checks.run(kind, dt, B=1 if kind == "tiny" else 1, H=128, W=128, steps=4, use_graph=graph)
        except Exception as e:  # noqa: BLE001
            traceback.print_exc()
            r = {"EXC": repr(e)[:400]}
        out[tag] = r
        print(f"== {tag}  ({time.time() - t0:.1f}s)")
        for k, v in r.items():
            print(f"   {k:32s} {v if isinstance(v, str) else format(v, '.3e')}")
        sys.stdout.flush()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
