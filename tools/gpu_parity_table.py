#!/usr/bin/env python
"""Run every full-size parity leg (tests/fullsize_parity.py) outside pytest and write the JSON behind DESIGN.md section 5's table.
  gpurun -- python tools/gpu_parity_table.py [stage ...]        -> gpurun_out/fullsize_parity.json"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import fullsize_parity as fp  # noqa: E402

if __name__ == "__main__":
    out = os.path.join(ROOT, "gpurun_out", "fullsize_parity.json")
    W = fp.World(torch.device("cuda", 0))
    kw = dict(stages=tuple(sys.argv[1:])) if len(sys.argv) > 1 else {}
    fp.run_all(W, out, **kw)
    W.dump(out)
    print("wrote", out)
