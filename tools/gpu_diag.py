"""Run every kernel-level parity check on the GPU and print/record ALL results (does not stop at the first failure).
Usage (on the GPU box):  python tools/gpu_diag.py [substring-filter]   -> gpurun_out/diag.json"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import kernel_checks as kc  # noqa: E402


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    print("device:", torch.cuda.get_device_name(0), "| cpu threads:", torch.get_num_threads(), flush=True)
    res = []
    for name, fn, tol in kc.all_checks():
        if flt and flt not in name:
            continue
        t0 = time.time()
        try:
            err = fn()
            torch.cuda.synchronize()
            status = "ok" if err <= tol else "FAIL"
        except Exception as e:  # noqa: BLE001
            err, status = None, "EXC: " + repr(e)[:300]
            traceback.print_exc()
        res.append(dict(name=name, err=err, tol=tol, status=status, sec=round(time.time() - t0, 3)))
        print(f"{status:5s} {name:45s} err={err if err is None else format(err, '.3e')} tol={tol:.1e}", flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
            json.dump(res, f, indent=1)
    bad = [r for r in res if r["status"] != "ok"]
    print(f"{len(res) - len(bad)}/{len(res)} ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
