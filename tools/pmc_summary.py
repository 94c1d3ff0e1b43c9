"""Summarise rocprofv3 --pmc CSV output (*counter_collection.csv) per kernel: dispatches, mean counter value.
  python tools/pmc_summary.py <dir-with-csvs> [<dir2> ...] -> JSON on stdout
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so HBM traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r"_Z\d+([a-z_0-9]+?)I(.*?)Ev", name)
    if m:
        return m.group(1)
    name = re.sub(r"^void ", "", name)
    return re.split(r"[<(]", name)[0][:60]


GEMM_FAMILIES = ("gemm_conv_kernel", "gemm_lin_kernel", "gemm_xattn_kernel")   # every kernel idmvton_gemm_conv launches


def summarise(dirs, label=None):
    """Two aggregates: every dispatch of the run, and (key `denoise_step`) only the dispatches between the first
    pack_input_kernel and the last cfg_step_kernel of the (serial, eager) run = the launches of the denoising steps, the
    set bench.py's roofline leg covers.  (Also imported by bench.py's own PMC leg.)"""
    agg, step = {}, {}
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) + \
            glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        for f in files:
            if f.endswith(".db"):                          # rocpd database (rocprofv3's default output format)
                import sqlite3
                c = sqlite3.connect(f)
                rows = sorted(((int(did), short(kn), cn, float(v)) for did, kn, cn, v in
                               c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection")), key=lambda t: t[0])
            else:
                with open(f, newline="") as fh:
                    rd = csv.DictReader(fh)
                    cols = {c.lower(): c for c in rd.fieldnames}
                    kn, cn, cv, di = cols["kernel_name"], cols["counter_name"], cols["counter_value"], cols["dispatch_id"]
                    rows = sorted(((int(r[di]), short(r[kn]), r[cn], float(r[cv])) for r in rd), key=lambda t: t[0])
            lo = min((t[0] for t in rows if t[1] == "pack_input_kernel"), default=None)
            hi = max((t[0] for t in rows if t[1] == "cfg_step_kernel"), default=None)
            for did, k, c, v in rows:
                a = agg.setdefault(k, {}).setdefault(c, [0, 0.0])
                a[0] += 1
                a[1] += v
                if lo is not None and hi is not None and lo <= did <= hi:
                    a = step.setdefault(k, {}).setdefault(c, [0, 0.0])
                    a[0] += 1
                    a[1] += v
    out = {}
    for k, cs in agg.items():
        e = {c: dict(dispatches=n, mean=s / n) for c, (n, s) in cs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
        out[k] = e
    res = {"per_kernel": out, "denoise_step": {}}
    for k, cs in step.items():
        e = {c: dict(dispatches=n, mean=s_ / n) for c, (n, s_) in cs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
        res["denoise_step"][k] = e
    # the C-ABI entry idmvton_gemm_conv launches three kernel families since round 4: the compiler-scheduled tiles, the hand-scheduled Linear
    # loop (gemm_lin_kernel) and the projection with the fused cross-attention epilogue (gemm_xattn_kernel): one launch-weighted mean over all
    fams = [res["denoise_step"][k] for k in GEMM_FAMILIES if "hbm_bytes_per_launch" in res["denoise_step"].get(k, {})]
    if fams:
        n = sum(f["FETCH_SIZE"]["dispatches"] for f in fams)
        res["gemm_conv_bytes_per_launch"] = sum(f["hbm_bytes_per_launch"] * f["FETCH_SIZE"]["dispatches"] for f in fams) / n
        res["gemm_conv_launches_counted"] = n
    if label:
        res["source"] = label
    return res


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--") and sys.argv[sys.argv.index(a) - 1] != "--label"]
    label = sys.argv[sys.argv.index("--label") + 1] if "--label" in sys.argv else None
    json.dump(summarise(dirs, label), sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
