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


def main():
    agg = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                rd = csv.DictReader(fh)
                cols = {c.lower(): c for c in rd.fieldnames}
                kn, cn, cv = cols["kernel_name"], cols["counter_name"], cols["counter_value"]
                for row in rd:
                    a = agg.setdefault(short(row[kn]), {}).setdefault(row[cn], [0, 0.0])
                    a[0] += 1
                    a[1] += float(row[cv])
    out = {}
    for k, cs in agg.items():
        e = {c: dict(dispatches=n, mean=s / n) for c, (n, s) in cs.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024.0
        out[k] = e
    res = {"per_kernel": out}
    if "gemm_conv_kernel" in out and "hbm_bytes_per_launch" in out["gemm_conv_kernel"]:
        res["gemm_conv_bytes_per_launch"] = out["gemm_conv_kernel"]["hbm_bytes_per_launch"]
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
