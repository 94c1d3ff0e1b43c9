"""rocprofv3 --pmc CSV (*counter_collection.csv) -> table per (kernel, grid): mean of every counter over its dispatches.
  python tools/pmc_by_kernel.py <dir> [<dir> ...] [--match substr]"""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.match(r"_Z\d+([a-z_0-9]+?)I(.*?)Ev", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    return re.sub(r"^void ", "", name)[:60]


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    agg = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                rd = csv.DictReader(fh)
                cols = {c.lower(): c for c in rd.fieldnames}
                for row in rd:
                    name = row[cols["kernel_name"]]
                    if match and match not in name:
                        continue
                    key = (short(name), int(row[cols["grid_size"]]) // max(int(row[cols["workgroup_size"]]), 1))
                    a = agg.setdefault(key, {}).setdefault(row[cols["counter_name"]], [0, 0.0])
                    a[0] += 1
                    a[1] += float(row[cols["counter_value"]])
    names = sorted({c for v in agg.values() for c in v})
    print("kernel".ljust(64) + " blocks " + " ".join(n[:18].rjust(18) for n in names))
    for (k, g), v in sorted(agg.items()):
        print(k[:64].ljust(64) + f" {g:6d} " + " ".join((f"{v[n][1] / v[n][0]:18.1f}" if n in v else " " * 18) for n in names))


if __name__ == "__main__":
    main()
