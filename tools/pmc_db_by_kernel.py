"""rocprofv3 --pmc rocpd databases (*_results.db) -> table per (kernel, grid): mean of every counter over its dispatches, plus the
mean duration when --kernel-trace was on.   python tools/pmc_db_by_kernel.py <dir-or-db> [...] [--match substr] [--split-duration]
--split-duration: persistent kernels launch the same grid for every shape; add the power-of-two bucket of the dispatch duration to the key so that
different shapes of one kernel get their own rows."""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"_Z\d+([a-z_0-9]+?)I(.*?)Ev", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    return re.sub(r"^void ", "", name)[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--") and (sys.argv[sys.argv.index(a) - 1] != "--match")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    split = "--split-duration" in sys.argv
    dbs = []
    for a in args:
        dbs += [a] if a.endswith(".db") else glob.glob(os.path.join(a, "**", "*_results.db"), recursive=True)
    agg = {}
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
        q = "select kernel_name, grid_size, workgroup_size, counter_name, value, start, end from counters_collection"
        if "grid_size" not in cols:
            q = "select kernel_name, grid_size_x, workgroup_size_x, counter_name, value, start, end from counters_collection"
        for name, grid, wg, cn, val, t0, t1 in c.execute(q):
            if match and match not in name:
                continue
            key = (short(name) + (f" ~2^{max(t1 - t0, 1).bit_length()}ns" if split else ""), int(grid) // max(int(wg), 1))
            a = agg.setdefault(key, {})
            e = a.setdefault(cn, [0, 0.0])
            e[0] += 1
            e[1] += float(val)
            d = a.setdefault("dur_us", [0, 0.0])
            d[0] += 1
            d[1] += (t1 - t0) / 1e3
    names = sorted({c for v in agg.values() for c in v})
    print("kernel".ljust(64) + " blocks " + " ".join(n[:20].rjust(20) for n in names))
    for (k, g), v in sorted(agg.items()):
        print(k[:64].ljust(64) + f" {g:6d} " + " ".join((f"{v[n][1] / v[n][0]:20.1f}" if n in v else " " * 20) for n in names))


if __name__ == "__main__":
    main()
