"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) into a short text table (names truncated).
Usage: python tools/rocpd_summary.py <results.db> [--by-grid] [--match substr]  > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"_Z\d+([a-z_0-9]+?)I(.*?)Ev", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, vg, ag, sg, lds in rows:
        if match and match not in name:
            continue
        key = (short(name), gx // max(wx, 1), wx, vg, ag, lds) if by_grid else (short(name),)
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"# {db}: {len(rows)} dispatches, total kernel time {tot / 1e6:.3f} ms")
    hdr = "kernel".ljust(72) + (" blocks  wg vgpr agpr    lds" if by_grid else "") + "    calls   total_ms    avg_us    min_us    max_us     %"
    print(hdr)
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        line = key[0].ljust(72)
        if by_grid:
            line += f" {key[1]:6d} {key[2]:3d} {key[3]:4d} {key[4]:4d} {key[5]:6d}"
        line += f" {a[0]:8d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:5.1f}"
        print(line)


if __name__ == "__main__":
    main()
