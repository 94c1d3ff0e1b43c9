"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) into a short text table (names truncated).
Usage: python tools/rocpd_summary.py <results.db> [--by-grid] [--match substr]  > profiles/<name>.txt
       python tools/rocpd_summary.py <results.db> --timeline [--steps-per-call 30] : GPU busy / idle inside the denoising loop of the
       LAST pipeline call (trace `bench.py --no-roofline` so that this is a timed hipGraph call): span, union of kernel
       intervals, sum of durations, idle gaps by size"""
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


def timeline(db, steps_per_call=30):
    """Busy / idle of the LAST pipeline call's denoising loop: the window from the pack_input of its first step to the end of its
    last cfg_step (`steps_per_call` cfg_step kernels back from the end).  Run the traced command with --no-roofline so that the last
    call is a TIMED call of the benchmarked form (hipGraph replay, two streams), not the instrumented eager replica of the roofline
    leg (whose per-launch event records open a 5-10 us gap after every kernel)."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    packs = [i for i, r in enumerate(rows) if "pack_input_kernel" in r[0]]
    cfgs = [i for i, r in enumerate(rows) if "cfg_step_kernel" in r[0]]
    if len(cfgs) < steps_per_call or not packs:
        print("no denoising window found"); return
    hi = cfgs[-1]
    first_cfg = cfgs[-steps_per_call]
    lo = max(p for p in packs if p < first_cfg)
    nsteps = steps_per_call
    t0, t1 = rows[lo][1], rows[hi][2]
    win = [r for r in rows[lo:] if r[1] < t1]            # includes the side stream's kernels that start inside the window
    busy, cur_s, cur_e, gaps = 0, win[0][1], win[0][2], []
    for _, s_, e_ in win[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            gaps.append(s_ - cur_e)
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += min(cur_e, t1) - cur_s
    span, sumdur = t1 - t0, sum(min(r[2], t1) - r[1] for r in win)
    fam = {}
    for n, s_, e_ in win:
        k = short(n).split("<")[0]
        fam[k] = fam.get(k, 0) + (min(e_, t1) - s_)
    print(f"# {db}")
    print(f"denoising loop of the last call: {nsteps} steps, {len(win)} kernels, span {span / 1e6:.2f} ms = {span / 1e6 / nsteps:.2f} ms/step")
    print(f"  GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms = {100 * busy / span:.1f} % of span; idle {100 * (span - busy) / span:.1f} %")
    print(f"  sum of kernel durations {sumdur / 1e6:.2f} ms = {sumdur / busy:.2f} x busy time (> 1: kernels of the two streams overlap)")
    for lo_us, hi_us in ((0, 2), (2, 5), (5, 10), (10, 50), (50, 1e9)):
        g = [x for x in gaps if lo_us * 1e3 <= x < hi_us * 1e3]
        print(f"  idle gaps {lo_us:>3}-{hi_us if hi_us < 1e9 else 'inf':>3} us: {len(g):6d} gaps, {sum(g) / 1e6:8.3f} ms")
    # what one pipeline call spends OUTSIDE the loop: from the end of the previous call's last cfg_step to this call's first pack_input
    # = decode + postprocess of call i-1, then preprocessing + 3 VAE encodes + Resampler + step-invariant K/V and embedding tables of call i
    if len(cfgs) >= steps_per_call + 1:
        prev_end = rows[cfgs[-steps_per_call - 1]][2]
        ow = [r for r in rows if r[1] >= prev_end and r[2] <= t0]
        if ow:
            ofam = {}
            for n, s_, e_ in ow:
                k = short(n).split("<")[0]
                ofam[k] = ofam.get(k, 0) + (e_ - s_)
            print(f"outside the loop (decode of the previous call + prepare of this one): span {(t0 - prev_end) / 1e6:.2f} ms, {len(ow)} kernels, "
                  f"kernel time {sum(ofam.values()) / 1e6:.2f} ms: " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in sorted(ofam.items(), key=lambda kv: -kv[1])[:10]))
    print("  kernel time by family inside the window (ms per step): " + ", ".join(f"{k} {v / 1e6 / nsteps:.2f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8]))


def main():
    db = sys.argv[1]
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--steps-per-call") + 1]) if "--steps-per-call" in sys.argv else 30
        return timeline(db, n)
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
