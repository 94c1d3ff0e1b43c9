#!/bin/bash
# shader clock / socket power sampled every 0.25 s while bench.py runs: the evidence behind "the loop is power-limited" (DESIGN §6).
# bash tools/gpu_power_trace.sh [tag] [perf level | ""] [extra bench.py arguments, e.g. "--dtype f16"]
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-power}
LEVEL=${2:-}                                    # optional: a rocm-smi performance level to try for the run (auto afterwards)
EXTRA=${3:-}
[ -n "$LEVEL" ] && { rocm-smi --setperflevel $LEVEL 2>&1 | grep -v "^$" | head -5; }
rocm-smi --showperflevel --showmaxpower --showpower --showclocks > $O/${TAG}_smi_idle.txt 2>&1
( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $O/${TAG}_smi_samples.jsonl &
SMI=$!
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-fp16-leg $EXTRA 2>/dev/null | tail -1 > $O/${TAG}_bench.json
kill $SMI
[ -n "$LEVEL" ] && rocm-smi --setperflevel auto > /dev/null 2>&1
python - "$O/${TAG}_smi_samples.jsonl" "$O/${TAG}_bench.json" <<'PY' | tee $O/${TAG}_summary.txt
import json, sys, re
rows = []
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get("card0", {})
    def num(pat):
        for k, v in c.items():
            if re.search(pat, k, re.I):
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                if m:
                    return float(m.group(0))
        return None
    rows.append((num(r"sclk clock speed|sclk"), num(r"power"), num(r"temperature.*(junction|hotspot)|temperature")))
rows = [r for r in rows if r[0] is not None]
print("samples", len(rows))
if rows:
    busy = [r for r in rows if r[1] is not None and r[1] > 0.5 * max(x[1] for x in rows if x[1] is not None)]
    f = lambda xs: (min(xs), sum(xs) / len(xs), max(xs)) if xs else None
    print("all samples   sclk MHz (min, mean, max):", f([r[0] for r in rows]), " power W:", f([r[1] for r in rows if r[1] is not None]))
    print("under load    sclk MHz (min, mean, max):", f([r[0] for r in busy]), " power W:", f([r[1] for r in busy]), " temp C:", f([r[2] for r in busy if r[2] is not None]))
b = json.load(open(sys.argv[2]))
print("bench", round(b["value"], 4), "images/s", round(b["ms_per_step"], 1), "ms per call")
PY
head -40 $O/${TAG}_smi_idle.txt
