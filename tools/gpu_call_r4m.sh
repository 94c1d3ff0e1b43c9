#!/bin/bash
# round 4, call M: same-box A/B of the round-3 final state (a build of commit 78abdc5 in an untracked worktree) and the current state
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
  (cd $R/.ab_r03 && timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r4m_bench_r03_$i.json 2>> $O/r4m.err)
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp16-leg > $O/r4m_bench_r04_$i.json 2>> $O/r4m.err
done
python - <<'PY'
import json
for tag in ("r03_1", "r04_1", "r03_2", "r04_2"):
    d = json.loads(open(f"gpurun_out/r4m_bench_{tag}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(tag, round(d["value"], 4), round(d["ms_per_step"], 1), d.get("call_parts_ms"), d.get("loop_ms_per_denoise_step"), round(r["frac"], 4), r.get("frac_in_loop"), r["step_kernel_ms"])
PY
