#!/bin/bash
# Round 4, call U: where fp16's deficit goes: clock / power / temperature sampled beside the bf16 and the fp16 engine,
# alternating twice on one box; then the per-family kernel time of both from the serial replica (same box).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4u_build.log 2>&1 || { echo BUILD FAILED; exit 1; }
for r in 1 2; do
  bash tools/gpu_power_trace.sh r4u_power_bf16_$r "" "" | tail -4
  bash tools/gpu_power_trace.sh r4u_power_f16_$r "" "--dtype f16" | tail -4
done
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-leg > $O/r4u_roofline_bf16.json 2>/dev/null
timeout 400 python bench.py --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline --no-fp16-leg > $O/r4u_roofline_f16.json 2>/dev/null
python - <<'PY'
import json
for n in ("bf16","f16"):
    d=json.loads(open(f"gpurun_out/r4u_roofline_{n}.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print(n, round(d["value"],4), "images/s  frac", round(r["frac"],4), "step_kernel_ms", r["step_kernel_ms"], "attn TF", round(r["attn_fwd"]["achieved"],1))
PY
