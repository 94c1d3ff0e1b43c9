#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for mode in "" "--no-graph" "--no-overlap" "--no-overlap --no-graph"; do
  tag=$(echo "default$mode" | tr -d ' ')
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline $mode > $O/r2q_bench_$tag.json 2> $O/r2q_bench_$tag.err; echo "$tag rc=$? $(python -c "
import json; d=json.loads(open('$O/r2q_bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
bash tools/gpu_pmc_traffic.sh "round 2 final kernels: 16-byte epilogue, GarmentNet batched over 6 timesteps (rocprofv3 --pmc over the serial eager bench command, 6 denoising steps = one block)"
