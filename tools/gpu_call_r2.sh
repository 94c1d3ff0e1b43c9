#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_dropin_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
for mode in "" "--no-graph" "--no-overlap"; do
  tag=$(echo "default$mode" | tr -d ' ')
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $mode > $O/r2r_bench_$tag.json 2> $O/r2r_bench_$tag.err; echo "$tag rc=$? $(python -c "
import json; d=json.loads(open('$O/r2r_bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
