#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
IDMVTON_EPILOGUE_8B=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2l_bench_8B_$i.json 2> $O/r2l_bench_8B.err; echo "bench(8B) rc=$?"; python -c "
import json,sys; d=json.loads(open('$O/r2l_bench_8B_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['step_kernel_ms'])"
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2l_bench_16B_$i.json 2> $O/r2l_bench_16B.err; echo "bench(16B) rc=$?"; python -c "
import json,sys; d=json.loads(open('$O/r2l_bench_16B_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['step_kernel_ms'])"
done
