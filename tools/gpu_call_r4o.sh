#!/bin/bash
# round 4, call O: round-end evidence on the frozen state (tests, smoke, bench line, other configurations, rocprofv3 kernel traces) + PMC traffic
cd "$(dirname "$0")/.." || exit 1
bash tools/gpu_final.sh r04_final
bash tools/gpu_pmc_traffic.sh "round 4 kernels (hand-scheduled Linear tiles, fused cross-attention, split-precision VAE decode): rocprofv3 --pmc over the serial eager bench command, one uniform 6-timestep block"
timeout 600 python bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r04_final_bench_f16.json 2>/dev/null; cut -c1-160 gpurun_out/r04_final_bench_f16.json
timeout 600 python bench.py --dtype f16 --attn-fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r04_final_bench_f16_fp8.json 2>/dev/null; cut -c1-160 gpurun_out/r04_final_bench_f16_fp8.json
