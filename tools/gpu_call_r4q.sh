#!/bin/bash
# Round 4, call Q: e4m3 operands written by the projections themselves (IDMVTON_IO_OUT_F8): kernel checks, the fp8 model / DressCode
# tests, then fp16 vs fp16 + fp8 (fused / two-launch) bench lines on one box, alternating.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/r4q_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r4q_build.log; exit 1; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "f8 or vt or qkv or colscale" --tb=short 2>&1 | tail -15 | cut -c1-250
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_dresscode_gpu.py tests/test_fullsize_gpu.py -q -p no:cacheprovider -k "fp8 or f8" --tb=short 2>&1 | tail -15 | cut -c1-250
for r in 1 2; do
  timeout 600 python bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r4q_f16_$r.json 2>/dev/null; cut -c1-150 $O/r4q_f16_$r.json
  timeout 600 python bench.py --dtype f16 --attn-fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r4q_f16_fp8_fused_$r.json 2>/dev/null; cut -c1-150 $O/r4q_f16_fp8_fused_$r.json
  IDMVTON_F8_FUSED=0 timeout 600 python bench.py --dtype f16 --attn-fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r4q_f16_fp8_twolaunch_$r.json 2>/dev/null; cut -c1-150 $O/r4q_f16_fp8_twolaunch_$r.json
done
