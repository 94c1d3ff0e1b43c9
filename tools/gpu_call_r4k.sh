#!/bin/bash
# round 4, call K: re-tune with the 256x192 tile and the fused cross-attention signatures, then the bench line and the whole GPU suite on the result
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "xattn" > $O/r4k_kchecks.log 2>&1; tail -3 $O/r4k_kchecks.log | cut -c1-300
timeout 1500 python tools/gpu_tune.py > $O/r4k_tune.log 2>&1; tail -3 $O/r4k_tune.log
cp $O/tune_gfx950.json $R/idm-vton_amd/tune_gfx950.json
timeout 900 python bench.py --steps 5 --warmup 2 > $O/r4k_bench.json 2> $O/r4k_bench.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4k_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "call_parts_ms", "loop_ms_per_denoise_step", "fp16")}, {k: d["roofline"][k] for k in ("frac", "frac_in_loop", "step_kernel_ms", "launches_per_denoise_step")})
PY
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/r4k_pytest_gpu.log 2>&1; tail -6 $O/r4k_pytest_gpu.log | cut -c1-300
