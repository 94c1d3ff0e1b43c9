#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "groupnorm or layernorm or elementwise" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "tiny_pipeline or bit_reproducible" 2>&1 | tail -3
for tag in cur prev cur prev; do
  lib=$R/idm-vton_amd/libidmvton_hip.so; [ $tag != cur ] && lib=$R/idm-vton_amd/libidmvton_hip_$tag.so
  IDMVTON_HIP_LIB=$lib timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value'],4), round(d['ms_per_step'],1), d['roofline']['step_kernel_ms'])"
done
