#!/bin/bash
# same-box A/B of library builds (current vs earlier kernel states) through bench.py
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
for tag in cur v3 v4 cur v3; do
  lib=$PWD/idm-vton_amd/libidmvton_hip.so; [ $tag != cur ] && lib=$PWD/idm-vton_amd/libidmvton_hip_$tag.so
  IDMVTON_HIP_LIB=$lib timeout 40 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value'],4), round(d['ms_per_step'],1))" | tee -a $O/ab_libs.log
done
