#!/bin/bash
# round 2, GPU call B: new attention kernels -- sanity (short timeout), kernel parity checks, variant timing; GEMM experiments
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python - > $O/r2b_sanity.log 2>&1 <<'PY'
import torch
from tests import kernel_checks as kc
for dt in (torch.bfloat16, torch.float16):
    for tn, tag in ((kc.pp_tune(2, 0), "s2d0"), (kc.pp_tune(3, 1), "s3d1"), (0, "auto")):
        e = kc.check_attn_self(4, 4, 768, dt, "cuda", n_garm=768, b0=2, tune=tn); torch.cuda.synchronize()
        print(dt, tag, "2seg", e, flush=True)
        e = kc.check_attn_self(2, 2, 200, dt, "cuda", n_garm=200, b0=1, tune=tn); torch.cuda.synchronize()
        print(dt, tag, "ragged", e, flush=True)
    print(dt, "cross", kc.check_attn_cross(4, 4, 768, dt, "cuda"), flush=True)
    print(dt, "vt", kc.check_vt(2, 768, 640, dt, "cuda"), flush=True)
PY
echo "sanity rc=$?"; cat $O/r2b_sanity.log | tail -20
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=line -p no:cacheprovider -k "attn or vt or probe or ring or gpu_available" > $O/r2b_pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -15 $O/r2b_pytest_attn.log
timeout 600 python tools/gpu_r2_probe.py attn > $O/r2b_probe_attn.log 2>&1; echo "attn probe rc=$?"; tail -100 $O/r2b_probe_attn.log
timeout 600 python tools/gpu_r2_probe.py gemm > $O/r2b_probe_gemm.log 2>&1; echo "gemm probe rc=$?"; tail -80 $O/r2b_probe_gemm.log
