#!/bin/bash
# HBM traffic of the denoising-step kernels: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over the
# serial eager form of the bench command (per-dispatch counters need plain launches), summarised into profiles/pmc_traffic.json
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
LABEL=${1:-"unlabelled"}
export TMPDIR=/tmp
cd /tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $cnt -d $O/pmc_traffic_$cnt -o bench -- python $R/bench.py --steps 1 --warmup 0 --denoise-steps 6 --no-ramp --no-graph --no-overlap --no-cpu-baseline --no-roofline --no-fp16-leg > $O/pmc_traffic_$cnt.json 2> $O/pmc_traffic_$cnt.err; echo "pmc $cnt rc=$?"
done
cd $R
python tools/pmc_summary.py $O/pmc_traffic_FETCH_SIZE $O/pmc_traffic_WRITE_SIZE --label "$LABEL" > $O/pmc_traffic.json && python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_traffic.json"))
print({k:d.get(k) for k in ("gemm_conv_bytes_per_launch","gemm_conv_launches_counted","source")})
for k,v in d["denoise_step"].items():
    if "hbm_bytes_per_launch" in v: print(k, round(v["hbm_bytes_per_launch"]/1e6,2), "MB/launch over", v["FETCH_SIZE"]["dispatches"])
PY
find $O/pmc_traffic_FETCH_SIZE $O/pmc_traffic_WRITE_SIZE -name "*.db" -size +30M -delete 2>/dev/null
