#!/bin/bash
# Round 5, call c: the 12-wave 320x192 tile (round-robin DMA piece assignment) -- kernel checks, tuner, bench A/B against the installed table;
# the fp16 engine with and without the tuning entries mirrored from the bf16 measurements; the multi-rank dry run on one GPU (2 and 4 ranks);
# PMC counters of the self-attention kernel variants on the TryonNet level-1 shape.
cd "$(dirname "$0")/.." || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/r5c_build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/r5c_build.log; exit 1; }
t0=$(date +%s)
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "w12_320x192 or w8_320x256 or xattn_fused" > $O/r5c_pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -4 $O/r5c_pytest_gpu.log | cut -c1-220
timeout 600 python tools/gpu_tune.py --out $O/r5c_tune_gfx950.json > $O/r5c_tune.log 2>&1; echo "tune rc=$?"; tail -2 $O/r5c_tune.log | cut -c1-200
show() { python - <<PY
import json
d = json.load(open("$1"))
r = d.get("roofline", {})
print("$2", round(d["value"], 4), "img/s loop", round(d["loop_ms_per_denoise_step"], 3), "ms/step frac", round(r.get("frac", 0), 4), r.get("step_kernel_ms"))
PY
}
for tab in installed new installed new; do
  [ $tab = new ] && export IDMVTON_TUNE_TABLE=$O/r5c_tune_gfx950.json || unset IDMVTON_TUNE_TABLE
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp16-leg --no-pmc > $O/r5c_bench_$tab.json.tmp 2> $O/r5c_bench_$tab.err; echo "bench $tab rc=$?"
  show $O/r5c_bench_$tab.json.tmp $tab; cat $O/r5c_bench_$tab.json.tmp >> $O/r5c_bench_$tab.json
done
unset IDMVTON_TUNE_TABLE
for m in mirrored untuned mirrored untuned; do
  [ $m = untuned ] && export IDMVTON_TUNE_NO_MIRROR=1 || unset IDMVTON_TUNE_NO_MIRROR
  timeout 300 python bench.py --dtype f16 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/r5c_bench_f16_$m.json.tmp 2> $O/r5c_bench_f16_$m.err; echo "bench f16 $m rc=$?"
  show $O/r5c_bench_f16_$m.json.tmp f16_$m; cat $O/r5c_bench_f16_$m.json.tmp >> $O/r5c_bench_f16_$m.json
done
unset IDMVTON_TUNE_NO_MIRROR
for n in 2 4; do
  timeout 600 python tools/multi_gpu_check.py --dry $n $O/r5c_multi_rank_dry_$n.json > $O/r5c_dry_$n.log 2>&1; echo "dry $n rc=$?"; tail -1 $O/r5c_dry_$n.log | cut -c1-600
done
cd /tmp
rm -rf $O/pmc_r5attn_a $O/pmc_r5attn_b
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_r5attn_a -o run -- python $R/tools/gpu_attn_pmc.py 3 > $O/r5c_attn_pmc_a.log 2>&1; echo "attn pmc a rc=$?"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $O/pmc_r5attn_b -o run -- python $R/tools/gpu_attn_pmc.py 3 > $O/r5c_attn_pmc_b.log 2>&1; echo "attn pmc b rc=$?"
cd $R
python tools/pmc_db_by_kernel.py $O/pmc_r5attn_a $O/pmc_r5attn_b --match attn | tee $O/r5c_pmc_attention.txt | cut -c1-250
find $O/pmc_r5attn_a $O/pmc_r5attn_b -name "*.db" -size +20M -delete 2>/dev/null
