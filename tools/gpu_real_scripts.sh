#!/bin/bash
# Runs the reference's UNMODIFIED inference.py / inference_dc.py to pixels on the MI355X.  Called on the
# build box: makes an UNTRACKED scratch copy of the two scripts (reference sources are never committed), spends one gpurun call on
# tests/test_dropin_gpu.py, deletes the copy.  Log -> gpurun_out/real_scripts.log (copied to profiles/ by hand).
cd "$(dirname "$0")/.." || exit 1
mkdir -p .scratch_ref gpurun_out
cp /root/reference/inference.py /root/reference/inference_dc.py .scratch_ref/ || exit 1
/usr/local/graft/bin/gpurun --timeout 1200 -- 'export TMPDIR=/tmp; python -m pytest tests/test_dropin_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/real_scripts.log 2>&1; tail -15 gpurun_out/real_scripts.log'
rc=$?
rm -rf .scratch_ref
exit $rc
