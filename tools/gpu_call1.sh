#!/bin/bash
# One gpurun call: kernel/parity tests by family (separate processes: a faulting variant must not take the rest down),
# per-shape tuning restricted to the families that passed, A/B of the engine modes, then the standard bench line.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
T="timeout 600 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider"
$T -k "not ring_ and not _w2s and not _w4s and not _w8s and not overlap" > $O/pytest_base.log 2>&1; RB=$?
$T -k "ring_" > $O/pytest_ring.log 2>&1; RR=$?
$T -k "_w2s or _w4s or _w8s" > $O/pytest_attnvar.log 2>&1; RA=$?
$T -k "overlap" > $O/pytest_overlap.log 2>&1; RO=$?
echo "pytest rc: base=$RB ring=$RR attnvar=$RA overlap=$RO" | tee $O/pytest_rc.txt
for f in base ring attnvar overlap; do echo "== $f"; tail -4 $O/pytest_$f.log; done
FL=""
[ $RR -ne 0 ] && FL="$FL --skip-ring"
[ $RA -ne 0 ] && FL="$FL --skip-attn-variants"
timeout 900 python tools/gpu_tune.py $FL > $O/tune.log 2>&1; echo "tune rc=$?"; tail -3 $O/tune.log
MODES="serial_graph,overlap_graph,overlap_eager"
[ $RO -ne 0 ] && MODES="serial_graph"
timeout 900 python tools/gpu_ab.py --modes $MODES > $O/ab.log 2>&1; echo "ab rc=$?"; grep '^{' $O/ab.log
