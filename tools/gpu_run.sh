#!/bin/bash
# One parametrised GPU-box runner (replaces the per-experiment gpu_call_r*.sh scripts): tools/gpu_run.sh TAG 'command' -- runs the command from the
# repo root on the box with stdout+stderr into gpurun_out/TAG.log (bounded by TIMEOUT seconds, default 600) and prints the tail.
TAG=$1; shift
mkdir -p gpurun_out
timeout ${TIMEOUT:-600} bash -c "$*" > gpurun_out/$TAG.log 2>&1
echo "rc=$? ($TAG)"; tail -${TAIL:-30} gpurun_out/$TAG.log
