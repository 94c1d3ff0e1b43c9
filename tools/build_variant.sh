#!/bin/bash
# A/B library builds: tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS" -> .ab_r06/libNAME.so (= the in-tree objects with FILE.hip rebuilt under the flags);
# select it at run time with IDMVTON_HIP_LIB=.ab_r06/libNAME.so.  .ab_r06/ is git-ignored and travels to the GPU box.
set -e
cd "$(dirname "$0")/../idm-vton_amd/csrc"
NAME=$1; FILE=$2; EXTRA=$3
OUT=../../.ab_r06; mkdir -p $OUT
BASEFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value"
case $FILE in attention.hip|attention_f8.hip) BASEFLAGS="$BASEFLAGS ${ATTN_FLAGS--mllvm -amdgpu-mfma-vgpr-form} -fno-honor-nans";; esac
/opt/rocm/bin/hipcc $BASEFLAGS $EXTRA -c $FILE -o $OUT/${FILE%.hip}_$NAME.o
OBJS=""
for f in gemm_conv gemm_tiles_v0 gemm_tiles_v1 gemm_tiles_v2 gemm_tiles_w8 gemm_tiles_xattn gemm_lin attention attention_f8 attn_small norm elementwise rccl_arena; do
  if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS $OUT/${f}_$NAME.o"; else OBJS="$OBJS $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o $OUT/lib$NAME.so
echo built $OUT/lib$NAME.so
