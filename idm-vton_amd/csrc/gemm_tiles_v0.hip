// gemm_tiles_v0.hip -- tile family "v0: 2-stage 4-wave tiles, 2-3 workgroups per CU hide the load latency" of the MFMA implicit GEMM (see gemm_conv.hip / gemm_body.cuh).
#include "gemm_body.cuh"

template <typename T>
static int run(const GemmParams& p, int bn, int bm, bool lin, hipStream_t st) {
    if (bn == 128 && bm == 128) launch_cfg<T, 128, 128, 2, 2, 2, false, 2>(p, lin, st);
    else if (bn == 128 && bm == 64) launch_cfg<T, 128, 64, 2, 2, 2, false, 3>(p, lin, st);
    else if (bn == 64 && bm == 64) launch_cfg<T, 64, 64, 2, 2, 2, false, 3>(p, lin, st);
    else return 1;
    return 0;
}
int gemm_tiles_v0(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, lin, st) : run<f16_t>(p, bn, bm, lin, st);
}
