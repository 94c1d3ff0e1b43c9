// gemm_tiles_v2.hip -- tile family "v2: ring tiles with the fragments of k-step s+1 read ahead of the MFMAs of step s (the two tiles the tuning table selects)" of the MFMA implicit GEMM (see gemm_conv.hip / gemm_body.cuh).
#include "gemm_body.cuh"

template <typename T>
static int run(const GemmParams& p, int bn, int bm, bool lin, hipStream_t st) {
    if (bn == 128 && bm == 256) launch_cfg<T, 128, 256, 2, 4, 3, true, 2, true>(p, lin, st);
    else if (bn == 64 && bm == 64) launch_cfg<T, 64, 64, 2, 2, 4, true, 2, true>(p, lin, st);
    else return 1;
    return 0;
}
int gemm_tiles_v2(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, lin, st) : run<f16_t>(p, bn, bm, lin, st);
}
