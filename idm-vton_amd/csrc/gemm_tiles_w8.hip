// gemm_tiles_w8.hip -- tile family "w8: 8-wave forms of the one-workgroup-per-CU tiles" of the MFMA implicit GEMM (see gemm_conv.hip / gemm_body.cuh).
#include "gemm_body.cuh"

template <typename T>
static int run(const GemmParams& p, int bn, int bm, bool lin, hipStream_t st) {
    if (bn == 128 && bm == 128) launch_cfg<T, 128, 128, 2, 4, 3, true, 2>(p, lin, st);          // 64x32 per wave, two waves per SIMD on one k-tile
    else if (bn == 320 && bm == 256) launch_cfg<T, 320, 256, 2, 4, 2, true, 2>(p, lin, st);     // 160x64 per wave (5 x 2 MFMA tiles), 144 KiB
    else return 1;
    return 0;
}
int gemm_tiles_w8(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, lin, st) : run<f16_t>(p, bn, bm, lin, st);
}
