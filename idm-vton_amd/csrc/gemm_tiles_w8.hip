// gemm_tiles_w8.hip -- tile family "w8": more waves per workgroup on the tiles that run ONE workgroup per CU (see gemm_conv.hip, variant 6).
// The low nibble of the BM field selects the form of the 128x128 tile: 0 = 8 waves, 3-stage ring; 1 = the same with the fragments of k-step
// s+1 read ahead of the MFMAs of step s; 2 = 8 waves, 4-stage ring (128 KiB).
#include "gemm_body.cuh"

template <typename T>
static int run(const GemmParams& p, int bn, int bm, int form, bool lin, hipStream_t st) {
    if (bn == 128 && bm == 128 && form == 0) launch_cfg<T, 128, 128, 2, 4, 3, true, 2>(p, lin, st);          // 64x32 per wave, two waves per SIMD on one k-tile
    else if (bn == 128 && bm == 128 && form == 1) launch_cfg<T, 128, 128, 2, 4, 3, true, 2, true>(p, lin, st);
    else if (bn == 320 && bm == 192) launch_cfg<T, 320, 192, 2, 6, 2, true, 3>(p, lin, st);                  // 12 waves of 160x32: M = 49152 gives 256 tiles (320x256: 192)
    else if (bn == 256 && bm == 192) launch_cfg<T, 256, 192, 4, 3, 2, true, 3>(p, lin, st);                  // 12 waves of 64x64: 3072 x 3840 / 9216 x 1280 -> 240 tiles
    else if (bn == 256 && bm == 256) launch_cfg<T, 256, 256, 4, 4, 2, true, 4>(p, lin, st);                  // 16 waves of 64x64: four waves per SIMD
    else if (bn == 128 && bm == 256) launch_cfg<T, 128, 256, 2, 8, 3, true, 4>(p, lin, st);                  // 16 waves of 64x32
    else return 1;
    return 0;
}
int gemm_tiles_w8(const GemmParams& p, bool bf16, int bn, int bm, int form, bool lin, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, form, lin, st) : run<f16_t>(p, bn, bm, form, lin, st);
}
