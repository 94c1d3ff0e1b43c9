// gemm_tiles_xattn.hip -- attn2.to_q with the cross-attention as its epilogue (xattn.cuh): plain Linear loader, ring pipeline.
#include "gemm_body.cuh"

template <typename T>
static int run(const GemmParams& p0, int bn, int bm, int w8, hipStream_t st) {
    GemmParams p = p0;
    p.gm = idmvton_choose_gm(p.tiles_m, p.tiles_n, bm, bn, p.Ktot);
    const dim3 grid(p.tiles_n * p.tiles_m);
    if (bn == 128 && bm == 128 && w8) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 128, 2, 4, 3, 2>), grid, dim3(512), 0, st, p);   // 64x32 per wave
    else if (bn == 128 && bm == 64) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 64, 2, 2, 3, 2>), grid, dim3(256), 0, st, p);
    else if (bn == 128 && bm == 128) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 128, 2, 2, 3, 1>), grid, dim3(256), 0, st, p);
    else if (bn == 128 && bm == 256) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 256, 2, 4, 3, 2>), grid, dim3(512), 0, st, p);
    else return 1;
    return 0;
}
int gemm_tiles_xattn(const GemmParams& p, bool bf16, int bn, int bm, int w8, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, w8, st) : run<f16_t>(p, bn, bm, w8, st);
}
