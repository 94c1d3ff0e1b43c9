// gemm_tiles_sk.hip -- 128x128 tile with intra-workgroup split-K (gemm_body.cuh, SK = 2): 8 waves = two 4-wave groups on alternate k-tiles.
#include "gemm_body.cuh"

template <typename T, bool PF>
static void launch_sk(const GemmParams& p, bool lin, hipStream_t st) {
    const dim3 grid(p.tiles_n * p.tiles_m), block(512);
    if (lin) hipLaunchKernelGGL((gemm_sk_kernel<T, 128, 128, 2, 2, 2, true, PF>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_sk_kernel<T, 128, 128, 2, 2, 2, false, PF>), grid, block, 0, st, p);
}
template <typename T>
static int run(const GemmParams& p, int bn, int bm, int pf, bool lin, hipStream_t st) {
    if (!(bn == 128 && bm == 128)) return 1;
    if (pf) launch_sk<T, true>(p, lin, st); else launch_sk<T, false>(p, lin, st);
    return 0;
}
int gemm_tiles_sk(const GemmParams& p, bool bf16, int bn, int bm, int pf, bool lin, hipStream_t st) {
    return bf16 ? run<bf16_t>(p, bn, bm, pf, lin, st) : run<f16_t>(p, bn, bm, pf, lin, st);
}
