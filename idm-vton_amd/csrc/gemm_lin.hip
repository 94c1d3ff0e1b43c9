// gemm_lin.hip -- hand-scheduled main loop for the large plain-Linear GEMMs of the denoising loop (GEGLU projection, ff.net.2,
// QKV, the GarmentNet M = 9216 shapes: include/idmvton_hip.h, tile_hint variant 4 / 5).
//
// Same arithmetic, operand layout and epilogue as gemm_conv.hip's 256x256 tile (D^T formulation, both operand tiles HBM -> LDS
// by LDS-DMA with the XOR swizzle on the DMA source address and on the ds_read_b128, 32x32x16 MFMA, fp32 accumulate) -- what
// differs is WHO places the instructions.  hipcc clusters a k-step as {all ds_reads} {all DMA issues} {16 MFMAs}: with one wave
// per SIMD nothing runs on the matrix pipe while the wave issues the first two groups (profiles/r03_quick_gemm_4wave_tile.log:
// 785 TFLOP/s against 918-946 for the 8-wave tile).  Here the loop is software-pipelined by hand and the placement is pinned
// with sched_group_barrier sequences, one MFMA per group:
//
//   tile t (64 deep, LDS buffer t&1), fragments double-buffered in registers (set A / set B), k-steps s = 0..3 of 16:
//     s = 0 : 16 MFMA(set A) | gaps 0-7: the 8 ds_reads of step 1 -> set B | gaps 8-15: LDS-DMA of tile t+1, weight rows
//     s = 1 : 16 MFMA(set B) | gaps 0-7: ds_reads of step 2 -> set A       | gaps 8-15: LDS-DMA of tile t+1, activation rows
//     s = 2 : 16 MFMA(set A) | gaps 0-7: ds_reads of step 3 -> set B
//     s = 3 : 16 MFMA(set B) | after WAIT_AT MFMAs: s_waitcnt vmcnt(0) (tile t+1 was issued >= 1000 cycles earlier) + s_barrier,
//                              then the ds_reads of tile t+1 step 0 -> set A, one per gap
//   so an MFMA gap (32 cycles of matrix pipe = ~8 issue slots) carries at most one LDS read or one DMA (+ its two scalar
//   instructions), the barrier is crossed with MFMAs in flight on both sides, and the only full drain per 64-deep tile is the
//   one vmcnt(0) on loads that have had half a tile to land.  One barrier per tile is enough: RAW -- every wave waits for its own
//   share of tile t+1 before the barrier; WAR -- buffer t&1 is refilled (tile t+2, issued in tile t+1) only after this barrier,
//   and every wave's last ds_read of tile t was waited for before its first MFMA of step 3, i.e. before it reached the barrier.
// Two geometries: 4 waves x (128 x 128) (one wave per SIMD, 256 accumulator registers in AGPRs) and 8 waves x (128 x 64).
#include "gemm_common.cuh"

template <typename T, int MI, int DMA_SPLIT, int WAIT_AT, int PRIO, bool TR>
__device__ __forceinline__ void gemm_lin_body(const GemmParams& p, char* smem, const int m0, const int n0) {
    typedef typename VT<T>::v8 v8;
    constexpr int BN = 256, BM = 256, NI = 4, SN = 128, SM = MI * 32;
    constexpr int WM = BM / SM, NW = 2 * WM;             // waves: 2 along n x WM along m
    constexpr int PW = 256 / (8 * NW);                   // 8-row DMA pieces per wave per operand tile (8 | 4)
    constexpr int OPB = 256 * 128;                       // bytes of one operand tile stage
    constexpr int BUF = 2 * OPB;                         // LDS map: [W0 | X0 | W1 | X1]

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int u = lane >> 5, l31 = lane & 31;

    // ---- loader: this lane's (row, swizzled 16-byte chunk) of each 8-row piece; the k offset of a tile rides in soffset ----
    const int lrow = lane >> 3, lslot = lane & 7;
    uint32_t w_off[PW], x_off[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int R = (wave * PW + i) * 8 + lrow;
        const int c = lslot ^ ((R >> 1) & 7);
        w_off[i] = ((uint32_t)(n0 + R) * (uint32_t)p.Ktot + c * 8) * 2u;                 // rows >= N: beyond num_records -> zeros
        const int m = m0 + R;
        x_off[i] = m < p.M ? ((uint32_t)m * (uint32_t)p.seg[0].pitch + p.seg[0].coff + c * 8) * 2u : OOB_SENTINEL;
    }
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    char* const dW = smem + wave * (PW * 1024);          // + buffer * BUF + piece * 1024
    char* const dX = smem + OPB + wave * (PW * 1024);
    auto dma_w = [&](int i, int buf_off, uint32_t koff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(dW + buf_off + i * 1024), 16, w_off[i], koff, 0, 0);
    };
    auto dma_x = [&](int i, int buf_off, uint32_t koff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(dX + buf_off + i * 1024), 16, x_off[i], koff, 0, 0);
    };

    // ---- fragment addresses: row l31 of the wave's sub-tile, chunk (2s + u) ^ swizzle; 32-row groups are immediate offsets ----
    const int swz = (l31 >> 1) & 7;
    int fa_off[4], fb_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ch = ((2 * s + u) ^ swz) << 4;
        fa_off[s] = (wn * SN + l31) * 128 + ch;
        fb_off[s] = OPB + (wm * SM + l31) * 128 + ch;
    }
    v8 fa[2][NI], fb[2][MI];
    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    auto read_a = [&](int set, int s, int buf_off, int ni) { fa[set][ni] = *(const v8*)(smem + buf_off + fa_off[s] + ni * 4096); };
    auto read_b = [&](int set, int s, int buf_off, int mi) { fb[set][mi] = *(const v8*)(smem + buf_off + fb_off[s] + mi * 4096); };
    auto read_all = [&](int set, int s, int buf_off) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) read_a(set, s, buf_off, ni);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) read_b(set, s, buf_off, mi);
    };
    auto mfma_range = [&](int set, int i0, int i1) {
#pragma unroll
        for (int idx = i0; idx < i1; ++idx) {
            const int ni = idx / MI, mi = idx % MI;
            acc[ni][mi] = TR ? VT<T>::mfma(fb[set][mi], fa[set][ni], acc[ni][mi]) : VT<T>::mfma(fa[set][ni], fb[set][mi], acc[ni][mi]);
        }
    };
    constexpr int NMF = NI * MI;                         // MFMAs per k-step (16 | 8)
    constexpr int NRD = NI + MI;                         // fragment reads per k-step (8 | 6)
    // pin one k-step: every gap starts with 1 MFMA; the first `nrd` gaps carry 1 DS read, the LAST `nv` gaps 1 VMEM (an LDS-DMA piece)
    auto pin = [&](int nrd, int nv) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
            if (i < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // DS read
            if (i >= NMF - nv) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    // VMEM
        }
    };

    const int nt = p.Ktot >> 6;
    // ---- prologue: tile 0 -> buffer 0, its step-0 fragments -> set 0 ----
#pragma unroll
    for (int i = 0; i < PW; ++i) { dma_w(i, 0, 0u); dma_x(i, 0, 0u); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_all(0, 0, 0);

    if constexpr (PRIO == 1) { if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1); }   // static priority for the second-dispatched half (guide T5)
    int cur = 0;                                         // byte offset of the buffer being computed (0 | BUF)
    for (int t = 0; t + 1 < nt; ++t) {
        const int nxt = cur ^ BUF;
        const uint32_t koff = (uint32_t)(t + 1) * 128u;
        // ---- step 0 ----
        read_all(1, 1, cur);
        if constexpr (DMA_SPLIT == 0) {                  // weight pieces behind step 0, activation pieces behind step 1
#pragma unroll
            for (int i = 0; i < PW; ++i) dma_w(i, nxt, koff);
        } else {                                         // everything behind step 0 (two pieces per late gap)
#pragma unroll
            for (int i = 0; i < PW; ++i) { dma_w(i, nxt, koff); dma_x(i, nxt, koff); }
        }
        mfma_range(0, 0, NMF);
        pin(NRD, DMA_SPLIT == 0 ? PW : 2 * PW);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 1 ----
        read_all(0, 2, cur);
        if constexpr (DMA_SPLIT == 0) {
#pragma unroll
            for (int i = 0; i < PW; ++i) dma_x(i, nxt, koff);
        }
        mfma_range(1, 0, NMF);
        pin(NRD, DMA_SPLIT == 0 ? PW : 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2 ----
        read_all(1, 3, cur);
        mfma_range(0, 0, NMF);
        pin(NRD, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 3: cross into tile t+1 ----
        mfma_range(1, 0, WAIT_AT);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        read_all(0, 0, nxt);
        mfma_range(1, WAIT_AT, NMF);
#pragma unroll
        for (int i = 0; i < NMF - WAIT_AT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    // ---- last tile: nothing left to fetch ----
    read_all(1, 1, cur);
    mfma_range(0, 0, NMF);
    pin(NRD, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_all(0, 2, cur);
    mfma_range(1, 0, NMF);
    pin(NRD, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_all(1, 3, cur);
    mfma_range(0, 0, NMF);
    pin(NRD, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_range(1, 0, NMF);

    gemm_epilogue<T, NI, MI, SN, SM, TR, BN, BM, NW * 64>(p, acc, m0, n0, wn, wm, lane, nullptr, smem);
}

template <typename T, int MI, int DMA_SPLIT, int WAIT_AT, int PRIO>
__global__ __launch_bounds__(MI == 4 ? 256 : 512, MI == 4 ? 1 : 2) void gemm_lin_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * 256 * 128];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tm, tn;
    {   // grouped raster (gemm_conv_kernel): an XCD's concurrent tiles form a ~1024-row output patch
        constexpr int GM = 4;
        const int width = GM * p.tiles_n;
        const int grp = wg / width, rem = wg - grp * width;
        const int first = grp * GM;
        const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
        tn = rem / gsz; tm = first + (rem - tn * gsz);
    }
    const int m0 = tm * 256, n0 = tn * 256;
    if (p.vt != nullptr && n0 >= p.vt_n0) gemm_lin_body<T, MI, DMA_SPLIT, WAIT_AT, PRIO, true>(p, smem, m0, n0);   // block-uniform: the V^T part of a fused QKV
    else gemm_lin_body<T, MI, DMA_SPLIT, WAIT_AT, PRIO, false>(p, smem, m0, n0);
}

// Called by gemm_conv.hip's launch_gemm for tile_hint variant 5 (8 waves x 128x64); `form` (the low nibble of tile_hint's BM field, always 0
// for a 256-row tile) selects the placement under measurement.  Preconditions checked by the caller: plain Linear (one K segment, no
// gather), 16-byte epilogue (or a V^T part), no folded LayerNorm.  (Variant 4 = 4 waves x 128x128, one wave per SIMD, was measured in
// round 4 and removed: 887-913 TFLOP/s on the 3072x10240x1280 GEGLU against 989-1008 for this geometry -- an LDS-DMA instruction costs
// its wave ~60 issue cycles, twice an MFMA gap, and with one wave per SIMD nobody else feeds the matrix pipe meanwhile.)
template <typename T>
static int launch_lin(const GemmParams& p, int form, hipStream_t st) {
    const dim3 grid(p.tiles_n * p.tiles_m), block(512);
    switch (form) {
    case 0: hipLaunchKernelGGL((gemm_lin_kernel<T, 2, 0, 2, 0>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((gemm_lin_kernel<T, 2, 1, 0, 0>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((gemm_lin_kernel<T, 2, 1, 4, 0>), grid, block, 0, st, p); break;
    case 4: hipLaunchKernelGGL((gemm_lin_kernel<T, 2, 1, 2, 1>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_lin_kernel<T, 2, 1, 2, 0>), grid, block, 0, st, p); break;
    }
    return 0;
}
int launch_gemm_lin(const GemmParams& p, bool bf16, int form, hipStream_t st) {
    return bf16 ? launch_lin<bf16_t>(p, form, st) : launch_lin<f16_t>(p, form, st);
}
