// gemm_lin.hip -- hand-scheduled main loop for the large plain-Linear GEMMs of the denoising loop (GEGLU projection, ff.net.2,
// QKV, the GarmentNet M = 9216 shapes: include/idmvton_hip.h, tile_hint variant 5).
//
// Same arithmetic, operand layout and epilogue as gemm_conv.hip's 256-column tiles (D^T formulation, both operand tiles HBM -> LDS
// by LDS-DMA with the XOR swizzle on the DMA source address and on the ds_read_b128, 32x32x16 MFMA, fp32 accumulate) -- what
// differs is WHO places the instructions.  hipcc clusters a k-step as {all ds_reads} {all DMA issues} {all MFMAs}; here the loop is
// software-pipelined by hand and the placement is pinned with sched_group_barrier sequences, one MFMA per group (read back from the emitted
// ISA: steps 1-3 are exactly the pinned interleave; in step 0 of form 1 the backend keeps the 6 reads and the 8 DMA pieces together behind the
// first MFMA and issues the remaining 7 MFMAs back to back -- which is the fastest form measured):
//
//   tile t (64 deep, LDS buffer t&1), fragments double-buffered in registers (set A / set B), k-steps s = 0..3 of 16:
//     s = 0 : MFMAs(set A) | first gaps: the ds_reads of step 1 -> set B | last gaps: LDS-DMA pieces of tile t+1
//     s = 1 : MFMAs(set B) | first gaps: ds_reads of step 2 -> set A     | last gaps: the rest of tile t+1 (DMA_SPLIT = 0)
//     s = 2 : MFMAs(set A) | first gaps: ds_reads of step 3 -> set B
//     s = 3 : MFMAs(set B) | after WAIT_AT MFMAs: s_waitcnt vmcnt(0) (tile t+1 was issued >= 1000 cycles earlier) + s_barrier,
//                            then the ds_reads of tile t+1 step 0 -> set A, one per gap
//   so an MFMA gap (32 cycles of matrix pipe = ~8 issue slots) carries at most one LDS read and one DMA (+ its scalar m0 write), the
//   barrier is crossed with MFMAs in flight on both sides, and the only full drain per 64-deep tile is the one vmcnt(0) on loads
//   that have had half a tile to land.  One barrier per tile is enough: RAW -- every wave waits for its own share of tile t+1
//   before the barrier; WAR -- buffer t&1 is refilled (tile t+2, issued in tile t+1) only after this barrier, and every wave's last
//   ds_read of tile t was waited for before its first MFMA of step 3, i.e. before it reached the barrier.
//
// Geometries (8 waves = two per SIMD: one wave's DMA issue -- ~60 cycles per LDS-DMA instruction, twice an MFMA gap -- is covered by
// its partner's MFMAs; the 4-wave 128x128-per-wave form measured 887-913 TFLOP/s on the 3072x10240x1280 GEGLU against 989-1027 for the
// 8-wave forms, profiles/r04_gemm_probe_h4_vs_h5_v1.log, and the same geometry fed by plain buffer loads + ds_write_b128 instead of LDS-DMA
// (fire-and-forget issue, a whole k-tile for the data to land) 855 against 1024, profiles/r04_gemm_probe_h4g_buffer_load_form.log: one wave
// per SIMD is not held back by the DMA issue but by having nobody to cover ANY of its waits -- both removed, kept in history).
// What bounds the 8-wave loop (profiles/r04_gemm_diag_k_frozen_no_dma.log, 8192^3): with NO operand delivery inside the loop it runs 1599
// TFLOP/s = hipBLASLt's whole kernel (1616); every k-tile re-reading tile 0 (L2-hot) 1403; the real thing 1323.  So the fragment reads,
// MFMAs and the barrier are at the library's level, the delivery costs 12 % even L2-hot and L2 misses another 6 %.  Staging the same
// pieces through registers (buffer_load_b128 in step 0, ds_write_b128 in step 2) instead of LDS-DMA is 3-4 % slower on every shape
// (profiles/r04_gemm_probe_h5v_register_staged.log) -- removed too:
//
// PERSISTENT TILES: the grid is min(tiles, CUs) workgroups and each walks tiles b, b + G, b + 2G, ... of the XCD-aware grouped raster.  While a
// tile computes its LAST k-tile, the first k-tile of the workgroup's next output tile is already on its way into the other LDS buffer, so only
// the first tile of a workgroup pays the cold start (every later one finds its operands landed when its epilogue ends); the epilogue's stores
// drain under the next tile's main loop.  Launches with more than one round of tiles (GEGLU: 1.9 - 11 rounds) lose one operand round trip
// per round otherwise.
//   256 x 256 : 2 (n) x 4 (m) waves of 128 x 64
//   256 x 192 : 4 (n) x 2 (m) waves of  64 x 96   -- 3072 x 3840 (fused QKV) and 9216 x 1280 give 240 tiles of it (one per CU, 94 % of the
//                                                    chip) where 256 x 256 gives 180 (70 %)
#include "gemm_common.cuh"
#include <type_traits>

template <typename T, int BM, int WN, int WM, int DMA_SPLIT, int WAIT_AT>
__device__ __forceinline__ void gemm_lin_persistent(const GemmParams& p, char* smem) {
    typedef typename VT<T>::v8 v8;
    constexpr int BN = 256, NW = WN * WM;
    constexpr int SN = BN / WN, SM = BM / WM, NI = SN / 32, MI = SM / 32;
    constexpr int PWW = BN / (8 * NW), PWX = BM / (8 * NW);      // 8-row DMA pieces per wave per tile: weight rows, activation rows
    static_assert(PWW * 8 * NW == BN && PWX * 8 * NW == BM && NI * 32 * WN == BN && MI * 32 * WM == BM, "tile / wave-count mismatch");
    constexpr int OPW = BN * 128;                        // bytes of one weight tile stage
    constexpr int BUF = (BN + BM) * 128;                 // LDS map: [W0 | X0 | W1 | X1]

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int u = lane >> 5, l31 = lane & 31;

    // ---- tile walk: workgroup b owns tiles b, b + G, ... (G % 8 == 0 keeps a workgroup's tiles in its XCD's range of the raster) ----
    const int ntiles = p.tiles_m * p.tiles_n, G = gridDim.x;
    auto coords = [&](int idx, int& m0, int& n0) {
        const int wg = xcd_remap(idx, ntiles);
        const int GM = p.gm > 0 ? p.gm : 1024 / BM;      // grouped raster (gemm_conv_kernel): an XCD's concurrent tiles form a GM-tile-tall output patch
        const int width = GM * p.tiles_n;
        const int grp = wg / width, rem = wg - grp * width;
        const int first = grp * GM;
        const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
        const int tn = rem / gsz;
        m0 = (first + (rem - tn * gsz)) * BM;
        n0 = tn * BN;
    };

    // ---- loader: this lane's (row, swizzled 16-byte chunk) of each 8-row piece; the k offset of a tile rides in soffset ----
    const int lrow = lane >> 3, lslot = lane & 7;
    uint32_t w_off[PWW], x_off[PWX];
    auto set_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PWW; ++i) {
            const int R = (wave * PWW + i) * 8 + lrow;
            w_off[i] = ((uint32_t)(n0 + R) * (uint32_t)p.Ktot + (lslot ^ ((R >> 1) & 7)) * 8) * 2u;   // rows >= N: beyond num_records -> zeros
        }
#pragma unroll
        for (int i = 0; i < PWX; ++i) {
            const int R = (wave * PWX + i) * 8 + lrow;
            const int m = m0 + R;
            x_off[i] = m < p.M ? ((uint32_t)m * (uint32_t)p.seg[0].pitch + p.seg[0].coff + (lslot ^ ((R >> 1) & 7)) * 8) * 2u : OOB_SENTINEL;
        }
    };
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    char* const dW = smem + wave * (PWW * 1024);         // + buffer * BUF + piece * 1024
    char* const dX = smem + OPW + wave * (PWX * 1024);
    auto dma_w = [&](int i, int buf_off, uint32_t koff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(dW + buf_off + i * 1024), 16, w_off[i], koff, 0, 0);
    };
    auto dma_x = [&](int i, int buf_off, uint32_t koff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(dX + buf_off + i * 1024), 16, x_off[i], koff, 0, 0);
    };
    auto dma_first = [&](int buf_off) {                  // k-tile 0 of the tile the offsets point at
#pragma unroll
        for (int i = 0; i < PWW; ++i) dma_w(i, buf_off, 0u);
#pragma unroll
        for (int i = 0; i < PWX; ++i) dma_x(i, buf_off, 0u);
    };

    // ---- fragment addresses: row l31 of the wave's sub-tile, chunk (2s + u) ^ swizzle; 32-row groups are immediate offsets ----
    const int swz = (l31 >> 1) & 7;
    int fa_off[4], fb_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ch = ((2 * s + u) ^ swz) << 4;
        fa_off[s] = (wn * SN + l31) * 128 + ch;
        fb_off[s] = OPW + (wm * SM + l31) * 128 + ch;
    }
    v8 fa[2][NI], fb[2][MI];
    f32x16 acc[NI][MI];

    auto read_all = [&](int set, int s, int buf_off) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fa[set][ni] = *(const v8*)(smem + buf_off + fa_off[s] + ni * 4096);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fb[set][mi] = *(const v8*)(smem + buf_off + fb_off[s] + mi * 4096);
    };
    constexpr int NMF = NI * MI;                         // MFMAs per k-step (8 | 6)
    constexpr int NRD = NI + MI;                         // fragment reads per k-step (6 | 5)
    // pin one k-step: every gap starts with 1 MFMA; the first `nrd` gaps carry 1 DS read, the LAST `nv` gaps 1 VMEM (an LDS-DMA piece);
    // more pieces than gaps: a second one per gap from the front
    auto pin = [&](int nrd, int nv) {
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
            if (i < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // DS read
            if (i >= NMF - nv) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);    // VMEM
            if (i < nv - NMF) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);     // VMEM (overflow)
        }
    };

    const int nt = p.Ktot >> 6;
    int idx = blockIdx.x, m0, n0, nm0 = 0, nn0 = 0;
    bool has_next = false;
    int cur = 0;                                         // byte offset of the buffer being computed (0 | BUF)

    // one output tile: k-tile 0 is in buffer `cur` (every wave has waited for its own pieces); ends with the accumulators stored and,
    // if there is a next tile, its k-tile 0 landed in the other buffer and the offsets pointing at it
    auto run_tile = [&](auto trc) {
        constexpr bool TR = decltype(trc)::value;
        auto mfma_range = [&](int set, int i0, int i1) {
#pragma unroll
            for (int k = i0; k < i1; ++k) {
                const int ni = k / MI, mi = k % MI;
                acc[ni][mi] = TR ? VT<T>::mfma(fb[set][mi], fa[set][ni], acc[ni][mi]) : VT<T>::mfma(fa[set][ni], fb[set][mi], acc[ni][mi]);
            }
        };
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
        __builtin_amdgcn_s_barrier();                    // everybody's pieces of k-tile 0 have landed; nobody still reads the other buffer
        asm volatile("" ::: "memory");
        read_all(0, 0, cur);
        for (int t = 0; t + 1 < nt; ++t) {
            const int nxt = cur ^ BUF;
            const uint32_t koff = (uint32_t)(t + 1) * 128u;
            // ---- step 0 ----
            read_all(1, 1, cur);
#pragma unroll
            for (int i = 0; i < PWW; ++i) dma_w(i, nxt, koff);
            if constexpr (DMA_SPLIT == 1) {              // everything behind step 0
#pragma unroll
                for (int i = 0; i < PWX; ++i) dma_x(i, nxt, koff);
            }
            mfma_range(0, 0, NMF);
            pin(NRD, DMA_SPLIT == 0 ? PWW : PWW + PWX);
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 1 ----
            read_all(0, 2, cur);
            if constexpr (DMA_SPLIT == 0) {              // weight pieces behind step 0, activation pieces behind step 1
#pragma unroll
                for (int i = 0; i < PWX; ++i) dma_x(i, nxt, koff);
            }
            mfma_range(1, 0, NMF);
            pin(NRD, DMA_SPLIT == 0 ? PWX : 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 2 ----
            read_all(1, 3, cur);
            mfma_range(0, 0, NMF);
            pin(NRD, 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 3: cross into tile t+1 ----
            mfma_range(1, 0, WAIT_AT);
            __builtin_amdgcn_sched_barrier(0);
            // vmcnt: this wave's pieces of k-tile t+1 have landed.  lgkmcnt: ALL its fragment reads of k-tile t are done before anybody may
            // refill that buffer (they were issued a k-step ago; with 3 activation fragments per step the first WAIT_AT MFMAs alone would leave
            // the last one formally in flight across the barrier)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            read_all(0, 0, nxt);
            mfma_range(1, WAIT_AT, NMF);
#pragma unroll
            for (int i = 0; i < NMF - WAIT_AT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i == NMF - WAIT_AT - 1 && NRD > NMF - WAIT_AT) __builtin_amdgcn_sched_group_barrier(0x100, NRD - (NMF - WAIT_AT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        // ---- last k-tile: the other buffer is free (its k-tile was read before the barrier above): send the NEXT output tile's k-tile 0 there ----
        if (has_next) {                                  // block-uniform
            set_offsets(nm0, nn0);
            dma_first(cur ^ BUF);
        }
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
        // the next tile's pieces have had this whole k-tile to land; waiting here (not after the stores below) keeps the stores out of the wait
        if (has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gemm_epilogue<T, NI, MI, SN, SM, TR>(p, acc, m0, n0, wn, wm, lane);
    };

    coords(idx, m0, n0);
    set_offsets(m0, n0);
    dma_first(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (;;) {
        const int nidx = idx + G;
        has_next = nidx < ntiles;
        if (has_next) coords(nidx, nm0, nn0);
        if (p.vt != nullptr && n0 >= p.vt_n0) run_tile(std::true_type{});      // block-uniform: the V^T part of a fused QKV
        else run_tile(std::false_type{});
        if (!has_next) break;
        cur ^= BUF; idx = nidx; m0 = nm0; n0 = nn0;
    }
}

template <typename T, int BM, int WN, int WM, int DMA_SPLIT, int WAIT_AT>
__global__ __launch_bounds__(512, 2) void gemm_lin_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * (256 + BM) * 128];
    gemm_lin_persistent<T, BM, WN, WM, DMA_SPLIT, WAIT_AT>(p, smem);
}

// Called by gemm_conv.hip's launch_gemm for tile_hint variant 5 (BN = 256, BM = 256 | 192); `form` (the low nibble of tile_hint's BM field) selects
// the placement of the 256-row tile: 0 = DMA split over steps 0 / 1, 1 (the form the tuning table uses) = everything behind step 0.  Forms 2-4
// of the first measurement (barrier after 0 / 4 MFMAs, static priority for the younger half of the workgroup) were within noise of form 1
// (profiles/r04_gemm_probe_h5_forms_v1.log) and were removed.  Precondition checked by the caller: plain Linear (one K segment, no gather).  grid_cap > 0 (tests only) limits the persistent grid so that small shapes walk several tiles per workgroup.
static int persistent_grid(int ntiles, int grid_cap) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus = n & ~7;                                    // a multiple of the 8 XCDs: tile b + j*G stays in workgroup b's XCD range
    }
    const int cap = grid_cap > 0 ? grid_cap : cus;
    return ntiles < cap ? ntiles : cap;
}
template <typename T>
static int launch_lin(const GemmParams& p0, int bm, int form, int grid_cap, hipStream_t st) {
    GemmParams p = p0;
    p.gm = idmvton_choose_gm(p.tiles_m, p.tiles_n, bm, 256, p.Ktot);
    const dim3 grid(persistent_grid(p.tiles_n * p.tiles_m, grid_cap)), block(512);
    if (bm == 192) hipLaunchKernelGGL((gemm_lin_kernel<T, 192, 4, 2, 0, 2>), grid, block, 0, st, p);
    else if (form == 0) hipLaunchKernelGGL((gemm_lin_kernel<T, 256, 2, 4, 0, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_lin_kernel<T, 256, 2, 4, 1, 2>), grid, block, 0, st, p);
    return 0;
}
int launch_gemm_lin(const GemmParams& p, bool bf16, int bm, int form, int grid_cap, hipStream_t st) {
    return bf16 ? launch_lin<bf16_t>(p, bm, form, grid_cap, st) : launch_lin<f16_t>(p, bm, form, grid_cap, st);
}
