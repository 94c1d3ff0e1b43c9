// gemm_body.cuh -- the MFMA implicit-GEMM main loop (gemm_body), its kernels (gemm_conv_kernel, gemm_xattn_kernel) and the per-tile launcher
// (launch_cfg), shared by the translation units that instantiate the tile configurations (gemm_tiles_*.hip: one family each, so the library
// builds in parallel) -- see gemm_conv.hip for the formulation, data movement and pipeline forms.
#pragma once
#include <type_traits>
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "gemm_body.cuh: the counted vmcnt pipeline is written for gfx94x/gfx950 (stores counted in vmcnt)"
#endif
#include "common.cuh"
#include "gemm_common.cuh"

// TR = this n-tile is written transposed (V^T epilogue): the MFMA operands are swapped so the accumulator is D[m][n].
// Block = WN x WM waves; wave (wn, wm) owns the (BN/WN) x (BM/WM) sub-tile as NI x MI 32x32 MFMA tiles.
template <typename T, int BN, int BM, int WN, int WM, int ST, bool V1, bool LIN, bool PF, bool TR, bool XA = false>
__device__ __forceinline__ void gemm_body(const GemmParams& p, char* smem, const int m0, const int n0) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int NW = WN * WM;
    constexpr int SN = BN / WN, SM = BM / WM;           // wave sub-tile
    constexpr int NI = SN / 32, MI = SM / 32;           // 32x32 MFMA tiles per wave along n / m
    constexpr int WBYTES = BN * 128, XBYTES = BM * 128; // one LDS stage of each operand
    // DMA instructions per wave per tile (8-row pieces).  Tiles whose piece counts divide by the wave count give every wave a contiguous run of
    // pieces; the others (320x192 on 12 waves: 40 + 24 pieces) deal the pieces round-robin, some waves issuing one piece fewer -- which a counted
    // vmcnt cannot express, so those tiles run the two-stage pipeline (vmcnt(0) in front of every barrier).
    constexpr int PW_ = BN / 8, PX_ = BM / 8;
    constexpr bool UNEVEN = (PW_ % NW != 0) || (PX_ % NW != 0);
    constexpr int WI = (PW_ + NW - 1) / NW, XI = (PX_ + NW - 1) / NW;
    static_assert(BN % 8 == 0 && BM % 8 == 0 && WN * SN == BN && WM * SM == BM && NI * 32 == SN && MI * 32 == SM, "tile / wave-count mismatch");
    static_assert(!UNEVEN || (V1 && ST == 2), "round-robin piece assignment needs the two-stage ring");
    char* sW = smem;
    char* sX = smem + ST * WBYTES;

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int u = lane >> 5, l31 = lane & 31;

    // ---- loader state: this lane's rows / swizzled chunk for each DMA instruction it issues ----
    const int lrow = lane >> 3, lslot = lane & 7;
    auto wj = [&](int i) { return UNEVEN ? wave + i * NW : wave * WI + i; };    // index of this wave's i-th weight / activation piece in the tile
    auto xj = [&](int i) { return UNEVEN ? wave + i * NW : wave * XI + i; };    // (UNEVEN: pieces >= PW_ / PX_ do not exist and are skipped)
    uint32_t w_off[WI];                                  // byte offset of (row, chunk) in W, k0 excluded
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int R = wj(i) * 8 + lrow;
        const int c = lslot ^ ((R >> 1) & 7);
        w_off[i] = ((uint32_t)(n0 + R) * (uint32_t)p.Ktot + c * 8) * 2u;   // rows >= N fall beyond num_records -> 0
    }
    int x_pix[XI], x_oy[XI], x_ox[XI], x_c8[XI];
    uint32_t x_off[XI];                                  // LIN only
    const int HoWo = p.Ho * p.Wo;
    auto kbyte = [&](int t) -> uint32_t { return (uint32_t)t * 128u; };
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int R = xj(i) * 8 + lrow;
        const int m = m0 + R;
        x_c8[i] = (lslot ^ ((R >> 1) & 7)) * 8;
        if constexpr (LIN) {
            x_off[i] = m < p.M ? ((uint32_t)m * (uint32_t)p.seg[0].pitch + p.seg[0].coff + x_c8[i]) * 2u : OOB_SENTINEL;
        } else if (m < p.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            x_pix[i] = b * p.Hi * p.Wi; x_oy[i] = oy * p.stride; x_ox[i] = ox * p.stride;
        } else {
            x_pix[i] = 0; x_oy[i] = -(1 << 28); x_ox[i] = 0;   // fails every bounds test -> zeros
        }
    }
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t rs_x0 = make_rsrc(p.seg[0].ptr, p.seg[0].bytes);
    // fused nearest upsample: the upsampled image IS the output grid (a same-size 3x3 / 1x1 convolution over it); Ho x Wo may be one short of
    // 2 Hi x 2 Wi -- diffusers' `upsample_size` for skip tensors of odd size (src/unet_hacked_tryon.py:1084-1090,1357-1379): F.interpolate(size=)
    // nearest maps output y to floor(y * Hi / Ho), which is y >> 1 for every Ho in {2 Hi - 1, 2 Hi}
    const int hin = p.ups ? p.Ho : p.Hi, win = p.ups ? p.Wo : p.Wi;

    int si = 0, kseg = 0;                                // K-segment cursor of the NEXT tile to issue
    auto issue = [&](int t, int buf) {
        char* dW = sW + buf * WBYTES;
        char* dX = sX + buf * XBYTES;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            if (UNEVEN && wj(i) >= PW_) continue;         // wave-uniform
            dma16(rs_w, dW + wj(i) * 1024, w_off[i] + kbyte(t));
        }
        if constexpr (LIN) {
#pragma unroll
            for (int i = 0; i < XI; ++i) {                // OOB_SENTINEL + t*128 stays >= 2 GiB > num_records
                if (UNEVEN && xj(i) >= PX_) continue;
                dma16(rs_x0, dX + xj(i) * 1024, x_off[i] + kbyte(t));
            }
        } else {
            const idmvton_seg sg = p.seg[si];
            const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(sg.ptr, sg.bytes);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                if (UNEVEN && xj(i) >= PX_) continue;
                int iy = x_oy[i] + sg.dy, ix = x_ox[i] + sg.dx;
                const bool ok = (unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const uint32_t off = ((uint32_t)(x_pix[i] + iy * p.Wi + ix) * (uint32_t)sg.pitch + sg.coff + kseg + x_c8[i]) * 2u;
                dma16(rs_x, dX + xj(i) * 1024, ok ? off : OOB_SENTINEL);
            }
            kseg += 64;
            if (kseg >= sg.len) { kseg = 0; ++si; }
        }
    };

    // The same tile issued in four parts (DMA instructions j = s, s+4, ... in part s; j < WI: weight rows, else activation
    // rows), one part in front of each 16-deep k-step's MFMAs: measured on MI355X (tools/gpu_bw_probe.py) one CU pulls
    // 50-60 B/clk from L2 by LDS-DMA when the requests are spread out, but a burst of 6-8 KiB-sized DMA instructions per
    // wave right after the barrier costs ~150 cycles of issue each with the matrix pipe idle.
    idmvton_seg sgc = p.seg[0];
    __amdgpu_buffer_rsrc_t rs_xc = rs_x0;
    auto issue_part = [&](int t, int buf, int s) {
        constexpr int LPT_ = WI + XI;
        char* dW = sW + buf * WBYTES;
        char* dX = sX + buf * XBYTES;
        if constexpr (!LIN) {
            if (s == 0) { sgc = p.seg[si]; rs_xc = make_rsrc(sgc.ptr, sgc.bytes); }
        }
#pragma unroll
        for (int j = 0; j < LPT_; ++j) {
            if ((j & 3) != s) continue;
            if (j < WI) {
                if (UNEVEN && wj(j) >= PW_) continue;
                dma16(rs_w, dW + wj(j) * 1024, w_off[j] + kbyte(t));
            } else {
                const int i = j - WI;
                if (UNEVEN && xj(i) >= PX_) continue;
                if constexpr (LIN) dma16(rs_x0, dX + xj(i) * 1024, x_off[i] + kbyte(t));
                else {
                    int iy = x_oy[i] + sgc.dy, ix = x_ox[i] + sgc.dx;
                    const bool ok = (unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win;
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    const uint32_t off = ((uint32_t)(x_pix[i] + iy * p.Wi + ix) * (uint32_t)sgc.pitch + sgc.coff + kseg + x_c8[i]) * 2u;
                    dma16(rs_xc, dX + xj(i) * 1024, ok ? off : OOB_SENTINEL);
                }
            }
        }
        if constexpr (!LIN) {
            if (s == 3) { kseg += 64; if (kseg >= sgc.len) { kseg = 0; ++si; } }
        }
    };

    // ---- fragment read addresses (row*128 and the row's swizzle key) ----
    int a_row[NI], a_swz[NI], b_row[MI], b_swz[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * SN + ni * 32 + l31; a_row[ni] = r * 128; a_swz[ni] = (r >> 1) & 7; }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * SM + mi * 32 + l31; b_row[mi] = r * 128; b_swz[mi] = (r >> 1) & 7; }

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    auto frag_a = [&](const char* bW, int s, int ni) { return *(const v8*)(bW + a_row[ni] + (((2 * s + u) ^ a_swz[ni]) << 4)); };
    auto frag_b = [&](const char* bX, int s, int mi) { return *(const v8*)(bX + b_row[mi] + (((2 * s + u) ^ b_swz[mi]) << 4)); };
    auto compute = [&](int buf, bool spread = false, int t_next = 0, int buf_next = 0) {
        const char* bW = sW + buf * WBYTES;
        const char* bX = sX + buf * XBYTES;
        if constexpr (!PF) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v8 a[NI], b[MI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) a[ni] = frag_a(bW, s, ni);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) b[mi] = frag_b(bX, s, mi);
                if constexpr (V1) {
                    if (spread) issue_part(t_next, buf_next, s);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = TR ? VT<T>::mfma(b[mi], a[ni], acc[ni][mi]) : VT<T>::mfma(a[ni], b[mi], acc[ni][mi]);
            }
        } else {
            // one wave per SIMD (no partner wave to cover the LDS latency): fragments double-buffered in registers, the
            // ds_reads of k-step s+1 are pinned ahead of the MFMAs of step s
            v8 a[2][NI], b[2][MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) a[0][ni] = frag_a(bW, 0, ni);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) b[0][mi] = frag_b(bX, 0, mi);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) a[(s + 1) & 1][ni] = frag_a(bW, s + 1, ni);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) b[(s + 1) & 1][mi] = frag_b(bX, s + 1, mi);
                }
                if (spread) issue_part(t_next, buf_next, s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[ni][mi] = TR ? VT<T>::mfma(b[s & 1][mi], a[s & 1][ni], acc[ni][mi])
                                         : VT<T>::mfma(a[s & 1][ni], b[s & 1][mi], acc[ni][mi]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // fused cross-attention (xattn.cuh): this wave's head and batch element; its K fragments travel under the main loop
    v8 xkf[XA ? XA_NK : 1];
    int xa_b = 0, xa_h = 0;
    if constexpr (XA) {
        static_assert(NI == 2 && !TR, "fused cross-attention: one 64-channel head per wave");
        const int mw = m0 + wm * SM;
        xa_b = (mw < p.M ? mw : p.M - 1) / p.xa.tokens;
        xa_h = (n0 + wn * SN) >> 6;
        xattn_load_k<T>(p.xa, xa_b, xa_h, lane, xkf);
    }
    const int nt = p.Ktot >> 6;
    if constexpr (!V1) {
        issue(0, 0);
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
            compute(t & 1);
        }
    } else {
        constexpr int LPT = WI + XI;                     // DMA instructions per wave per tile
        static_assert((ST - 2) * LPT < 64, "vmcnt immediate is 6 bits");
#pragma unroll
        for (int s = 0; s < ST - 1; ++s)
            if (s < nt) issue(s, s);
        int cbuf = 0, ibuf = ST - 1;                     // buffer computed this iteration / buffer refilled this iteration
        for (int t = 0; t < nt; ++t) {
            // tiles t .. t+ST-2 are outstanding (fewer at the tail); only tile t has to have landed
            if (t + ST - 2 < nt) wait_vmcnt<(ST - 2) * LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                // every wave's share of tile t landed; all are done with tile t-1
            asm volatile("" ::: "memory");
            compute(cbuf, t + ST - 1 < nt, t + ST - 1, ibuf);    // tile t+ST-1's DMA is issued in four parts between the k-steps
            cbuf = cbuf + 1 == ST ? 0 : cbuf + 1;
            ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
        }
    }

    if constexpr (XA) {                                  // the accumulators are q of one head per wave: replace them by the cross-attention output
        v8 xvf[XA_NV];
        xattn_load_v<T>(p.xa, xa_b, xa_h, lane, xvf);
        xattn_compute<T, MI>(p.xa, p.M, acc, m0 + wm * SM, lane, xkf, xvf);
    }
    gemm_epilogue<T, NI, MI, SN, SM, TR>(p, acc, m0, n0, wn, wm, lane);
}

// Tile configurations.  id = the `variant` field of tile_hint (bits 28..31); BN/BM in bits 16..27 / 0..15.
//   v0 (ST=2, 4 waves): 128x128, 128x64, 64x64           -- 2-3 blocks per CU hide the load latency
//   v1 (ring)         : 256x256 8 waves ST=2 (128 KiB, 32 B/clk/CU of operand traffic at MFMA peak),
//                       128x256 8 waves ST=3 (144 KiB, 1 block/CU), 128x128 4 waves ST=3 (96 KiB),
//                       128x64 4 waves ST=3 (72 KiB, 2 blocks/CU), 64x64 4 waves ST=4 (64 KiB, 2 blocks/CU)
//   (round 2 measured and removed two ideas: v3 = a per-tile K rotation, -0..-35 %; v4 = deep rings of 5-8 stages with one
//    block per CU, +-0 % on every shape: the operand stream is not bound by bytes in flight per CU but by the chip-wide L2 ->
//    LDS rate, ~10-12 TB/s with all 256 CUs streaming, so a launch's ceiling is its tile's arithmetic intensity times that.
//    A third: every launch touching the NEXT launch's weights (one dword per line, LDS-DMA into scratch) so they are cache
//    resident at its start -- -10..-32 % per launch in isolation behind a cache flush, +-0 in the pipeline (GEMM time per
//    denoising step 46.6 vs 46.3 ms, profiles/r02_bench_prefetch_ab.txt): removed.  A fourth: the next tile's DMA front-loaded
//    into the first one or two k-steps of the 8-wave tiles instead of spread over all four: -3..+3 %
//    (profiles/r02_probe_gemm_v4.log): removed.)
//   (also measured and removed in round 2, profiles/r02_gemm_ksweep_v3_w4_pp.json + r02_pmc_gemm_vs_hipblaslt.txt: a 256x256
//    ping-pong kernel -- five 32-deep LDS stages, counted vmcnt, two wave groups one barrier apart, s_setprio around 16-MFMA blocks --
//    raised MFMA-busy from 63 % to 75 % of the kernel's CYCLES on 8192^3, and the clock fell from 1.83 to 1.53 GHz: the same wall
//    time.  Large GEMMs on this part are power-limited; the library kernel it was compared with, hipBLASLt MT256x256x64 with 4 waves
//    of 128x128, is 87 % MFMA-busy at 1.5 GHz.  A 4-wave 128x128-per-wave instantiation of this file's loop had the same slope.)
//   v2                : the v1 tiles (except 128x128, which already has it) with the fragments of k-step s+1 read from LDS
//                       ahead of the MFMAs of step s (PMC: 45 % of wave cycles of the 8-wave tiles sit in s_waitcnt, mostly
//                       lgkmcnt in front of each k-step; both waves of a SIMD are barrier-aligned so neither covers the other)
template <typename T, int BN, int BM, int WN, int WM, int ST, bool V1, bool LIN, int OCC, bool PFX = false>
__global__ __launch_bounds__(WN * WM * 64, OCC) void gemm_conv_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[ST * (BN + BM) * 128];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tm, tn;
    if constexpr (!V1) { tm = wg / p.tiles_n; tn = wg - tm * p.tiles_n; }
    else {
        // grouped raster: the ~32-64 tiles an XCD runs concurrently form a ~1024 x 1024 output patch (GM m-tiles tall), so
        // the operand rows they share stay in that XCD's 4 MiB L2 instead of being re-fetched per tile row
        const int GM = p.gm > 0 ? p.gm : 1024 / BM;
        const int width = GM * p.tiles_n;
        const int grp = wg / width, rem = wg - grp * width;
        const int first = grp * GM;
        const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
        tn = rem / gsz; tm = first + (rem - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if constexpr (BN == 320) {                                 // never launched with a transposed part (launch_gemm)
        gemm_body<T, BN, BM, WN, WM, ST, V1, LIN, OCC == 1 || PFX, false>(p, smem, m0, n0);
    } else {
        if (p.vt != nullptr && n0 >= p.vt_n0) gemm_body<T, BN, BM, WN, WM, ST, V1, LIN, OCC == 1 || PFX, true>(p, smem, m0, n0);   // block-uniform
        else gemm_body<T, BN, BM, WN, WM, ST, V1, LIN, OCC == 1 || PFX, false>(p, smem, m0, n0);
    }
}

// The query projection of a cross-attention with the attention itself as its epilogue (xattn.cuh): plain Linear loader, ring pipeline.
template <typename T, int BN, int BM, int WN, int WM, int ST, int OCC>
__global__ __launch_bounds__(WN * WM * 64, OCC) void gemm_xattn_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[ST * (BN + BM) * 128];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int GM = p.gm > 0 ? p.gm : 1024 / BM;
    const int width = GM * p.tiles_n;
    const int grp = wg / width, rem = wg - grp * width;
    const int first = grp * GM;
    const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
    const int tn = rem / gsz, tm = first + (rem - tn * gsz);
    gemm_body<T, BN, BM, WN, WM, ST, true, true, OCC == 1, false, true>(p, smem, tm * BM, tn * BN);
}

template <typename T, int BN, int BM, int WN, int WM, int ST, bool V1, int OCC, bool PFX = false>
static void launch_cfg(const GemmParams& p0, bool lin, hipStream_t st) {
    GemmParams p = p0;
    p.gm = V1 ? idmvton_choose_gm(p.tiles_m, p.tiles_n, BM, BN, p.Ktot) : 0;
    const dim3 grid(p.tiles_n * p.tiles_m), block(WN * WM * 64);
    if constexpr (V1) {                                  // the v0 kernels keep the one general loader
        if (lin) { hipLaunchKernelGGL((gemm_conv_kernel<T, BN, BM, WN, WM, ST, true, true, OCC, PFX>), grid, block, 0, st, p); return; }
    }
    hipLaunchKernelGGL((gemm_conv_kernel<T, BN, BM, WN, WM, ST, V1, false, OCC, PFX>), grid, block, 0, st, p);
}


// Tile families, one translation unit each (gemm_tiles_*.hip).  Return 0 after launching, 1 when the family has no bn x bm tile.
int gemm_tiles_v0(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st);       // 2-stage 4-wave tiles 128x128, 128x64, 64x64
int gemm_tiles_v1(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st);       // LDS-ring tiles 256x256, 128x256, 128x128, 128x64, 64x64
int gemm_tiles_v2(const GemmParams& p, bool bf16, int bn, int bm, bool lin, hipStream_t st);       // ring tiles with register-prefetched fragments: 128x256, 64x64
int gemm_tiles_w8(const GemmParams& p, bool bf16, int bn, int bm, int form, bool lin, hipStream_t st);   // 8-wave 128x128 (3 forms), 320x256; 16-wave 256x256, 128x256
int gemm_tiles_xattn(const GemmParams& p, bool bf16, int bn, int bm, int w8, hipStream_t st);      // attn2.to_q + fused cross-attention (w8: 8-wave 128x128)
int launch_gemm_lin(const GemmParams& p, bool bf16, int bm, int form, int grid_cap, hipStream_t st);   // gemm_lin.hip (variant 5)
