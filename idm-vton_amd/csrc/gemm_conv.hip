// gemm_conv.hip -- MFMA implicit GEMM for every Linear / Conv2d of the IDM-VTON hot path (see include/idmvton_hip.h).
//
// D^T formulation: the MFMA "A" operand is the WEIGHT tile (rows n), the "B" operand the ACTIVATION tile (rows m), so
// each lane ends up with 4 consecutive output channels n for one output row m -> 8-byte vector stores, per-lane bias /
// residual vector loads, and a lane-local GEGLU (h and gate sit in the same lane, same register index).
//
// Data movement: both operands are K-contiguous in HBM (nn.Linear [N][K]; NHWC activations), so both tiles are
// [rows][64 k] = 128-byte rows.  They go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave
// instruction, no VGPR round trip); out-of-image conv taps, M/N tails and the fused zero padding all come from the
// buffer descriptor's bounds check (offset >= num_records reads 0), so the loader has no branches.
// LDS image: row r, 16-byte chunk c is stored at r*128 + ((c ^ ((r>>1)&7))<<4).  The DMA destination is lane-linear, so
// the XOR is applied to the per-lane SOURCE address (permutation inside one 128-byte line: still one cache line per
// 8 lanes) and again on the ds_read_b128 fragment reads, which makes every 16-lane read group hit 16 distinct 16-byte
// bank slots (conflict-free).
// Pipeline, two forms (template parameter ST = LDS stages):
//   ST == 2 : wait(tile t, vmcnt 0) -> __syncthreads -> issue DMA(tile t+1) -> MFMA(tile t)      (latency hidden by 2-3 blocks/CU)
//   ST >= 3 : ring of ST buffers, DMA runs ST-1 tiles ahead and stays in flight ACROSS the barrier: counted
//             `s_waitcnt vmcnt((ST-2)*loads_per_tile)` (only tile t must have landed) -> raw s_barrier -> issue
//             DMA(tile t+ST-1) into the buffer tile t-1 was read from (every wave is past compute(t-1) once it has passed
//             this barrier) -> MFMA(tile t).  Used with 8-wave 256x128 tiles (144 KiB LDS, 48 B/clk/CU of operand traffic
//             instead of 64 for 128x128) and with small tiles where 1-2 blocks per CU cannot hide the HBM/L2 latency.
// LIN = plain Linear (one K segment, no spatial gather): the activation operand's DMA offsets are precomputed like the
// weight's, so issuing a tile costs one add per DMA instead of the ~12 VALU of the conv gather.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "gemm_conv.hip: the in-launch LayerNorm hand-off and the counted vmcnt pipeline are written for gfx94x/gfx950 (stores counted in vmcnt)"
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
    constexpr int WI = BN / (8 * NW), XI = BM / (8 * NW); // DMA instructions per wave per tile (8 rows each)
    static_assert(WI >= 1 && XI >= 1 && WI * 8 * NW == BN && XI * 8 * NW == BM, "tile / wave-count mismatch");
    char* sW = smem;
    char* sX = smem + ST * WBYTES;

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int wn = wave / WM, wm = wave % WM;
    const int u = lane >> 5, l31 = lane & 31;

    // ---- loader state: this lane's rows / swizzled chunk for each DMA instruction it issues ----
    const int lrow = lane >> 3, lslot = lane & 7;
    uint32_t w_off[WI];                                  // byte offset of (row, chunk) in W, k0 excluded
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int R = (wave * WI + i) * 8 + lrow;
        const int c = lslot ^ ((R >> 1) & 7);
        w_off[i] = ((uint32_t)(n0 + R) * (uint32_t)p.Ktot + c * 8) * 2u;   // rows >= N fall beyond num_records -> 0
    }
    int x_pix[XI], x_oy[XI], x_ox[XI], x_c8[XI];
    uint32_t x_off[XI];                                  // LIN only
    const int HoWo = p.Ho * p.Wo;
    auto kbyte = [&](int t) -> uint32_t { return (uint32_t)t * 128u; };
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int R = (wave * XI + i) * 8 + lrow;
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
        char* dW = sW + buf * WBYTES + wave * (WI * 1024);
        char* dX = sX + buf * XBYTES + wave * (XI * 1024);
#pragma unroll
        for (int i = 0; i < WI; ++i) dma16(rs_w, dW + i * 1024, w_off[i] + kbyte(t));
        if constexpr (LIN) {
#pragma unroll
            for (int i = 0; i < XI; ++i)                  // OOB_SENTINEL + t*128 stays >= 2 GiB > num_records
                dma16(rs_x0, dX + i * 1024, x_off[i] + kbyte(t));
        } else {
            const idmvton_seg sg = p.seg[si];
            const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(sg.ptr, sg.bytes);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                int iy = x_oy[i] + sg.dy, ix = x_ox[i] + sg.dx;
                const bool ok = (unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                const uint32_t off = ((uint32_t)(x_pix[i] + iy * p.Wi + ix) * (uint32_t)sg.pitch + sg.coff + kseg + x_c8[i]) * 2u;
                dma16(rs_x, dX + i * 1024, ok ? off : OOB_SENTINEL);
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
        char* dW = sW + buf * WBYTES + wave * (WI * 1024);
        char* dX = sX + buf * XBYTES + wave * (XI * 1024);
        if constexpr (!LIN) {
            if (s == 0) { sgc = p.seg[si]; rs_xc = make_rsrc(sgc.ptr, sgc.bytes); }
        }
#pragma unroll
        for (int j = 0; j < LPT_; ++j) {
            if ((j & 3) != s) continue;
            if (j < WI) dma16(rs_w, dW + j * 1024, w_off[j] + kbyte(t));
            else {
                const int i = j - WI;
                if constexpr (LIN) dma16(rs_x0, dX + i * 1024, x_off[i] + kbyte(t));
                else {
                    int iy = x_oy[i] + sgc.dy, ix = x_ox[i] + sgc.dx;
                    const bool ok = (unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win;
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    const uint32_t off = ((uint32_t)(x_pix[i] + iy * p.Wi + ix) * (uint32_t)sgc.pitch + sgc.coff + kseg + x_c8[i]) * 2u;
                    dma16(rs_xc, dX + i * 1024, ok ? off : OOB_SENTINEL);
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

    // folded LayerNorm: this thread's row of the tile, (rstd, -rstd*mean) as its producer's last tile left it; one 8-byte load in
    // flight under the whole main loop (the per-tile fold of 20-40 partials this replaces cost 5-17 us per consumer launch)
    static_assert(NW * 64 >= BM, "one thread per tile row");
    float2 ln_ab = make_float2(1.f, 0.f);
    if (p.ln_rowstats && (int)threadIdx.x < BM && m0 + (int)threadIdx.x < p.M) ln_ab = ((const float2*)p.ln_rowstats)[m0 + threadIdx.x];

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

    const float* fin = nullptr;                          // LDS: (rstd, -rstd*mean) of this tile's rows, nullptr = no folded LayerNorm
    if (p.ln_rowstats) {                                 // block-uniform: LayerNorm of the activation operand folded into this GEMM
        __syncthreads();                                 // every wave is done with the last LDS stage
        if ((int)threadIdx.x < BM) ((float2*)smem)[threadIdx.x] = ln_ab;
        __syncthreads();
        fin = (const float*)smem;
    }
    if constexpr (XA) {                                  // the accumulators are q of one head per wave: replace them by the cross-attention output
        v8 xvf[XA_NV];
        xattn_load_v<T>(p.xa, xa_b, xa_h, lane, xvf);
        xattn_compute<T, MI>(p.xa, p.M, acc, m0 + wm * SM, lane, xkf, xvf);
    }
    gemm_epilogue<T, NI, MI, SN, SM, TR>(p, acc, m0, n0, wn, wm, lane, fin);

    if (p.rs_counter) {                                  // block-uniform: producer of LayerNorm row statistics
        // Inter-workgroup hand-off inside the launch (guide G16, "payload write-through + counter"): the partials left as agent-scope
        // stores; every wave drains them, one lane counts this tile on its row tile; the tile that arrives last reads the row tile's
        // partials back with agent-scope loads (never plain ones: this CU's L1 / this XCD's L2 may hold older lines of the same
        // addresses), folds them in a fixed order and leaves the counter zero for the next launch.
        constexpr int NT = NW * 64, TPR = NT / BM;       // threads per row
        static_assert(TPR >= 1 && TPR * BM == NT, "threads per row");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // also: every wave is done with `fin`
        unsigned* flag = (unsigned*)smem;
        const int tid = threadIdx.x;
        uint32_t* cnt = p.rs_counter + m0 / BM;
        if (tid == 0) {
            // acq_rel at agent scope: a release for this tile's partials (already written through and drained above; the fence makes
            // that a property of the memory model instead of the gfx9 vmcnt counting stores) and an acquire for the last arriver
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            *flag = old + 1u == (unsigned)p.tiles_n ? 1u : 0u;
        }
        __syncthreads();
        if (*flag) {
            float* scr = (float*)(smem + 16);            // [TPR][BM][2] partial folds
            const int r = tid % BM, part = tid / BM;
            const int P = p.rs_parts, chunk = (P + TPR - 1) / TPR;
            const int m = m0 + r;
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M) {
                const float* rs = p.rowstats_out + (size_t)m * P * 2;
                const int j1 = (part + 1) * chunk < P ? (part + 1) * chunk : P;
                for (int j = part * chunk; j < j1; j += 8) {   // eight loads in flight per batch, summed in index order
                    float2 t[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = j + q < j1 ? ld_agent_f2(rs + (j + q) * 2) : make_float2(0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) { s1 += t[q].x; s2 += t[q].y; }
                }
            }
            scr[(part * BM + r) * 2] = s1; scr[(part * BM + r) * 2 + 1] = s2;
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                double a1 = 0.0, a2 = 0.0;                    // E[x^2] - mean^2 cancels in fp32 on rows with a large mean: fold in double
#pragma unroll
                for (int q = 0; q < TPR; ++q) { a1 += (double)scr[(q * BM + tid) * 2]; a2 += (double)scr[(q * BM + tid) * 2 + 1]; }
                const double invc = 1.0 / (double)(P * 32);
                const double mean = a1 * invc;
                double var = a2 * invc - mean * mean;
                var = var > 0.0 ? var : 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)p.rs_eps));
                *(float2*)(p.rs_final + (size_t)(m0 + tid) * 2) = make_float2(rstd, (float)(-(double)rstd * mean));   // read by the NEXT launch: plain store
            }
            if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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
        constexpr int GM = 1024 / BM;
        const int width = GM * p.tiles_n;
        const int grp = wg / width, rem = wg - grp * width;
        const int first = grp * GM;
        const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
        tn = rem / gsz; tm = first + (rem - tn * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if constexpr (BN == 256 && BM == 256 && WN * WM == 4) {    // 128x128 per wave: never launched with a transposed part (launch_gemm)
        gemm_body<T, BN, BM, WN, WM, ST, V1, LIN, true, false>(p, smem, m0, n0);
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
    constexpr int GM = 1024 / BM;
    const int width = GM * p.tiles_n;
    const int grp = wg / width, rem = wg - grp * width;
    const int first = grp * GM;
    const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
    const int tn = rem / gsz, tm = first + (rem - tn * gsz);
    gemm_body<T, BN, BM, WN, WM, ST, true, true, OCC == 1, false, true>(p, smem, tm * BM, tn * BN);
}

template <typename T, int BN, int BM, int WN, int WM, int ST, bool V1, int OCC, bool PFX = false>
static void launch_cfg(const GemmParams& p, bool lin, hipStream_t st) {
    const dim3 grid(p.tiles_n * p.tiles_m), block(WN * WM * 64);
    if constexpr (V1) {                                  // the v0 kernels keep the one general loader
        if (lin) { hipLaunchKernelGGL((gemm_conv_kernel<T, BN, BM, WN, WM, ST, true, true, OCC, PFX>), grid, block, 0, st, p); return; }
    }
    hipLaunchKernelGGL((gemm_conv_kernel<T, BN, BM, WN, WM, ST, V1, false, OCC, PFX>), grid, block, 0, st, p);
}

// gemm_lin.hip: the hand-scheduled Linear main loop (variant 5: 256x256 as 8 waves x 128x64, 256x192 as 8 waves x 64x96)
int launch_gemm_lin(const GemmParams& p, bool bf16, int bm, int form, int grid_cap, hipStream_t st);

template <typename T>
static int launch_gemm(const GemmParams& p0, int variant, int bn, int bm, bool lin, hipStream_t st) {
    GemmParams p = p0;
    int form = 0;
    int grid_cap = 0;                                        // tests only (bit 14 of the BM field): 5 persistent workgroups, so small shapes walk several tiles each
    if (variant == 5) { form = bm & 15; grid_cap = (bm & 0x4000) ? 5 : 0; bm &= ~(15 | 0x4000); }   // placement form: low nibble of the BM field
    p.tiles_n = (p.N + bn - 1) / bn;
    p.tiles_m = (p.M + bm - 1) / bm;
    if (p.mode == IDMVTON_EPI_XATTN) {                   // tiles whose waves own 64 columns: 128x64, 128x128 (2x2 waves), 128x256 (2x4)
        if (bm != 64 && p.xa.tokens % 64 != 0) {         // a wave's rows must lie in one batch element: 64-row waves need tokens % 64 == 0
            bm = 64;
            p.tiles_m = (p.M + bm - 1) / bm;
        }
        const dim3 grid(p.tiles_n * p.tiles_m);
        if (bn == 128 && bm == 64) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 64, 2, 2, 3, 2>), grid, dim3(256), 0, st, p);
        else if (bn == 128 && bm == 128) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 128, 2, 2, 3, 1>), grid, dim3(256), 0, st, p);
        else if (bn == 128 && bm == 256) hipLaunchKernelGGL((gemm_xattn_kernel<T, 128, 256, 2, 4, 3, 2>), grid, dim3(512), 0, st, p);
        else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN runs on the 128x64, 128x128 and 128x256 tiles (got %dx%d)", bn, bm);
        CHECK_LAUNCH("gemm_conv");
        return IDMVTON_OK;
    }
    if (variant == 0) {
        if (bn == 128 && bm == 128) launch_cfg<T, 128, 128, 2, 2, 2, false, 2>(p, lin, st);
        else if (bn == 128 && bm == 64) launch_cfg<T, 128, 64, 2, 2, 2, false, 3>(p, lin, st);
        else if (bn == 64 && bm == 64) launch_cfg<T, 64, 64, 2, 2, 2, false, 3>(p, lin, st);
        else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unsupported v0 tile %dx%d", bn, bm);
    } else if (variant == 1) {
        if (bn == 256 && bm == 256) launch_cfg<T, 256, 256, 2, 4, 2, true, 2>(p, lin, st);
        else if (bn == 128 && bm == 256) launch_cfg<T, 128, 256, 2, 4, 3, true, 2>(p, lin, st);
        else if (bn == 128 && bm == 128) launch_cfg<T, 128, 128, 2, 2, 3, true, 1>(p, lin, st);
        else if (bn == 128 && bm == 64) launch_cfg<T, 128, 64, 2, 2, 3, true, 2>(p, lin, st);
        else if (bn == 64 && bm == 64) launch_cfg<T, 64, 64, 2, 2, 4, true, 2>(p, lin, st);
        else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unsupported v1 tile %dx%d", bn, bm);
    } else if (variant == 2) {                           // ring tiles with the register-prefetched fragment pipeline on every tile
        if (bn == 256 && bm == 256) launch_cfg<T, 256, 256, 2, 4, 2, true, 2, true>(p, lin, st);
        else if (bn == 128 && bm == 256) launch_cfg<T, 128, 256, 2, 4, 3, true, 2, true>(p, lin, st);
        else if (bn == 128 && bm == 64) launch_cfg<T, 128, 64, 2, 2, 3, true, 2, true>(p, lin, st);
        else if (bn == 64 && bm == 64) launch_cfg<T, 64, 64, 2, 2, 4, true, 2, true>(p, lin, st);
        else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unsupported v2 tile %dx%d", bn, bm);
    } else if (variant == 3) {                           // 256x256, 4 waves x (128x128): one wave per SIMD, accumulators in AGPRs, a third fewer
        if (!(bn == 256 && bm == 256)) return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: variant 3 is the 256x256 tile");   // LDS fragment reads
        if (p.wide && !p.vt && !p.out8) launch_cfg<T, 256, 256, 2, 2, 2, true, 1>(p, lin, st);
        else launch_cfg<T, 256, 256, 2, 4, 2, true, 2>(p, lin, st);              // 8-byte epilogue, a V^T part or e4m3 output: the 8-wave tile
    } else if (variant == 5) {                           // hand-scheduled Linear loop; anything it does not cover runs on the 8-wave ring tile
        if (!(bn == 256 && (bm == 256 || bm == 192))) return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: variant 5 is the 256x256 / 256x192 tile");
        if (lin && !p.ln_rowstats && !p.rs_counter) launch_gemm_lin(p, std::is_same<T, bf16_t>::value, bm, form, grid_cap, st);
        else if (bm == 256) launch_cfg<T, 256, 256, 2, 4, 2, true, 2>(p, lin, st);
        else {                                           // 256x192 exists only hand-scheduled: the same launch on the 128x256 ring tile
            p.tiles_n = (p.N + 127) / 128; p.tiles_m = (p.M + 255) / 256;
            launch_cfg<T, 128, 256, 2, 4, 3, true, 2>(p, lin, st);
        }
    } else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unknown variant %d", variant);
    CHECK_LAUNCH("gemm_conv");
    return IDMVTON_OK;
}

extern "C" int idmvton_gemm_conv(const idmvton_gemm_conv_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "gemm_conv: null args");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "gemm_conv: dtype %d", a->dtype);
    CHECK_ARG(a->M > 0 && a->N > 0 && a->Ktot > 0, IDMVTON_E_SHAPE, "gemm_conv: M=%d N=%d K=%d", a->M, a->N, a->Ktot);
    CHECK_ARG(a->Ktot % 64 == 0, IDMVTON_E_SHAPE, "gemm_conv: Ktot=%d must be a multiple of 64", a->Ktot);
    CHECK_ARG(a->N % 4 == 0, IDMVTON_E_SHAPE, "gemm_conv: N=%d must be a multiple of 4", a->N);
    CHECK_ARG(a->nseg >= 1 && a->nseg <= IDMVTON_MAX_SEG, IDMVTON_E_SHAPE, "gemm_conv: nseg=%d", a->nseg);
    int ksum = 0;
    for (int s = 0; s < a->nseg; ++s) {
        const idmvton_seg& g = a->seg[s];
        CHECK_ARG(g.ptr && g.len > 0 && g.len % 64 == 0, IDMVTON_E_SHAPE, "gemm_conv: seg %d len=%d", s, g.len);
        CHECK_ARG(g.pitch % 8 == 0 && g.coff % 8 == 0, IDMVTON_E_ALIGN, "gemm_conv: seg %d pitch=%d coff=%d", s, g.pitch, g.coff);
        CHECK_ARG(g.bytes < 0x80000000u, IDMVTON_E_SHAPE, "gemm_conv: seg %d tensor >= 2 GiB", s);
        CHECK_ARG(((uintptr_t)g.ptr & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: seg %d pointer not 16-byte aligned", s);
        ksum += g.len;
    }
    CHECK_ARG(ksum == a->Ktot, IDMVTON_E_SHAPE, "gemm_conv: segment lengths sum to %d, Ktot=%d", ksum, a->Ktot);
    CHECK_ARG(a->Ho > 0 && a->Wo > 0 && a->Hi > 0 && a->Wi > 0 && a->stride > 0 && a->M % (a->Ho * a->Wo) == 0,
              IDMVTON_E_SHAPE, "gemm_conv: geometry M=%d Ho=%d Wo=%d", a->M, a->Ho, a->Wo);
    if (a->ups) CHECK_ARG(a->stride == 1 && a->Ho <= 2 * a->Hi && a->Ho >= 2 * a->Hi - 1 && a->Wo <= 2 * a->Wi && a->Wo >= 2 * a->Wi - 1, IDMVTON_E_SHAPE,
                          "gemm_conv: ups=1 needs stride 1 and an output grid of 2Hi x 2Wi or one short of it (Ho=%d Hi=%d Wo=%d Wi=%d)", a->Ho, a->Hi, a->Wo, a->Wi);
    CHECK_ARG(a->w && ((uintptr_t)a->w & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: weight pointer");
    const uint64_t wbytes = (uint64_t)a->N * a->Ktot * 2;
    CHECK_ARG(wbytes + (uint64_t)128 * a->Ktot * 2 < 0xFFFFFFFFull, IDMVTON_E_SHAPE, "gemm_conv: weight too large");
    const bool geglu = a->mode == IDMVTON_EPI_GEGLU, xattn = a->mode == IDMVTON_EPI_XATTN;
    CHECK_ARG(a->mode == IDMVTON_EPI_NONE || a->mode == IDMVTON_EPI_GELU || a->mode == IDMVTON_EPI_QUICKGELU || geglu || xattn, IDMVTON_E_ARG, "gemm_conv: mode %d", a->mode);
    CHECK_ARG(a->colscale_n >= 0 && a->colscale_n % 4 == 0 && !(geglu && a->colscale_n), IDMVTON_E_ARG, "gemm_conv: colscale_n=%d", a->colscale_n);
    if (geglu) CHECK_ARG(a->N % 64 == 0 && !a->res && !a->rowbias && !a->vt, IDMVTON_E_ARG, "gemm_conv: GEGLU needs N%%64==0, no res/rowbias/vt");
    CHECK_ARG(a->out || (a->vt && a->vt_n0 == 0), IDMVTON_E_ARG, "gemm_conv: null out");
    if (a->out) CHECK_ARG(a->ldo % 4 == 0 && ((uintptr_t)a->out & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: out alignment");
    if (a->res) CHECK_ARG(a->ldr % 4 == 0 && ((uintptr_t)a->res & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: res alignment");
    if (a->bias) CHECK_ARG(((uintptr_t)a->bias & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: bias alignment");
    if (a->rowbias) CHECK_ARG(a->rowbias_ld % 4 == 0 && a->rows_per_group > 0 && ((uintptr_t)a->rowbias & 7) == 0,
                              IDMVTON_E_ALIGN, "gemm_conv: rowbias");
    if (a->vt && a->vt_perm) CHECK_ARG(((uintptr_t)a->vt & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: vt pointer not 16-byte aligned");
    if (a->vt && a->vt_perm) CHECK_ARG(a->vt_tokens % 16 == 0, IDMVTON_E_ARG, "gemm_conv: vt_perm needs vt_tokens %% 16 == 0 (got %d)", a->vt_tokens);
    if (a->vt) CHECK_ARG(a->vt_tokens > 0 && a->vt_tokens % 4 == 0 && a->M % a->vt_tokens == 0 && a->vt_n0 % 64 == 0 &&
                         a->vt_n0 >= 0 && a->vt_n0 < a->N && ((uintptr_t)a->vt & 7) == 0,
                         IDMVTON_E_ARG, "gemm_conv: vt_tokens=%d vt_n0=%d", a->vt_tokens, a->vt_n0);

    GemmParams p;
    memset(&p.xa, 0, sizeof(p.xa));
    if (xattn) {
        const idmvton_xattn* x = a->xattn;
        CHECK_ARG(x != nullptr && (x->nseg == 1 || x->nseg == 2), IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN needs `xattn` with 1 or 2 key segments");
        CHECK_ARG(a->N % 64 == 0 && !a->bias && !a->res && !a->rowbias && !a->vt && !a->colscale_n && !a->io_flags && !a->ln_rowstats && !a->rowstats_out &&
                  a->nseg == 1 && a->Ho == 1 && a->Hi == 1 && a->Wo == a->M && a->Wi == a->M && a->out && a->ldo % 8 == 0 && ((uintptr_t)a->out & 15) == 0,
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN is a plain Linear with N %% 64 == 0 and nothing else in its epilogue");
        CHECK_ARG(x->tokens > 0 && x->tokens % 32 == 0 && a->M % x->tokens == 0, IDMVTON_E_SHAPE, "gemm_conv: xattn.tokens=%d (rows per batch element, a multiple of 32 dividing M=%d)", x->tokens, a->M);
        for (int s = 0; s < x->nseg; ++s) {
            const int nk_max = s == 0 ? 96 : 32;         // xattn.cuh holds 3 K blocks / 6 V^T steps for segment 0 (text) and 1 / 2 for segment 1 (image prompt)
            CHECK_ARG(x->k[s] && x->vt[s] && x->nk[s] > 0 && x->nk[s] <= nk_max && x->k_rows[s] >= ((x->nk[s] + 31) & ~31) && x->ldk[s] >= a->N && x->ldk[s] % 8 == 0 &&
                      x->ldvt[s] >= ((x->nk[s] + 15) & ~15) && x->ldvt[s] % 8 == 0 && (((uintptr_t)x->k[s] | (uintptr_t)x->vt[s]) & 15) == 0, IDMVTON_E_SHAPE,
                      "gemm_conv: xattn segment %d: nk=%d (<= %d) k_rows=%d (>= round32(nk)) ldk=%d ldvt=%d (>= round16(nk))", s, x->nk[s], nk_max, x->k_rows[s], x->ldk[s], x->ldvt[s]);
            p.xa.k[s] = x->k[s]; p.xa.vt[s] = x->vt[s]; p.xa.ldk[s] = x->ldk[s]; p.xa.ldvt[s] = x->ldvt[s]; p.xa.nk[s] = x->nk[s]; p.xa.krows[s] = x->k_rows[s];
        }
        p.xa.nseg = x->nseg; p.xa.tokens = x->tokens; p.xa.vchan = a->N; p.xa.ip_scale = x->ip_scale;
    } else CHECK_ARG(a->xattn == nullptr, IDMVTON_E_ARG, "gemm_conv: `xattn` without mode IDMVTON_EPI_XATTN");
    p.w = a->w; p.w_bytes = (uint32_t)wbytes; p.N = a->N; p.Ktot = a->Ktot;
    p.nseg = a->nseg;
    for (int s = 0; s < IDMVTON_MAX_SEG; ++s) p.seg[s] = a->seg[s < a->nseg ? s : a->nseg - 1];
    p.M = a->M; p.Ho = a->Ho; p.Wo = a->Wo; p.Hi = a->Hi; p.Wi = a->Wi; p.stride = a->stride; p.ups = a->ups ? 1 : 0;
    p.out = a->out; p.ldo = a->ldo; p.bias = a->bias; p.rowbias = a->rowbias; p.rowbias_ld = a->rowbias_ld;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1; p.res = a->res; p.ldr = a->ldr; p.mode = a->mode;
    p.vt = a->vt; p.vt_n0 = a->vt_n0; p.vt_tokens = a->vt_tokens > 0 ? a->vt_tokens : 4; p.vt_perm = a->vt_perm ? 1 : 0;
    p.colscale_n = a->colscale_n; p.colscale = a->colscale;
    p.tiles_m = p.tiles_n = 0;
    {
        auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
        const int n_out = geglu ? a->N / 2 : a->N;
        p.wide = n_out % 8 == 0 && a->N % 8 == 0 && a->colscale_n % 8 == 0 && (!a->out || (a->ldo % 8 == 0 && a16(a->out))) &&
                 (!a->res || (a->ldr % 8 == 0 && a16(a->res))) && (!a->bias || a16(a->bias)) &&
                 (!a->rowbias || (a->rowbias_ld % 8 == 0 && a16(a->rowbias))) && (!a->vt || a->vt_n0 % 8 == 0);
    }

    // Tile choice: tile_hint (variant<<28 | BN<<16 | BM) from the caller's tuning table; GEGLU needs 64-row wave tiles (BN >= 128).
    int variant = 1, bn = 64, bm = 64;
    static const bool env_narrow = getenv("IDMVTON_EPILOGUE_8B") != nullptr;   // measurement only (A/B of the whole pipeline)
    if ((a->tile_hint & 0x8000) || env_narrow) p.wide = 0;                      // measurement only: force the 8-byte epilogue
    p.rowstats_out = a->rowstats_out; p.rs_parts = a->N / 32;
    p.rs_final = a->rowstats_final; p.rs_counter = a->rowstats_counter; p.rs_eps = a->rowstats_eps;
    p.ln_rowstats = a->ln_rowstats; p.ln_colvec = a->ln_colvec;
    if (a->rowstats_out || a->rowstats_final || a->rowstats_counter) {
        CHECK_ARG(a->rowstats_out && a->rowstats_final && a->rowstats_counter && a->rowstats_eps > 0.f, IDMVTON_E_ARG,
                  "gemm_conv: rowstats_out, rowstats_final, rowstats_counter and rowstats_eps > 0 come together");
        CHECK_ARG(a->N % 32 == 0 && !geglu && !a->vt && a->out && p.wide && !(a->io_flags & IDMVTON_IO_OUT_F32) &&
                  ((uintptr_t)a->rowstats_out & 7) == 0 && ((uintptr_t)a->rowstats_final & 7) == 0 && ((uintptr_t)a->rowstats_counter & 3) == 0,
                  IDMVTON_E_ARG, "gemm_conv: rowstats_out needs N %% 32 == 0 (N=%d), the plain 16-byte epilogue, a 16-bit out", a->N);
    }
    if (a->ln_rowstats) {
        CHECK_ARG(a->ln_colvec && a->nseg == 1 && ((uintptr_t)a->ln_rowstats & 7) == 0 && ((uintptr_t)a->ln_colvec & 15) == 0,
                  IDMVTON_E_ARG, "gemm_conv: folded LayerNorm needs one K segment (nseg=%d) and ln_colvec", a->nseg);
    } else CHECK_ARG(!a->ln_colvec, IDMVTON_E_ARG, "gemm_conv: ln_colvec without ln_rowstats");
    p.res32 = (a->io_flags & IDMVTON_IO_RES_F32) ? 1 : 0;
    p.out32 = (a->io_flags & IDMVTON_IO_OUT_F32) ? 1 : 0;
    p.bias32 = (a->io_flags & IDMVTON_IO_BIAS_F32) ? 1 : 0;
    if (p.bias32) CHECK_ARG(a->bias && p.wide && !geglu && !a->vt && ((uintptr_t)a->bias & 15) == 0, IDMVTON_E_ARG,
                            "gemm_conv: IDMVTON_IO_BIAS_F32 needs a 16-byte aligned bias and the plain 16-byte epilogue (no GEGLU, no vt)");
    CHECK_ARG((a->io_flags & ~15) == 0, IDMVTON_E_ARG, "gemm_conv: io_flags=%d", a->io_flags);
    p.out8 = (a->io_flags & IDMVTON_IO_OUT_F8) ? 1 : 0;
    p.o8_scale = a->f8_out_scale; p.vt8_scale = a->f8_vt_scale;
    if (p.out8) {
        CHECK_ARG(a->io_flags == IDMVTON_IO_OUT_F8 && a->mode == IDMVTON_EPI_NONE && !a->res && !a->rowbias && !a->ln_rowstats && !a->rowstats_out && p.wide,
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_OUT_F8 needs the plain 16-byte epilogue (no activation / residual / rowbias / fp32 IO / LayerNorm fold)");
        CHECK_ARG((!a->out || a->f8_out_scale > 0.f) && (!a->vt || (a->f8_vt_scale > 0.f && a->vt_tokens % 64 == 0 && ((uintptr_t)a->vt & 15) == 0)),
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_OUT_F8: scales > 0, vt_tokens %% 64 == 0 (got %d), 16-byte aligned vt", a->vt_tokens);
    }
    if (p.res32 || p.out32) {
        CHECK_ARG((a->io_flags & ~7) == 0 && !geglu && !a->vt, IDMVTON_E_ARG, "gemm_conv: io_flags=%d (fp32 res / out: no GEGLU, no vt)", a->io_flags);
        CHECK_ARG(!p.res32 || a->res, IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_RES_F32 without res");
        CHECK_ARG(p.wide, IDMVTON_E_ALIGN, "gemm_conv: the fp32 residual stream needs the 16-byte epilogue (N, ldo, ldr multiples of 8, 16-byte aligned pointers)");
    }
    if (a->tile_hint) { variant = (a->tile_hint >> 28) & 0xf; bn = (a->tile_hint >> 16) & 0xfff; bm = a->tile_hint & 0x7fff; }
    else {
        // No hint: the largest ring tile that still gives every CU a tile (measured rule, profiles/r01_tune_report_*.json:
        // operand delivery per CU is the bound, so arithmetic intensity per tile wins until the grid no longer fills 256 CUs).
        static const int cand[5][2] = {{256, 256}, {128, 256}, {128, 128}, {128, 64}, {64, 64}};
        for (int i = 0; i < 5; ++i) {
            const int n_ = cand[i][0], m_ = cand[i][1];
            if (geglu && n_ < 128) continue;
            if (xattn && (n_ != 128)) continue;          // waves must own 64 columns (one head): the 128-column tiles
            if (a->vt && a->vt_n0 % n_ != 0) continue;
            // a tile wider than the (64-rounded) problem wastes its surplus columns' MFMAs: N = 128 on the 256-column tile ran the
            // VAE's full-resolution 128-channel convolutions at half rate (profiles/r03_v2_prof_default_kernel_stats_by_grid.txt)
            if (i < 4 && (n_ > ((a->N + 63) & ~63) || m_ > ((a->M + 63) & ~63))) continue;
            bn = n_; bm = m_;
            if ((long)((a->N + n_ - 1) / n_) * ((a->M + m_ - 1) / m_) >= 200) break;
        }
    }
    if (geglu) CHECK_ARG(bn >= 128, IDMVTON_E_ARG, "gemm_conv: GEGLU needs BN >= 128 (64-row wave tiles)");
    if (a->vt) CHECK_ARG(a->vt_n0 % bn == 0, IDMVTON_E_ARG, "gemm_conv: vt_n0 %% BN != 0");
    const idmvton_seg& s0 = a->seg[0];
    const bool lin = a->nseg == 1 && a->Ho == 1 && a->Hi == 1 && a->Wo == a->M && a->Wi == a->M && a->stride == 1 &&
                     !a->ups && s0.dy == 0 && s0.dx == 0 && (uint64_t)a->M * s0.pitch * 2 < 0x80000000ull;
    // no hint and the heuristic chose the 256x256 tile for a plain Linear: the hand-scheduled loop (gemm_lin.hip) won on every such shape it
    // was measured on (profiles/r04_gemm_probe_*.log: +5..11 % over the compiler-scheduled tile)
    if (!a->tile_hint && bn == 256 && bm == 256 && lin && !a->ln_rowstats && !a->rowstats_counter) { variant = 5; bm = 257; }
    hipStream_t st = (hipStream_t)stream;
    return a->dtype == IDMVTON_BF16 ? launch_gemm<bf16_t>(p, variant, bn, bm, lin, st) : launch_gemm<f16_t>(p, variant, bn, bm, lin, st);
}
