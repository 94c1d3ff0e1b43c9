// attention.hip -- flash attention for head_dim 64 on MFMA 32x32x16 (see include/idmvton_hip.h, idmvton_attn_fwd).
//
// Swapped formulation: S^T = K.Q^T and O^T = V^T.P^T, so every lane owns ONE query row (q = lane&31; the two half-waves
// hold different keys of that row).  Row max / row sum / the O rescale are then lane-local (one cross-half exchange per
// tile for the max), and P goes from the S^T accumulator straight into the PV MFMA "B" operand with no shuffle: the
// key <-> k-slot assignment of the PV contraction is chosen to be exactly the one the QK^T accumulator already has
// (slot jj of half-wave u  <->  key 16*step + (jj&3) + 8*(jj>>2) + 4u).  V arrives already transposed AND in that key order
// ([channel][position], position = key with bits 2 and 3 swapped inside each group of 16; written by the projection GEMM's
// vt epilogue with vt_perm = 1), so the 8 keys of a half-wave's k-slots are one 16-byte ds_read_b128 and K tiles ([key][64 d])
// and V^T tiles ([64 d][64 positions]) are both 64 rows x 128 B and share one LDS-DMA loader, one XOR swizzle and one
// conflict-free read pattern
// (row r, 16-byte chunk c stored at r*128 + ((c ^ ((r>>1)&7))<<4); swizzle applied on the DMA source address).
// Key segments: SELF mode walks up to two segments under one softmax (own tokens, garment tokens); a segment that is
// absent for this batch element (CFG-unconditional half: all-zero garment features) contributes nk keys with logit 0 and
// value 0 in closed form (m0 = 0, l0 = nk).  CROSS mode keeps the two segments as separate softmaxes and sums the outputs.
// Pipeline: ST LDS stages of 16 KiB, one barrier per 64-key tile.  ST == 2: wait(t, vmcnt 0) -> __syncthreads -> DMA(t+1) ->
// MFMA/softmax(t).  ST >= 3: ring, the DMA runs ST-1 tiles ahead and stays in flight across the raw s_barrier (counted
// vmcnt: only tile t must have landed), see gemm_conv.hip.  The O / l rescale is skipped (wave-uniform branch) when no
// lane's running max grew in this tile: alpha would be exactly 1, so the result is unchanged.
#include "common.cuh"

struct AttnParams {
    int B, heads, Nq;
    const void* q; int ldq; uint32_t qbytes;
    void* out; int ldo;
    int nseg;
    const void* k[2]; int ldk[2]; uint32_t kbytes[2];
    const void* vt[2]; int ldvt[2]; uint32_t vtbytes[2];
    int nk[2]; int krows[2]; int seg_b0[2];
    float ip_scale;
    int nqb, chunk, parts;
    int pp_flags; float pp_thr;
    int q_prescaled;
};

#define NEG_BIG (-1.0e30f)


// Work order.  A (batch, head) group costs in proportion to the key segments present for its batch (the CFG-unconditional
// batches b < seg_b0 skip the garment segment: half the tiles), and block id i runs on XCD i % 8 in id order.  Groups are
// therefore ordered longest first (descending b), each group's q-blocks cut into `parts` chunks of `chunk` q-blocks (a unit:
// its workgroups walk the same K/V, so they share it through one XCD's L2), and the units are dealt round-robin to the XCDs:
// every XCD -- and, since all of an XCD's workgroups are dispatched in id order, every CU -- gets the same mix of long and
// short work with the long work first.  (The previous contiguous split gave XCDs 0-3 only unconditional batches: they idled
// for half of the launch.)  Ids beyond the last unit / q-block are padding and exit.
__device__ __forceinline__ bool attn_work(const AttnParams& p, int bid, int& b, int& h, int& qb) {
    const int x = bid & 7, j = bid >> 3;
    const int u = j / p.chunk, r = j - u * p.chunk;
    const int U = u * 8 + x;
    const int g = U / p.parts, part = U - g * p.parts;
    qb = part * p.chunk + r;
    b = p.B - 1 - g / p.heads;
    h = g % p.heads;
    return g < p.B * p.heads && qb < p.nqb;
}
static void attn_grid(AttnParams& p, int qrows, dim3& grid) {
    p.nqb = (p.Nq + qrows - 1) / qrows;
    p.chunk = p.nqb >= 8 ? (p.nqb + 1) / 2 : 1;
    p.parts = (p.nqb + p.chunk - 1) / p.chunk;
    const int units = p.B * p.heads * p.parts;
    grid = dim3(8 * ((units + 7) / 8) * p.chunk);
}

// Work order of attn_sp_kernel.  Two cost classes: LONG (b >= lb0: every segment present) and SHORT (b < lb0: the second segment is absent, half
// the tiles).  The q-blocks of one (batch, head) group walk the same K/V; a group is cut into `parts` runs of `chunk` consecutive q-blocks (a unit:
// its workgroups go to ONE XCD and share K/V through that L2; parts = 8 / gcd(groups, 8), so that every XCD gets the same number of units of a
// class), unit U of its class goes to XCD U % 8 (block id i runs on XCD i % 8), and inside every XCD's id sequence long and short workgroups
// ALTERNATE -- kind(j) = (j ^ (j >> 5)) & 1 for the j-th workgroup of the XCD, so neighbours in dispatch order (j, j + 1) and workgroups one
// CU-round apart (j, j + 32) differ -- because with two workgroups resident per CU the hardware pairs ids by a rule HIP does not promise;
// alternating makes every pairing long + short, i.e. every CU carries the same work (the longest-first order of attn_work gave CUs two long or
// two short walks: no gain over one workgroup per CU).  When one class runs out on an XCD the other continues alone.  Ids beyond an XCD's count
// (and q-blocks beyond the last) are padding and exit.
__device__ __forceinline__ bool sp_work(const AttnParams& p, int bid, int& b, int& h, int& qb) {
    const int x = bid & 7, j = bid >> 3;
    const int nqb = p.nqb, lb0 = p.chunk;               // (chunk carries lb0, parts the two classes' part counts for this kernel)
    const int pl = p.parts & 0xffff, psh = p.parts >> 16;
    const int chl = (nqb + pl - 1) / pl, chs = (nqb + psh - 1) / psh;
    const int ngl = (p.B - lb0) * p.heads, ngs = lb0 * p.heads;
    const int cl = ((ngl * pl - x + 7) >> 3) * chl, cs = ((ngs * psh - x + 7) >> 3) * chs;     // this XCD's long / short workgroups
    const int m = cl < cs ? cl : cs;
    // ids with kind 0 take long, kind 1 short, while both classes last (2m ids; every aligned pair holds one of each kind, so id j < 2m is the
    // (j >> 1)-th of its kind)
    int t; bool lng;
    if (!(p.pp_flags & 1)) { lng = j < cl; t = lng ? j : j - cl; }      // more workgroups than the chip holds at once: longest first
    else if (j < 2 * m) { lng = (((j ^ (j >> 5)) & 1) == 0); t = j >> 1; }
    else { lng = cl > cs; t = j - m; }
    if (t >= (lng ? cl : cs)) return false;
    const int ch = lng ? chl : chs, pa = lng ? pl : psh;
    const int U = (t / ch) * 8 + x, g = U / pa;
    qb = (U - g * pa) * ch + (t - (t / ch) * ch);
    b = (lng ? lb0 : 0) + g / p.heads;
    h = g % p.heads;
    return qb < nqb;
}
static void sp_grid(AttnParams& p, int qrows, dim3& grid) {   // (also sets pp_flags bit 0: see sp_work)
    p.nqb = (p.Nq + qrows - 1) / qrows;
    int lb0 = p.nseg > 1 ? p.seg_b0[1] : 0;             // batches below it skip the second segment
    if (p.nseg > 0 && p.seg_b0[0] > lb0) lb0 = p.seg_b0[0];
    if (lb0 > p.B) lb0 = p.B;
    auto parts_of = [&](int ng) {
        if (ng <= 0) return 1;
        int g8 = 8;
        while (ng % g8) g8 >>= 1;                        // gcd(ng, 8)
        int pa = 8 / g8;
        return pa < p.nqb ? pa : p.nqb;
    };
    const int ngl = (p.B - lb0) * p.heads, ngs = lb0 * p.heads;
    const int pl = parts_of(ngl), psh = parts_of(ngs);
    p.chunk = lb0; p.parts = pl | (psh << 16);
    const int chl = (p.nqb + pl - 1) / pl, chs = (p.nqb + psh - 1) / psh;
    grid = dim3(8 * (((ngl * pl + 7) / 8) * chl + ((ngs * psh + 7) / 8) * chs));
    p.pp_flags = (qrows == 128 && grid.x <= 512) ? 1 : 0;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int MODE, int NWAVES, int ST>
__global__ __launch_bounds__(NWAVES * 64) void attn_kernel(const AttnParams p) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int QB = 32 * NWAVES;
    constexpr int IPW = 16 / NWAVES;                    // DMA instructions per wave per KV tile (8 K + 8 V^T in total)
    constexpr int STAGE = 16384;                        // one 64-key tile per LDS stage = per barrier
    constexpr int KT = 1;
    // [stage][K 8 KiB | V^T 8 KiB]; ring kernels add a per-wave 4 KiB Q tile (Q also arrives by LDS-DMA there: a plain
    // global load of Q before the loop makes hipcc re-wait for it -- vmcnt(0) -- inside every iteration, draining the ring)
    __shared__ __attribute__((aligned(1024))) char smem[ST * STAGE + (ST > 2 ? NWAVES * 4096 : 0)];

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int u = lane >> 5, l31 = lane & 31;

    int b, h, qb;
    if (!attn_work(p, blockIdx.x, b, h, qb)) return;     // block-uniform

    // ---- Q fragments (MFMA B operand): column q = l31, k = d in [16s + 8u, +8) ----
    const int q_row = qb * QB + wave * 32 + l31;
    const int q_ld = q_row < p.Nq ? q_row : p.Nq - 1;
    v8 qf[4];
    if constexpr (ST == 2) {
        const T* qp = (const T*)p.q + ((size_t)b * p.Nq + q_ld) * p.ldq + h * 64 + 8 * u;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const v8*)(qp + 16 * s);
    } else {
        // Q tile of this wave: 32 rows x 128 B -> LDS (4 DMA instructions, same swizzle as K); read back after the first wait
        const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(p.q, p.qbytes);
        char* dq = smem + ST * STAGE + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((R >> 1) & 7);
            int row = qb * QB + wave * 32 + R;
            row = row < p.Nq ? row : p.Nq - 1;
            dma16(rs_q, dq + i * 1024, (uint32_t)((((size_t)b * p.Nq + row) * p.ldq + h * 64 + c * 8) * 2));
        }
    }

    // ---- which segments exist for this batch element (block-uniform) ----
    const bool pres0 = p.nseg > 0 && b >= p.seg_b0[0];
    const bool pres1 = p.nseg > 1 && b >= p.seg_b0[1];
    const int nt0 = pres0 ? (p.nk[0] + 63) >> 6 : 0;
    const int nt1 = pres1 ? (p.nk[1] + 63) >> 6 : 0;
    const int nt = nt0 + nt1;

    // ---- loader ----
    const int lrow = lane >> 3, lslot = lane & 7;
    auto issue = [&](int t, char* tile) {
        const int sg = t < nt0 ? 0 : 1;
        const int kt = sg ? t - nt0 : t;
        const int nk = p.nk[sg];
        const int bsg = b - p.seg_b0[sg];
        char* dst = tile + wave * (IPW * 1024);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int j = wave * IPW + i;               // wave-uniform: 0..7 -> K rows, 8..15 -> V^T rows
            const int R = (j & 7) * 8 + lrow;
            const int c = lslot ^ ((R >> 1) & 7);
            if (j < 8) {
                const int key = kt * 64 + R;
                const uint32_t off = (uint32_t)((((size_t)bsg * p.krows[sg] + key) * p.ldk[sg] + h * 64 + c * 8) * 2);
                dma16(make_rsrc(p.k[sg], p.kbytes[sg]), dst + i * 1024, key < nk ? off : OOB_SENTINEL);
            } else {
                // chunk c of V^T row R holds tile positions [8c, 8c+8) = keys 16(c>>1) + 4(c&1) + {0..3, 8..11} (key order)
                const int kmin = kt * 64 + 16 * (c >> 1) + 4 * (c & 1);
                const uint32_t off = (uint32_t)((((size_t)bsg * p.heads * 64 + h * 64 + R) * p.ldvt[sg] + kt * 64 + c * 8) * 2);
                dma16(make_rsrc(p.vt[sg], p.vtbytes[sg]), dst + i * 1024, kmin < nk ? off : OOB_SENTINEL);
            }
        }
    };

    // ---- fragment addresses ----
    // K (A operand of S^T): row = key kb*32 + l31, chunk 2s + u
    int k_addr[2], k_swz[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int r = kb * 32 + l31; k_addr[kb] = r * 128; k_swz[kb] = (r >> 1) & 7; }
    // V^T (A operand of O^T): row = d db*32 + l31; PV step ks, half-wave u: chunk 2ks + u (V^T is stored in key order, so the
    // 8 keys 16ks + {0..3, 8..11} + 4u this half contracts are one 16-byte read: same conflict-free pattern as K)
    int v_addr[2], v_swz[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) { const int r = db * 32 + l31; v_addr[db] = 8192 + r * 128; v_swz[db] = (r >> 1) & 7; }

    // softmax scale (d^-0.5) folded with log2(e); 1 when the caller already multiplied Q by it (args flag q_prescaled)
    const float cs = p.q_prescaled ? 1.0f : 0.125f * 1.44269504088896341f;
    f32x16 oacc[2], ofin[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[db][r] = 0.f; ofin[db][r] = 0.f; }
    float m_run = NEG_BIG, l_run = 0.f;
    if (MODE == IDMVTON_ATTN_SELF) {
        // closed form for absent (all-zero) segments: nk keys with logit 0, value 0
        int nz = 0;
        if (p.nseg > 0 && !pres0) nz += p.nk[0];
        if (p.nseg > 1 && !pres1) nz += p.nk[1];
        if (nz > 0) { m_run = 0.f; l_run = u == 0 ? (float)nz : 0.f; }
    }

    const int ns = (nt + KT - 1) / KT;                    // stages
    auto issue_stage = [&](int s, int bufi) {
#pragma unroll
        for (int sub = 0; sub < KT; ++sub)
            if (s * KT + sub < nt) issue(s * KT + sub, smem + bufi * STAGE + sub * 16384);
    };
    if constexpr (ST == 2) { if (ns > 0) issue_stage(0, 0); }
    else {
#pragma unroll
        for (int s = 0; s < ST - 1; ++s)
            if (s < ns) issue_stage(s, s);
        // Q (the oldest DMAs, issued by this wave for itself) has landed once at most the prologue tiles are outstanding
        if (ns >= ST - 1) wait_vmcnt<(ST - 1) * IPW>(); else wait_vmcnt<0>();
        const char* dq = smem + ST * STAGE + wave * 4096 + l31 * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const v8*)(dq + (((2 * s + u) ^ ((l31 >> 1) & 7)) << 4));
    }
    int cbuf = 0, ibuf = ST - 1;
    for (int st_i = 0; st_i < ns; ++st_i) {
        if constexpr (ST == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st_i + 1 < ns) issue_stage(st_i + 1, (st_i + 1) & 1);
        } else {
            if (st_i + ST - 2 < ns) wait_vmcnt<(ST - 2) * IPW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st_i + ST - 1 < ns) issue_stage(st_i + ST - 1, ibuf);
        }
        const char* sbuf = smem + cbuf * STAGE;
        cbuf = cbuf + 1 == ST ? 0 : cbuf + 1;
        ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
#pragma unroll
      for (int sub = 0; sub < KT; ++sub) {
        const int t = st_i * KT + sub;
        if (t >= nt) break;
        const char* buf = sbuf + sub * 16384;
        const int sg = t < nt0 ? 0 : 1;
        const int kt = sg ? t - nt0 : t;
        const int valid = p.nk[sg] - kt * 64;            // keys of this tile that exist (>= 64: all)

        // ---- S^T = K . Q^T ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v8 kf = *(const v8*)(buf + k_addr[kb] + (((2 * s + u) ^ k_swz[kb]) << 4));
                sacc[kb] = VT<T>::mfma(kf, qf[s], sacc[kb]);
            }
        }
        // ---- online softmax (this lane: one q row, 32 of the tile's 64 keys) ----
        if (valid < 64) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= valid) sacc[kb][r] = NEG_BIG;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run, mx * cs);
        const bool grew = m_new > m_run;
        float psum = 0.f;
        v8 pf[4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], cs, -m_new));
                psum += pv;
                pf[kb * 2 + (r >> 3)][r & 7] = (T)pv;
            }
        if (ST == 2 || __any(grew)) {                    // no lane's max moved: alpha == 1 exactly, skip the O rescale
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        m_run = m_new;
        l_run += psum;
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const v8 vf = *(const v8*)(buf + v_addr[db] + (((2 * ks + u) ^ v_swz[db]) << 4));
                oacc[db] = VT<T>::mfma(vf, pf[ks], oacc[db]);
            }
        if (MODE == IDMVTON_ATTN_CROSS) {
            if (t == nt0 - 1) {                          // end of the text group: finalise it and restart the softmax
                const float lt = xhalf_sum(l_run);
                const float inv = 1.0f / lt;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { ofin[db][r] = oacc[db][r] * inv; oacc[db][r] = 0.f; }
                m_run = NEG_BIG; l_run = 0.f;
            }
        }
      }
    }

    // ---- finalise and store: lane holds O[q][h*64 + db*32 + 8g + 4u + j] ----
    const float lt = xhalf_sum(l_run);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    const float sc = MODE == IDMVTON_ATTN_CROSS ? p.ip_scale * inv : inv;
    if (q_row < p.Nq) {                                  // both lanes of a pair (l, l+32) hold the same query row
        T* op = (T*)p.out + ((size_t)b * p.Nq + q_row) * p.ldo + h * 64 + 8 * u;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                v4 o[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[k][j] = (T)(ofin[db][8 * gp + 4 * k + j] + oacc[db][8 * gp + 4 * k + j] * sc);
                store_cols8(op + db * 32 + 16 * gp, o[0], o[1]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_pp_kernel: SELF mode, 8 waves x 32 query rows, the two waves of every SIMD in PING-PONG.
//
// With head_dim 64 a 64-key tile costs a wave 16 MFMAs (512 matrix-pipe cycles) and ~170 VALU/transcendental instructions of
// online softmax -- about the same time.  A wave cannot overlap the two by itself (the softmax consumes the QK^T accumulator
// and produces the PV operand), and two waves that run the same code in lock-step after one barrier per tile (attn_kernel)
// collide on the matrix pipe and then on the VALU.  Here the per-tile work is cut into two barrier-delimited blocks,
//     MFMA block j : O^T += V^T(j-1) . P^T(j-1)   and   S^T(j) = K(j) . Q^T        (16 MFMA on 4 independent accumulators)
//     VALU block j : online softmax of S^T(j) -> P^T(j)                            (no LDS, no matrix pipe)
// and the two wave groups (the waves w and w+4 share a SIMD) run them one block apart: while group 0 is in MFMA block j,
// group 1 is in VALU block j-1, then they swap.  Every phase starts with one workgroup s_barrier, both groups execute the
// same number of barriers (2 per tile + 2).
// LDS stage s = { K tile s | V^T tile s-1 } (16 KiB): exactly what MFMA block s reads (group 0 in phase 2s, group 1 in phase
// 2s+1), filled by LDS-DMA ST-1 stages ahead by all 8 waves (2 instructions each), counted vmcnt across the raw barriers.
// PMC (profiles/r02_pmc_attn_v1_sq.txt): 155 VALU instructions per 16 MFMAs, SIMD issue 83 % busy, matrix pipe 41 %; the softmax was
// stripped on that reading (below).  Round 3 showed the block is NOT issue bound: three exact forms with a quarter fewer VALU instructions
// (packed fma; no row max per tile) ran +-0 ... +10 % (profiles/r03_attn_softmax_forms_quick_attn.log, r03_attn_lazy_max_experiment.patch) --
// what bounds a tile is its dependent chain between the two barriers (MFMA -> exp -> sum -> convert -> MFMA), so extra branches cost
// and fewer instructions do not pay.  What the softmax does:
//   * Q arrives (or is made) pre-multiplied by softmax_scale * log2(e): S is already in log2 units;
//   * the running max is SUBTRACTED BY THE MFMA: the first QK^T MFMA of a tile takes C = -m (16 registers holding this
//     lane's -m_run, rewritten only when the max moves), so S' = S - m_run comes out of the matrix pipe and P = exp2(S') is
//     ONE instruction per element (was v_fma + v_exp);
//   * the running max is only moved (and O, l rescaled) when some row's S' exceeds `thr` (log2 units); rows keep exponentiating
//     against the older max otherwise (P <= 2^thr; exact in the final normalisation because l carries the same scale).  The
//     decision for tile j is taken in VALU block j, after PV(j-1) has completed and before P(j) is exponentiated; on a move
//     O, l AND the pending S'(j) are brought to the new max (32 extra subtractions, a few tiles per walk).
// DEEP = 1: one workgroup per CU (up to 256 VGPRs): every MFMA block reads its 16 fragments from LDS up front (one exposed LDS
// latency per block); DEEP = 0: 128 VGPRs, two workgroups per CU, fragments read two MFMAs ahead (the other workgroup's waves
// fill the gaps).
// ABL (measurement only, wrong results): 1 = VALU block reduced to the P conversion (no max / exp / sum), 2 = MFMA block
// reduced to its LDS reads (no MFMA), 3 = both (the DMA / barrier / LDS-read skeleton) -- the template-ablation method of the CDNA guide for finding what bounds a phase.
template <typename T, int ST, bool DEEP, int ABL = 0>
__global__ __launch_bounds__(512, DEEP ? 2 : 4) void attn_pp_kernel(const AttnParams p) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int NW = 8, QB = 256, IPW = 2;
    __shared__ __attribute__((aligned(1024))) char smem[ST * 16384 + NW * 4096];   // [stage][K | V^T] + per-wave Q tile

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int u = lane >> 5, l31 = lane & 31;
    const int grp = (p.pp_flags & 1) ? (wave & 1) : (wave >> 2);      // wave-uniform
    const float thr = p.pp_thr;

    int b, h, qb;
    if (!attn_work(p, blockIdx.x, b, h, qb)) return;     // block-uniform
    const int q_row = qb * QB + wave * 32 + l31;

    {   // Q tile of this wave: 32 rows x 128 B -> LDS (4 DMA instructions, K's swizzle)
        const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(p.q, p.qbytes);
        char* dq = smem + ST * 16384 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((R >> 1) & 7);
            int row = qb * QB + wave * 32 + R;
            row = row < p.Nq ? row : p.Nq - 1;
            dma16(rs_q, dq + i * 1024, (uint32_t)((((size_t)b * p.Nq + row) * p.ldq + h * 64 + c * 8) * 2));
        }
    }
    const bool pres0 = p.nseg > 0 && b >= p.seg_b0[0];
    const bool pres1 = p.nseg > 1 && b >= p.seg_b0[1];
    const int nt0 = pres0 ? (p.nk[0] + 63) >> 6 : 0;
    const int nt1 = pres1 ? (p.nk[1] + 63) >> 6 : 0;
    const int nt = nt0 + nt1;

    // ---- loader: stage s = K(s) (DMA instructions j = 0..7: waves 0-3) | V^T(s-1) (j = 8..15: waves 4-7).  A wave loads
    // either K rows or V^T rows for the whole kernel, so everything that depends on that choice is folded into per-lane
    // constants here and the in-loop issue is branch-free: offset = rowbase[seg][i] + kt * tstep[seg]; a chunk is fetched iff
    // its smallest key index lim[i] + 64 kt exists (< nk[seg]); everything else (tiles -1 and nt, key tails) reads zeros.
    const int lrow = lane >> 3, lslot = lane & 7;
    const int nk0 = p.nk[0], nk1 = p.nk[1];             // in SGPRs: indexing p.nk[] by a loop value makes hipcc re-load it from the
    const bool is_k = wave < 4;                          // kernarg segment (s_load + lgkmcnt(0)) in every phase
    const int t_shift = is_k ? 0 : 1;                    // this wave's tile of stage s is s - t_shift
    uint32_t rowbase[2][IPW], tstep[2];
    int lim0 = 0;                                        // lim of instruction 0; instruction 1's is lim0 ^ lim_x (one register less)
    const int lim_x = is_k ? 8 : 32;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const size_t bsg = (size_t)(b - p.seg_b0[sg] > 0 ? b - p.seg_b0[sg] : 0);
        tstep[sg] = is_k ? (uint32_t)(64 * p.ldk[sg] * 2) : 128u;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int R = ((wave * IPW + i) & 7) * 8 + lrow;
            const int c = lslot ^ ((R >> 1) & 7);
            rowbase[sg][i] = is_k ? (uint32_t)(((bsg * p.krows[sg] + R) * p.ldk[sg] + h * 64 + c * 8) * 2)
                                  : (uint32_t)(((bsg * p.heads * 64 + h * 64 + R) * p.ldvt[sg] + c * 8) * 2);
            // V^T chunk c holds tile positions [8c, 8c+8) = keys 16(c>>1) + 4(c&1) + {0..3, 8..11} (key order)
            if (i == 0) lim0 = is_k ? R : 16 * (c >> 1) + 4 * (c & 1);     // (R1 = R0 + 8 = R0 ^ 8; c1 = c0 ^ 4 -> +-32)
        }
    }
    const __amdgpu_buffer_rsrc_t rs0 = is_k ? make_rsrc(p.k[0], p.kbytes[0]) : make_rsrc(p.vt[0], p.vtbytes[0]);
    const __amdgpu_buffer_rsrc_t rs1 = is_k ? make_rsrc(p.k[1], p.kbytes[1]) : make_rsrc(p.vt[1], p.vtbytes[1]);
    auto issue_stage = [&](int s, int bufi) {
        char* dst = smem + bufi * 16384 + wave * (IPW * 1024);
        const int t = s - t_shift;
        const bool in_range = t >= 0 && t < nt;
        const bool sg1 = t >= nt0;
        const int kt = sg1 ? t - nt0 : t;
        const int room = in_range ? (sg1 ? nk1 : nk0) - kt * 64 : 0;               // keys of this tile that exist
        const uint32_t toff = (uint32_t)kt * (sg1 ? tstep[1] : tstep[0]);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const uint32_t off = (sg1 ? rowbase[1][i] : rowbase[0][i]) + toff;
            const int lim = i == 0 ? lim0 : (lim0 ^ lim_x);
            if (sg1) dma16(rs1, dst + i * 1024, lim < room ? off : OOB_SENTINEL);
            else dma16(rs0, dst + i * 1024, lim < room ? off : OOB_SENTINEL);
        }
    };

    // ---- fragment addresses (row * 128 and the row's swizzle key); K rows = keys, V^T rows = d ----
    int f_addr[2], f_swz[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int r = kb * 32 + l31; f_addr[kb] = r * 128; f_swz[kb] = (r >> 1) & 7; }

    // NEGM: -m_run lives in 16 registers and is subtracted by the MFMA (C operand of the first QK^T MFMA).  Only the 256-register
    // build has room for it; the 128-register build subtracts in the softmax (v_fma + v_exp per element instead of v_exp).
    constexpr bool NEGM = DEEP;
    // LSUM: the softmax denominator is accumulated BY THE MATRIX PIPE: one more MFMA per PV k-step with an all-ones A operand gives, in every
    // row of a 32x32 accumulator, the column sums of P^T = this lane's row sum of P over the tile's 64 keys (both half-waves' keys: no
    // cross-half exchange at the end).  The ablation builds show the kernel bound by its VALU block with the MFMAs entirely hidden
    // (profiles/r05_pmc_attention_variants.txt: the kernel without any MFMA takes the same time), so 4 more MFMAs per tile for 32 fewer
    // VALU adds per lane and tile is the right trade -- in the 256-register build only: the accumulator costs 16 registers the 128-register
    // build does not have.  l then sums the ROUNDED probabilities, the same values the numerator's P.V product uses.
    constexpr bool LSUM = DEEP;
    f32x16 oacc[2], sacc[2], negm, lacc;
    v8 pf[4], ones;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[db][r] = 0.f; sacc[db][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[r] = 0.f; lacc[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[ks][j] = (T)0.f;
    // m_run starts at 0 (S' = S).  Without a closed-form segment the first tile FORCES the max to its own row max (whatever its
    // sign); with one (nk keys of logit 0, value 0: m = 0, l = nk) the ordinary threshold rule applies from the first tile on.
    float m_run = 0.f, l_run = 0.f;
    float thr_cur = -3.0e38f, floor_cur = -3.0e38f;      // first tile: the branch below is always taken and delta = the row max
    {
        int nz = 0;
        if (p.nseg > 0 && !pres0) nz += p.nk[0];
        if (p.nseg > 1 && !pres1) nz += p.nk[1];
        if (nz > 0) {
            l_run = u == 0 ? (float)nz : 0.f; thr_cur = thr; floor_cur = 0.f;
            if constexpr (LSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lacc[r] = (float)nz;
            }
        }
    }

#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s <= nt) issue_stage(s, s);
    if (nt >= ST - 2) wait_vmcnt<(ST - 1) * IPW>(); else wait_vmcnt<0>();       // Q (oldest DMAs, own rows) has landed
    // Q fragments (B operand of S^T), kept in registers.  cs: softmax scale folded with log2(e), 1 when q arrives pre-multiplied
    // (the NEGM build is only launched with a pre-multiplied q: scaling q here would round it a second time).
    const char* const dq = smem + ST * 16384 + wave * 4096 + l31 * 128;
    v8 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const v8*)(dq + (((2 * s4 + u) ^ ((l31 >> 1) & 7)) << 4));
    const float cs = p.q_prescaled ? 1.0f : 0.125f * 1.44269504088896341f;

    auto k_frag = [&](const char* buf, int kb, int s4) { return *(const v8*)(buf + f_addr[kb] + (((2 * s4 + u) ^ f_swz[kb]) << 4)); };
    auto v_frag = [&](const char* buf, int db, int ks) { return *(const v8*)(buf + 8192 + f_addr[db] + (((2 * ks + u) ^ f_swz[db]) << 4)); };

    // MFMA block s: PV(s-1) and QK^T(s), interleaved so that consecutive MFMAs never share an accumulator.  The edge blocks
    // run the same 16 MFMAs on harmless operands (block 0: P = 0 and the zero-filled V^T half; block nt: the zero-filled K
    // half, result unused) -- two half-blocks per workgroup instead of three code paths in the loop.
    auto mfma_block = [&](const char* buf) {
        v8 vf[2][4], kf[2][4];
        if constexpr (DEEP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { vf[0][i] = v_frag(buf, 0, i); vf[1][i] = v_frag(buf, 1, i); kf[0][i] = k_frag(buf, 0, i); kf[1][i] = k_frag(buf, 1, i); }
            __builtin_amdgcn_sched_barrier(0);           // all 16 ds_read_b128 issued before the first MFMA; the waits stay counted
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (!DEEP) { vf[0][i] = v_frag(buf, 0, i); vf[1][i] = v_frag(buf, 1, i); kf[0][i] = k_frag(buf, 0, i); kf[1][i] = k_frag(buf, 1, i); }
            if constexpr (ABL == 2 || ABL == 3) {        // keep the reads and the operands alive, issue no MFMA
                asm volatile("" :: "v"(vf[0][i]), "v"(vf[1][i]), "v"(kf[0][i]), "v"(kf[1][i]), "v"(pf[i]), "v"(qf[i]));
                continue;
            }
            oacc[0] = VT<T>::mfma(vf[0][i], pf[i], oacc[0]);
            oacc[1] = VT<T>::mfma(vf[1][i], pf[i], oacc[1]);
            if constexpr (LSUM) lacc = VT<T>::mfma(ones, pf[i], lacc);
            if (i == 0 && NEGM) {                        // S' = K.Q^T - m_run: the max subtraction rides on the accumulator input
                sacc[0] = VT<T>::mfma(kf[0][0], qf[0], negm);
                sacc[1] = VT<T>::mfma(kf[1][0], qf[0], negm);
            } else if (i == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                sacc[0] = VT<T>::mfma(kf[0][0], qf[0], z);
                sacc[1] = VT<T>::mfma(kf[1][0], qf[0], z);
            } else {
                sacc[0] = VT<T>::mfma(kf[0][i], qf[i], sacc[0]);
                sacc[1] = VT<T>::mfma(kf[1][i], qf[i], sacc[1]);
            }
        }
    };

    // VALU block j: online softmax of S^T(j) (this lane: one q row, 32 of the tile's 64 keys) -> P^T(j) as PV B-operand
    auto valu_block = [&](int j) {
        if constexpr (ABL == 1 || ABL == 3) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) pf[kb * 2 + (r >> 3)][r & 7] = (T)sacc[kb][r];
            l_run += 1.f;
            return;
        }
        const bool sg1 = j >= nt0;
        const int valid = sg1 ? nk1 - (j - nt0) * 64 : nk0 - j * 64;     // keys of this tile that exist (>= 64: all)
        if (valid < 64) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= valid) sacc[kb][r] = NEG_BIG;
                }
        }
        float mx = fmaxf(sacc[0][0], sacc[1][0]);       // (built with -fno-honor-nans: fmaxf chains fold to v_max3_f32 without
#pragma unroll                                           //  a canonicalising v_max per MFMA output)
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
        mx = xhalf_max(mx);                              // this row's max over the tile's 64 keys (NEGM: of S' = S - m_run, else of S)
        if constexpr (!NEGM) mx = fmaf(mx, cs, -m_run);
        if (__any(mx > thr_cur)) {                       // wave-uniform; taken on the first tile and when a max moved by > thr
            const float delta = fmaxf(mx, floor_cur);    // later tiles: the max never moves down
            const float alpha = __builtin_amdgcn_exp2f(fminf(-delta, 0.f));   // (first tile: O = l = 0, keep alpha finite)
            l_run *= alpha;
            if constexpr (LSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            m_run += delta;
            thr_cur = thr; floor_cur = 0.f;
            if constexpr (NEGM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { sacc[0][r] -= delta; sacc[1][r] -= delta; negm[r] = -m_run; }
            }
        }
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(NEGM ? sacc[kb][r] : fmaf(sacc[kb][r], cs, -m_run));
                if constexpr (!LSUM) { if (r & 1) ps1 += pv; else ps0 += pv; }
                pf[kb * 2 + (r >> 3)][r & 7] = (T)pv;
            }
        if constexpr (!LSUM) l_run += ps0 + ps1;
    };

    // Phase 2s starts with `sync_stage`: wait until this wave's share of stage s has landed (stages s .. s+ST-2 are outstanding,
    // fewer at the tail), workgroup barrier (=> every share landed, and every wave is done with stage s-1), refill the freed
    // buffer with stage s+ST-1.  The two groups run separate loops (same barrier count) so that neither carries the other's
    // register assignment across the blocks.
    int cbuf = 0, ibuf = ST - 1;
    auto sync_stage = [&](int s) -> const char* {
        if (s + ST - 2 <= nt) wait_vmcnt<(ST - 2) * IPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (s + ST - 1 <= nt) issue_stage(s + ST - 1, ibuf);
        const char* buf = smem + cbuf * 16384;
        cbuf = cbuf + 1 == ST ? 0 : cbuf + 1;
        ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
        return buf;
    };
    auto mid_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    if (grp == 0) {
        for (int s = 0; s <= nt; ++s) {
            const char* buf = sync_stage(s);             // phase 2s
            mfma_block(buf);
            mid_barrier();                               // phase 2s+1
            if (s < nt) valu_block(s);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int s = 0; s <= nt; ++s) {
            const char* buf = sync_stage(s);             // phase 2s
            if (s >= 1) valu_block(s - 1);
            mid_barrier();                               // phase 2s+1
            mfma_block(buf);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- finalise and store: lane holds O[q][h*64 + db*32 + 8g + 4u + j] ----
    const float lt = LSUM ? lacc[0] : xhalf_sum(l_run);     // (LSUM: every row of the accumulator holds this lane's query-row sum over all keys)
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    if (q_row < p.Nq) {
        T* op = (T*)p.out + ((size_t)b * p.Nq + q_row) * p.ldo + h * 64 + 8 * u;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                v4 o[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[k][j] = (T)(oacc[db][8 * gp + 4 * k + j] * inv);
                store_cols8(op + db * 32 + 16 * gp, o[0], o[1]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_pf_kernel: the ping-pong kernel with every MFMA block reduced to its MFMAs.
//
// attn_pp_kernel<DEEP> opens each MFMA block with the block's 16 ds_read_b128 and their latency, with the matrix pipe of that SIMD idle
// (the partner wave is in its VALU block) -- once per phase, two phases per tile.  Here a group reads the fragments of its NEXT MFMA block
// during its own VALU block, one phase early, into 64 registers that stay live across the barrier: an MFMA block is then 20 (LSUM) / 16
// back-to-back MFMAs on operands that are already in registers, and the LDS reads, the LDS-DMA issue and the -m_run splat that seeds the
// QK^T accumulators (the 8 v_mov_b64 hipcc put between the MFMAs of the older kernel) all sit beside the softmax in the VALU block.
// That needs the tile in LDS one phase earlier: 3-stage ring, stage s = { K(s) | V^T(s-1) } in buffer s % 3,
//     phase A_s : group 0  MFMA block s                       | group 1  reads F1(s) (stage s), softmax(s-1)
//     phase B_s : group 0  reads F0(s+1) (stage s+1), softmax(s) | group 1  MFMA block s
// Barrier A_s only aligns the phases; barrier M_s (between A_s and B_s) is where every wave has waited for its share of stage s+1
// (counted vmcnt: stage s+2 stays in flight) and after which buffer s % 3 (last read in A_s) is refilled with stage s+3.
// Always with q pre-multiplied by softmax_scale * log2(e) (the running max is subtracted by seeding the accumulator with -m_run).
#ifdef ATTN_DBG
// Anatomy build (tools only, -DATTN_DBG): every wave sums, over its tile walk, the cycles it spends in its phase-A work, waiting at barrier M,
// in its phase-B work and waiting at barrier A (s_memtime after each barrier release and before each arrival; ~50 cycles each on the path).
__device__ unsigned long long g_attn_dbg[4096 * 8 * 8];
extern "C" int idmvton_attn_dbg_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_dbg), bytes); }
#define DBG_STAMP(slot) do { const unsigned long long t_ = __builtin_readcyclecounter(); dbg_acc[slot] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define DBG_STAMP(slot) do {} while (0)
#endif
template <typename T, bool LSUM, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn_pf_kernel(const AttnParams p) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int NW = 8, QB = 256, IPW = 2, ST = 3;
    __shared__ __attribute__((aligned(1024))) char smem[ST * 16384 + NW * 4096];   // [stage][K | V^T] + per-wave Q tile

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int u = lane >> 5, l31 = lane & 31;
    // ABL (anatomy builds, wrong results): 1 no in-loop DMA, 2 no fragment reads, 3 group roles swapped (the older waves 0-3 take group 1's
    // schedule), 4 static s_setprio 1 for waves 4-7, 5 = 1 + 2
    const int grp = ABL == 3 ? 1 - (wave >> 2) : wave >> 2;   // waves w and w + 4 share a SIMD
    const float thr = p.pp_thr;
    if constexpr (ABL == 4) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
#ifdef ATTN_DBG
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = 0;
#endif

    int b, h, qb;
    if (!attn_work(p, blockIdx.x, b, h, qb)) return;     // block-uniform
    const int q_row = qb * QB + wave * 32 + l31;

    {   // Q tile of this wave: 32 rows x 128 B -> LDS (4 DMA instructions, K's swizzle)
        const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(p.q, p.qbytes);
        char* dq = smem + ST * 16384 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((R >> 1) & 7);
            int row = qb * QB + wave * 32 + R;
            row = row < p.Nq ? row : p.Nq - 1;
            dma16(rs_q, dq + i * 1024, (uint32_t)((((size_t)b * p.Nq + row) * p.ldq + h * 64 + c * 8) * 2));
        }
    }
    const bool pres0 = p.nseg > 0 && b >= p.seg_b0[0];
    const bool pres1 = p.nseg > 1 && b >= p.seg_b0[1];
    const int nt0 = pres0 ? (p.nk[0] + 63) >> 6 : 0;
    const int nt1 = pres1 ? (p.nk[1] + 63) >> 6 : 0;
    const int nt = nt0 + nt1;

    // ---- loader (as attn_pp_kernel): waves 0-3 fetch K(s), waves 4-7 V^T(s-1); per-lane constants, branch-free issue ----
    const int lrow = lane >> 3, lslot = lane & 7;
    const int nk0 = p.nk[0], nk1 = p.nk[1];
    const bool is_k = wave < 4;
    const int t_shift = is_k ? 0 : 1;
    uint32_t rowbase[2][IPW], tstep[2];
    int lim0 = 0;
    const int lim_x = is_k ? 8 : 32;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const size_t bsg = (size_t)(b - p.seg_b0[sg] > 0 ? b - p.seg_b0[sg] : 0);
        tstep[sg] = is_k ? (uint32_t)(64 * p.ldk[sg] * 2) : 128u;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int R = ((wave * IPW + i) & 7) * 8 + lrow;
            const int c = lslot ^ ((R >> 1) & 7);
            rowbase[sg][i] = is_k ? (uint32_t)(((bsg * p.krows[sg] + R) * p.ldk[sg] + h * 64 + c * 8) * 2)
                                  : (uint32_t)(((bsg * p.heads * 64 + h * 64 + R) * p.ldvt[sg] + c * 8) * 2);
            if (i == 0) lim0 = is_k ? R : 16 * (c >> 1) + 4 * (c & 1);
        }
    }
    const __amdgpu_buffer_rsrc_t rs0 = is_k ? make_rsrc(p.k[0], p.kbytes[0]) : make_rsrc(p.vt[0], p.vtbytes[0]);
    const __amdgpu_buffer_rsrc_t rs1 = is_k ? make_rsrc(p.k[1], p.kbytes[1]) : make_rsrc(p.vt[1], p.vtbytes[1]);
    auto issue_stage = [&](int s, int bufi) {
        char* dst = smem + bufi * 16384 + wave * (IPW * 1024);
        const int t = s - t_shift;
        const bool in_range = t >= 0 && t < nt;
        const bool sg1 = t >= nt0;
        const int kt = sg1 ? t - nt0 : t;
        const int room = in_range ? (sg1 ? nk1 : nk0) - kt * 64 : 0;
        const uint32_t toff = (uint32_t)kt * (sg1 ? tstep[1] : tstep[0]);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const uint32_t off = (sg1 ? rowbase[1][i] : rowbase[0][i]) + toff;
            const int lim = i == 0 ? lim0 : (lim0 ^ lim_x);
            if (sg1) dma16(rs1, dst + i * 1024, lim < room ? off : OOB_SENTINEL);
            else dma16(rs0, dst + i * 1024, lim < room ? off : OOB_SENTINEL);
        }
    };

    int f_addr[2], f_swz[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int r = kb * 32 + l31; f_addr[kb] = r * 128; f_swz[kb] = (r >> 1) & 7; }

    f32x16 oacc[2], sacc[2], lacc;
    v8 pf[4], ones, vf[2][4], kf[2][4];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[db][r] = 0.f; sacc[db][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (T)1.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[ks][j] = (T)0.f;
    float m_run = 0.f, l_run = 0.f;
    float thr_cur = -3.0e38f, floor_cur = -3.0e38f;      // first tile: the max is forced to the tile's own row max (see attn_pp_kernel)
    {
        int nz = 0;
        if (p.nseg > 0 && !pres0) nz += p.nk[0];
        if (p.nseg > 1 && !pres1) nz += p.nk[1];
        if (nz > 0) {
            l_run = u == 0 ? (float)nz : 0.f; thr_cur = thr; floor_cur = 0.f;
            if constexpr (LSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lacc[r] = (float)nz;
            }
        }
    }

    // ---- prologue: stages 0..2 in flight, Q and stage 0 landed ----
    const int npro = nt + 1 < ST ? nt + 1 : ST;          // stages 0..nt exist
#pragma unroll
    for (int s = 0; s < ST; ++s)
        if (s < npro) issue_stage(s, s);
    if (npro == 3) wait_vmcnt<2 * IPW>(); else if (npro == 2) wait_vmcnt<IPW>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    const char* const dq = smem + ST * 16384 + wave * 4096 + l31 * 128;
    v8 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const v8*)(dq + (((2 * s4 + u) ^ ((l31 >> 1) & 7)) << 4));

    auto read_frags = [&](const char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                vf[x][i] = *(const v8*)(buf + 8192 + f_addr[x] + (((2 * i + u) ^ f_swz[x]) << 4));
                kf[x][i] = *(const v8*)(buf + f_addr[x] + (((2 * i + u) ^ f_swz[x]) << 4));
            }
        }
    };
    // MFMA block s: PV(s-1) and QK^T(s) on five (four) independent accumulators; S' = K.Q^T - m_run (sacc was seeded with -m_run)
    auto mfma_block = [&]() {
        if constexpr (ABL == 6) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oacc[0] = VT<T>::mfma(vf[0][i], pf[i], oacc[0]);
            oacc[1] = VT<T>::mfma(vf[1][i], pf[i], oacc[1]);
            if constexpr (LSUM) lacc = VT<T>::mfma(ones, pf[i], lacc);
            sacc[0] = VT<T>::mfma(kf[0][i], qf[i], sacc[0]);
            sacc[1] = VT<T>::mfma(kf[1][i], qf[i], sacc[1]);
        }
    };
    // VALU block j: online softmax of S'(j) -> P(j) (PV B operand); re-seeds sacc with -m_run for QK^T(j+1)
    auto valu_block = [&](int j) {
        const bool sg1 = j >= nt0;
        const int valid = sg1 ? nk1 - (j - nt0) * 64 : nk0 - j * 64;
        if (valid < 64) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= valid) sacc[kb][r] = NEG_BIG;
                }
        }
        float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[1][r]);
        mx = xhalf_max(mx);
        if (__any(mx > thr_cur)) {
            const float delta = fmaxf(mx, floor_cur);
            const float alpha = __builtin_amdgcn_exp2f(fminf(-delta, 0.f));
            l_run *= alpha;
            if constexpr (LSUM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            m_run += delta;
            thr_cur = thr; floor_cur = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[0][r] -= delta; sacc[1][r] -= delta; }
        }
        DBG_STAMP(4);
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sacc[kb][r]);
                if constexpr (!LSUM) { if (r & 1) ps1 += pv; else ps0 += pv; }
                pf[kb * 2 + (r >> 3)][r & 7] = (T)pv;
            }
        if constexpr (!LSUM) l_run += ps0 + ps1;
        __builtin_amdgcn_sched_barrier(0);
        DBG_STAMP(5);
        const float nm = -m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[0][r] = nm; sacc[1][r] = nm; }
        __builtin_amdgcn_sched_barrier(0);
        DBG_STAMP(6);
    };
    auto phase_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    // barrier M_s: own share of stage s+1 landed (stage s+2, if it exists, stays in flight), LDS reads of this phase done
    auto mid_barrier = [&](int s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 <= nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    int b0i = 0, b1i = 1;                                // buffers of stage s and stage s + 1
    auto rotate = [&]() { b0i = b1i; b1i = b1i + 1 == ST ? 0 : b1i + 1; };

#ifdef ATTN_DBG
    dbg_t = __builtin_readcyclecounter();
#endif
    if (grp == 0) {
        read_frags(smem);                                // F(0)
        for (int s = 0; s <= nt; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            DBG_STAMP(2);
            phase_barrier();                             // A_s
            DBG_STAMP(3);
            mfma_block();
            __builtin_amdgcn_sched_barrier(0);
            DBG_STAMP(0);
            mid_barrier(s);                              // B_s
            DBG_STAMP(1);
            if ((ABL != 2 && ABL < 5) && s + 1 <= nt) read_frags(smem + b1i * 16384);
            __builtin_amdgcn_sched_barrier(0);
            if ((ABL != 1 && ABL < 5) && s + 3 <= nt) issue_stage(s + 3, b0i);
            if (s < nt) valu_block(s);
            rotate();
        }
    } else {
        for (int s = 0; s <= nt; ++s) {
            DBG_STAMP(2);
            phase_barrier();                             // A_s
            DBG_STAMP(3);
            if (ABL != 2 && ABL < 5) read_frags(smem + b0i * 16384);
            __builtin_amdgcn_sched_barrier(0);
            if (s >= 1) valu_block(s - 1);
            __builtin_amdgcn_sched_barrier(0);
            DBG_STAMP(0);
            mid_barrier(s);                              // B_s
            DBG_STAMP(1);
            mfma_block();
            __builtin_amdgcn_sched_barrier(0);
            if ((ABL != 1 && ABL < 5) && s + 3 <= nt) issue_stage(s + 3, b0i);
            rotate();
        }
    }
#ifdef ATTN_DBG
    if (lane == 0 && blockIdx.x < 4096) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_attn_dbg[(blockIdx.x * 8 + wave) * 8 + i] = dbg_acc[i];
    }
#endif

    const float lt = LSUM ? lacc[0] : xhalf_sum(l_run);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    if (q_row < p.Nq) {
        T* op = (T*)p.out + ((size_t)b * p.Nq + q_row) * p.ldo + h * 64 + 8 * u;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                v4 o[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[k][j] = (T)(oacc[db][8 * gp + 4 * k + j] * inv);
                store_cols8(op + db * 32 + 16 * gp, o[0], o[1]);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// attn_sp_kernel: SELF mode, 8 waves x 32 query rows, every wave SOFTWARE-PIPELINED over three tiles.
//
// What bounds head_dim-64 attention on this chip is the SIMD's shared VALU issue port, not the matrix pipe
// (profiles/r06_probe_mfma_valu_overlap.txt: beside MFMAs a plain VALU instruction costs ~4 cycles of the port, a v_exp_f32 ~8 -- the
// transcendentals of the two waves of a SIMD do not overlap each other or plain VALU work -- and an MFMA issue ~10; a 64-key tile of one wave
// needs 32 v_exp).  The ping-pong kernels above (attn_pp / attn_pf) put the softmax of a wave into a phase of its own (a ~650-cycle dependent
// chain: serial row max -> any() branch -> 32 exponentials -> converts -> accumulator seed) and pay two such phases per tile
// (profiles/r06_attention_anatomy.txt).  Here no wave waits for its partner and the instruction count per tile is cut to what the port must carry:
//   * in iteration i a wave issues, as ONE basic block,
//         matrix pipe :  O^T += V^T(i-2) . P^T(i-2)      and      S'^T(i) = K(i) . Q^T - m          (16 MFMAs)
//         VALU        :  P^T(i-1) = exp2(S'^T(i-1)), row sum                                         (independent of both products)
//     so its own softmax runs in the gaps of its own MFMAs and the two waves of a SIMD simply share the port;
//   * the exponentials are SPECULATIVE: S' already carries -m (accumulator seed), P is exponentiated against the old max before anything about
//     the tile is known;
//   * there is NO row-max pass: the per-lane row sum of the tile (needed for l anyway) bounds every P of the lane, so "sum <= LIMIT" proves
//     that no P exceeded LIMIT (and none overflowed: inf fails the test).  Only when a lane's sum exceeds it (wave-uniform branch AFTER the
//     block; always on the first tile, practically never later) is the true row max taken, O and l rescaled, P(i-1) re-exponentiated from the
//     intact S'(i-1) and the in-flight S'(i) shifted -- PV(i-2) is complete at that point, P(i-1) and S'(i) are the only values at the old scale
//     (the order rule of the deferred rescale).
// LDS ring of 3 stages, stage i = { K(i) | V^T(i-2) }, one workgroup barrier per tile (counted vmcnt across it); the loader walks the two key
// segments with running scalar state (one v_add per DMA instruction on full tiles).  Needs q pre-multiplied by softmax_scale * log2(e).
// ABL (dbg builds, timing only): bit0 no exponentials, bit1 no MFMA, bit2 no LDS fragment reads, bit3 no in-loop DMA, bit4 no barrier.
template <typename T, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 2) void attn_sp_kernel(const AttnParams p) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int QB = 32 * NW, IPW = 16 / NW, ST = 3;    // NW = 8: one workgroup per CU; NW = 4: two (independent) workgroups per CU
    __shared__ __attribute__((aligned(1024))) char smem[ST * 16384 + NW * 4096];   // [stage][K | V^T] + per-wave Q tile

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int u = lane >> 5, l31 = lane & 31;
    const float limit = p.pp_thr;                         // row-sum bound that keeps the old max (32 keys per lane at P <= 1 give <= 32)

    int b, h, qb;
    if (!sp_work(p, blockIdx.x, b, h, qb)) return;       // block-uniform
    const int q_row = qb * QB + wave * 32 + l31;

    {   // Q tile of this wave: 32 rows x 128 B -> LDS (4 DMA instructions, K's swizzle)
        const __amdgpu_buffer_rsrc_t rs_q = make_rsrc(p.q, p.qbytes);
        char* dq = smem + ST * 16384 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((R >> 1) & 7);
            int row = qb * QB + wave * 32 + R;
            row = row < p.Nq ? row : p.Nq - 1;
            dma16(rs_q, dq + i * 1024, (uint32_t)((((size_t)b * p.Nq + row) * p.ldq + h * 64 + c * 8) * 2));
        }
    }
    const bool pres0 = p.nseg > 0 && b >= p.seg_b0[0];
    const bool pres1 = p.nseg > 1 && b >= p.seg_b0[1];
    const int nt0 = pres0 ? (p.nk[0] + 63) >> 6 : 0;
    const int nt1 = pres1 ? (p.nk[1] + 63) >> 6 : 0;
    const int nt = nt0 + nt1;
    const int nk0 = p.nk[0], nk1 = p.nk[1];

    // ---- loader: waves 0-3 fetch K(i), waves 4-7 V^T(i-2).  The stages are issued strictly in order, so the segment walk is running scalar
    // state: ld_t = tile of the next stage for this wave (negative: V^T's two-stage lag; >= nt: past the end -> zero fill), ld_off = byte
    // offset of that tile inside its segment, ld_room = keys of the segment from that tile on, ld_seg = segment. ----
    const int lrow = lane >> 3, lslot = lane & 7;
    const bool is_k = wave < NW / 2;
    // per-lane byte offsets of this wave's IPW DMA rows inside segment sg (tile 0).  Only the CURRENT segment's set lives in registers; it is
    // recomputed at the segment switch.  lane_lim(i): the smallest key index piece i fetches (only a segment's partial last tile needs it).
    uint32_t rb[IPW];
    auto seg_base = [&](int sg) {
        const size_t bsg = (size_t)(b - p.seg_b0[sg] > 0 ? b - p.seg_b0[sg] : 0);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int R = ((wave * IPW + i) & 7) * 8 + lrow;
            const int c = lslot ^ ((R >> 1) & 7);
            rb[i] = is_k ? (uint32_t)(((bsg * p.krows[sg] + R) * p.ldk[sg] + h * 64 + c * 8) * 2)
                         : (uint32_t)(((bsg * p.heads * 64 + h * 64 + R) * p.ldvt[sg] + c * 8) * 2);
        }
    };
    auto lane_lim = [&](int i) {
        const int R = ((wave * IPW + i) & 7) * 8 + lrow;
        const int c = lslot ^ ((R >> 1) & 7);
        return is_k ? R : 16 * (c >> 1) + 4 * (c & 1);
    };
    const uint32_t tstep0 = is_k ? (uint32_t)(64 * p.ldk[0] * 2) : 128u, tstep1 = is_k ? (uint32_t)(64 * p.ldk[1] * 2) : 128u;
    const __amdgpu_buffer_rsrc_t rs0 = is_k ? make_rsrc(p.k[0], p.kbytes[0]) : make_rsrc(p.vt[0], p.vtbytes[0]);
    const __amdgpu_buffer_rsrc_t rs1 = is_k ? make_rsrc(p.k[1], p.kbytes[1]) : make_rsrc(p.vt[1], p.vtbytes[1]);
    int ld_t = is_k ? 0 : -2, ld_room = nt0 > 0 ? nk0 : nk1;
    bool ld_seg1 = nt0 == 0;
    uint32_t ld_off = 0;
    seg_base(ld_seg1 ? 1 : 0);
    auto issue_next = [&](int bufi) {
        char* dst = smem + bufi * 16384 + wave * (IPW * 1024);
        if (ld_t < 0 || ld_t >= nt) {                    // outside the walk: zero fill (all lanes out of range)
#pragma unroll
            for (int i = 0; i < IPW; ++i) dma16(rs0, dst + i * 1024, OOB_SENTINEL);
        } else {
            if (!ld_seg1 && ld_t == nt0) { ld_seg1 = true; ld_off = 0; ld_room = nk1; seg_base(1); }
            if (ld_room >= 64) {                          // full tile: no per-lane masking
#pragma unroll
                for (int i = 0; i < IPW; ++i) { if (ld_seg1) dma16(rs1, dst + i * 1024, rb[i] + ld_off); else dma16(rs0, dst + i * 1024, rb[i] + ld_off); }
            } else {
#pragma unroll
                for (int i = 0; i < IPW; ++i) {
                    const uint32_t off = lane_lim(i) < ld_room ? rb[i] + ld_off : OOB_SENTINEL;
                    if (ld_seg1) dma16(rs1, dst + i * 1024, off); else dma16(rs0, dst + i * 1024, off);
                }
            }
            ld_off += ld_seg1 ? tstep1 : tstep0;
            ld_room -= 64;
        }
        ++ld_t;
    };

    int f_addr[2], f_swz[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { const int r = kb * 32 + l31; f_addr[kb] = r * 128; f_swz[kb] = (r >> 1) & 7; }

    f32x16 oacc[2], sA[2], sB[2];
    v8 P[4];                                              // P^T of ONE tile: PV(i-2) reads it early in a block, the converts of tile i-1 rewrite it late
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[db][r] = 0.f; sA[db][r] = 0.f; sB[db][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) P[ks][j] = (T)0.f;
    // m_run starts at 0 (S' = S).  Without a closed-form segment the first tile FORCES the max to its own row max (whatever its sign: lim_cur < 0
    // sends it through the rescale branch); with one (nk keys of logit 0, value 0: m = 0, l = nk) the ordinary rule applies from the first tile on.
    float m_run = 0.f, l_run = 0.f;
    float lim_cur = -1.f, floor_cur = -3.0e38f;
    {
        int nz = 0;
        if (p.nseg > 0 && !pres0) nz += nk0;
        if (p.nseg > 1 && !pres1) nz += nk1;
        if (nz > 0) { l_run = u == 0 ? (float)nz : 0.f; lim_cur = limit; floor_cur = 0.f; }
    }

    // ---- prologue: stages 0, 1 in flight; Q and stage 0 landed ----
    const int last = nt + 1;                              // stages 0 .. nt + 1 exist
    issue_next(0);
    issue_next(1);
    wait_vmcnt<IPW>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    const char* const dq = smem + ST * 16384 + wave * 4096 + l31 * 128;
    v8 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const v8*)(dq + (((2 * s4 + u) ^ ((l31 >> 1) & 7)) << 4));

    auto k_frag = [&](const char* buf, int kb, int s4) { return *(const v8*)(buf + f_addr[kb] + (((2 * s4 + u) ^ f_swz[kb]) << 4)); };
    auto v_frag = [&](const char* buf, int db, int ks) { return *(const v8*)(buf + 8192 + f_addr[db] + (((2 * ks + u) ^ f_swz[db]) << 4)); };

    // top of iteration i: stage i landed for every wave, every wave done with stage i-1's buffer -> refill it with stage i+2
    int cbuf = 0, ibuf = 2;
    auto sync_stage = [&](int i) -> const char* {
        if (i + 1 <= last) wait_vmcnt<IPW>(); else wait_vmcnt<0>();
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (!(ABL & 8) && i + 2 <= last) issue_next(ibuf);
        const char* buf = smem + cbuf * 16384;
        cbuf = cbuf + 1 == ST ? 0 : cbuf + 1;
        ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
        return buf;
    };

    // One iteration = block(i) + tail(i).  block: exp(i-1) [Scur -> Pcur, speculative], PV(i-2) [Pprev], QK^T(i) [-> Snext, seeded with -m], one
    // basic block; tail: the row-sum test, the rare fixup, l, and the seed of Scur (the next iteration's Snext).  jvalid = keys of tile i-1.
    auto block = [&](int jvalid, const char* buf, f32x16 (&Scur)[2], f32x16 (&Snext)[2], float (&ps)[4]) {
        if (jvalid < 64) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= jvalid) Scur[kb][r] = NEG_BIG;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- one basic block: 16 MFMAs, 16 LDS reads, 32 exponentials + converts + the row sum ----
        v8 vf[2][4], kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (ABL & 4) { vf[0][ks] = qf[ks]; vf[1][ks] = qf[ks]; kf[0][ks] = qf[ks]; kf[1][ks] = qf[ks]; }
            else { vf[0][ks] = v_frag(buf, 0, ks); vf[1][ks] = v_frag(buf, 1, ks); kf[0][ks] = k_frag(buf, 0, ks); kf[1][ks] = k_frag(buf, 1, ks); }
        }
        if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                oacc[0] = VT<T>::mfma(vf[0][ks], P[ks], oacc[0]);
                oacc[1] = VT<T>::mfma(vf[1][ks], P[ks], oacc[1]);
                Snext[0] = VT<T>::mfma(kf[0][ks], qf[ks], Snext[0]);
                Snext[1] = VT<T>::mfma(kf[1][ks], qf[ks], Snext[1]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(vf[0][ks]), "v"(vf[1][ks]), "v"(kf[0][ks]), "v"(kf[1][ks]), "v"(P[ks]));
        }
        ps[0] = ps[1] = ps[2] = ps[3] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = (ABL & 1) ? Scur[kb][r] : __builtin_amdgcn_exp2f(Scur[kb][r]);
                ps[r & 3] += pv;
                P[kb * 2 + (r >> 3)][r & 7] = (T)pv;     // (after the PV MFMA that read the old P[kb * 2 + (r >> 3)]: same registers)
            }
        // the speculative P must exist at the end of this block (hipcc otherwise sinks the exponentials behind the branch of the tail, into both arms)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(P[ks]));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tail = [&](f32x16 (&Scur)[2], f32x16 (&Snext)[2], float (&ps)[4]) {
        float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);   // this lane's 32 keys; every P of the lane is <= psum
        if (__any(!(psum <= lim_cur))) {                 // rare (always on the first tile): some P may be large -- take the true row max
            float mx = fmaxf(Scur[0][0], Scur[1][0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, Scur[0][r]), Scur[1][r]);
            mx = xhalf_max(mx);
            const float delta = fmaxf(mx, floor_cur);    // later tiles: the max never moves down
            const float alpha = __builtin_amdgcn_exp2f(fminf(-delta, 0.f));   // (first tile: O = l = 0, keep alpha finite)
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
            m_run += delta;
            lim_cur = limit; floor_cur = 0.f;
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(Scur[kb][r] - delta);
                    if (r & 1) q1 += pv; else q0 += pv;
                    P[kb * 2 + (r >> 3)][r & 7] = (T)pv;
                    Snext[kb][r] -= delta;
                }
            psum = q0 + q1;
        }
        l_run += psum;
        const float nm = -m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Scur[0][r] = nm; Scur[1][r] = nm; }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- iteration 0: QK^T(0) only ----
    {
        const char* buf = sync_stage(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sA[0] = VT<T>::mfma(k_frag(buf, 0, ks), qf[ks], sA[0]);
            sA[1] = VT<T>::mfma(k_frag(buf, 1, ks), qf[ks], sA[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- iterations 1 .. nt, two per trip (the S register sets trade places); jv = keys of tile i-1 that exist (running scalar).
    // (Measured and dropped: rotating the schedule of waves 4-7 by the tail -- block(i) sync(i+1) tail(i) -- so that one wave's barrier / test /
    // seed section lies beside its SIMD partner's MFMA block: +5 % time, profiles/r06_attention_sp_ablation.txt.)
    float ps[4];
    int i = 1, jv = nt0 > 0 ? nk0 : nk1;
    auto next_valid = [&](int ii) { jv -= 64; if (ii == nt0 && nt1 > 0) jv = nk1; };   // called after iteration ii: tile ii's count
    for (; i + 1 <= nt; i += 2) {
        const char* buf = sync_stage(i);
        block(jv, buf, sA, sB, ps);
        tail(sA, sB, ps);
        next_valid(i);
        buf = sync_stage(i + 1);
        block(jv, buf, sB, sA, ps);
        tail(sB, sA, ps);
        next_valid(i + 1);
    }
    if (i <= nt) {                                        // one iteration left
        const char* buf = sync_stage(i);
        block(jv, buf, sA, sB, ps);
        tail(sA, sB, ps);
    }
    // ---- iteration nt + 1: PV(nt-1) only, then normalise and store (lane holds O[q][h*64 + db*32 + 8g + 4u + j]) ----
    if (nt > 0) {
        const char* buf = sync_stage(nt + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            oacc[0] = VT<T>::mfma(v_frag(buf, 0, ks), P[ks], oacc[0]);
            oacc[1] = VT<T>::mfma(v_frag(buf, 1, ks), P[ks], oacc[1]);
        }
    }
    const float lt = xhalf_sum(l_run);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    if (q_row < p.Nq) {
        T* op = (T*)p.out + ((size_t)b * p.Nq + q_row) * p.ldo + h * 64 + 8 * u;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                v4 o[2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[k][j] = (T)(oacc[db][8 * gp + 4 * k + j] * inv);
                store_cols8(op + db * 32 + 16 * gp, o[0], o[1]);
            }
    }
}

template <typename T, int MODE>
static int launch_attn(AttnParams& p, int tune, hipStream_t st) {
    // 32 query rows per wave.  Pick waves/block so the grid has >= ~2 workgroups per CU when the problem allows it.
    // tune = (flags << 24) | (kernel << 16) | (ST << 8) | waves overrides (tuning table / tests); ST: 2 = two-buffer loop, 3/4 = LDS ring;
    // kernel = 2 | 3: attn_pp_kernel (3: one workgroup per CU, deep fragment prefetch); flags: bit0 pair adjacent waves, bit1 no
    // setprio, bits 2-3 rescale-threshold selector.  kernel = 7 | 8: attn_pf_kernel (row sums on the matrix pipe | on the VALU).  kernel = 16:
    // attn_sp_kernel, waves 8 | 4 (256 | 128 query rows per workgroup), flags bits 2-3 = row-sum limit selector.
    // No tune: 8 waves (256 query rows per workgroup: fewest K/V re-reads) while that still gives most CUs a workgroup,
    // else 4 waves with the 3-stage ring, 2 waves only for tiny problems (measured rule, profiles/r01_tune_report_*.json).
    const long bh = (long)p.B * p.heads;
    int nw = 4, stg = 3;
    if (bh * ((p.Nq + 255) / 256) >= 200) { nw = 8; stg = 2; }
    else if (bh * ((p.Nq + 127) / 128) < 64 && p.Nq <= 64) { nw = 2; stg = 2; }
    if (tune) { nw = tune & 0xff; stg = (tune >> 8) & 0xff; }
    else if (MODE == IDMVTON_ATTN_SELF && p.q_prescaled && p.Nq >= 128) {
        // measured (profiles/r06_attention_sp_tuner_lines.txt, r06_attention_variants.txt): the software-pipelined kernel, 256 query rows per
        // workgroup when that still gives every CU several of them, else 128 (two workgroups per CU, long + short walks paired)
        nw = (bh * ((p.Nq + 255) / 256) >= 400 && p.Nq >= 2048) ? 8 : 4; stg = 3;
        tune = (16 << 16) | (3 << 8) | nw;
    } else if (MODE == IDMVTON_ATTN_SELF && bh * ((p.Nq + 255) / 256) >= 200 && p.Nq >= 1024) {
        tune = (2 << 16) | (2 << 8) | 8; nw = 8; stg = 2;   // measured (profiles/r02_probe_attn_*): the ping-pong kernel wins on the long walks
    }
    if (((tune >> 16) & 0xff) == 2 || ((tune >> 16) & 0xff) == 3) {   // ping-pong kernel (3: one workgroup per CU, deep fragment prefetch)
        const bool deep = ((tune >> 16) & 0xff) == 3 && p.q_prescaled;   // the deep build needs q pre-multiplied (see NEGM)
        if (MODE != IDMVTON_ATTN_SELF || nw != 8 || (stg != 2 && stg != 3))
            return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: the ping-pong kernel needs SELF mode, 8 waves, 2 or 3 stages");
        static const float thr_tab[4] = {4.f, 0.f, 8.f, 2.f};
        p.pp_flags = (tune >> 24) & 3;
        p.pp_thr = thr_tab[(tune >> 26) & 3];
        dim3 gridp;
        attn_grid(p, 256, gridp);
        const dim3 blockp(512);
        // (s_setprio around the MFMA block was measured: no effect in this structure; the 128-register build exists with 2 stages)
        if (!deep) hipLaunchKernelGGL((attn_pp_kernel<T, 2, false>), gridp, blockp, 0, st, p);
        else if (stg == 2) hipLaunchKernelGGL((attn_pp_kernel<T, 2, true>), gridp, blockp, 0, st, p);
        else hipLaunchKernelGGL((attn_pp_kernel<T, 3, true>), gridp, blockp, 0, st, p);
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
    if (((tune >> 16) & 0xff) >= 4 && ((tune >> 16) & 0xff) <= 6) {   // ablation builds of the ping-pong kernel (timing only)
        p.pp_flags = 0; p.pp_thr = 4.f;
        dim3 gridp;
        attn_grid(p, 256, gridp);
        if (MODE != IDMVTON_ATTN_SELF) return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: ablation kernels are SELF mode only");
        if (((tune >> 16) & 0xff) == 4) hipLaunchKernelGGL((attn_pp_kernel<T, 2, false, 1>), gridp, dim3(512), 0, st, p);
        else if (((tune >> 16) & 0xff) == 5) hipLaunchKernelGGL((attn_pp_kernel<T, 2, false, 2>), gridp, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((attn_pp_kernel<T, 2, false, 3>), gridp, dim3(512), 0, st, p);
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
    if (((tune >> 16) & 0xff) == 7 || ((tune >> 16) & 0xff) == 8) {   // attn_pf_kernel: fragments prefetched a phase early (7: row sums on the matrix pipe)
        if (MODE != IDMVTON_ATTN_SELF || !p.q_prescaled)
            return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: the prefetch kernel needs SELF mode and a pre-multiplied q");
        static const float thr_tab[4] = {4.f, 0.f, 8.f, 2.f};
        p.pp_flags = 0;
        p.pp_thr = thr_tab[(tune >> 26) & 3];
        dim3 gridp;
        attn_grid(p, 256, gridp);
        if (((tune >> 16) & 0xff) == 7) hipLaunchKernelGGL((attn_pf_kernel<T, true>), gridp, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((attn_pf_kernel<T, false>), gridp, dim3(512), 0, st, p);
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
    if (((tune >> 16) & 0xff) == 16) {   // attn_sp_kernel (software-pipelined, speculative exponentials, row-sum overflow test)
        if (MODE != IDMVTON_ATTN_SELF || !p.q_prescaled)
            return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: the software-pipelined kernel needs SELF mode and a pre-multiplied q");
        static const float lim_tab[4] = {512.f, 32.f, 8192.f, 128.f};   // row-sum limits (tune bits 26-27): P <= limit while the max is kept
        p.pp_thr = lim_tab[(tune >> 26) & 3];
        dim3 gridp;
        if (nw == 4) {                                   // 128 query rows per workgroup, two workgroups per CU: finer units for short / uneven walks
            sp_grid(p, 128, gridp);
            hipLaunchKernelGGL((attn_sp_kernel<T, 4>), gridp, dim3(256), 0, st, p);
        } else {
            sp_grid(p, 256, gridp);
            hipLaunchKernelGGL((attn_sp_kernel<T, 8>), gridp, dim3(512), 0, st, p);
        }
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
#ifdef ATTN_DBG
    if (((tune >> 16) & 0xff) >= 32 && ((tune >> 16) & 0xff) < 160) {   // ablation builds of attn_sp_kernel<row sums on the VALU> (timing only): selector - 32 = mask
        p.pp_flags = 0; p.pp_thr = 512.f;                               // 1 no exponentials, 2 no MFMA, 4 no LDS reads, 8 no in-loop DMA, 16 no barrier
        dim3 gridp;
        sp_grid(p, 256, gridp);
        switch (((tune >> 16) & 0xff) - 32) {                            // + 32 / + 64: every wave on group 0's / group 1's schedule (no rotation)
#define SPA(m) case m: hipLaunchKernelGGL((attn_sp_kernel<T, 8, m>), gridp, dim3(512), 0, st, p); break;
        SPA(0) SPA(1) SPA(2) SPA(3) SPA(4) SPA(8) SPA(12) SPA(15) SPA(16) SPA(31) SPA(29)
#undef SPA
        default: return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: ablation mask not built");
        }
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
    if (((tune >> 16) & 0xff) >= 9 && ((tune >> 16) & 0xff) <= 14) {   // anatomy builds of attn_pf_kernel<LSUM>: ABL = selector - 8
        p.pp_flags = 0; p.pp_thr = 4.f;
        dim3 gridp;
        attn_grid(p, 256, gridp);
        switch ((tune >> 16) & 0xff) {
        case 9: hipLaunchKernelGGL((attn_pf_kernel<T, true, 1>), gridp, dim3(512), 0, st, p); break;
        case 10: hipLaunchKernelGGL((attn_pf_kernel<T, true, 2>), gridp, dim3(512), 0, st, p); break;
        case 11: hipLaunchKernelGGL((attn_pf_kernel<T, true, 3>), gridp, dim3(512), 0, st, p); break;
        case 12: hipLaunchKernelGGL((attn_pf_kernel<T, true, 4>), gridp, dim3(512), 0, st, p); break;
        case 13: hipLaunchKernelGGL((attn_pf_kernel<T, true, 5>), gridp, dim3(512), 0, st, p); break;
        default: hipLaunchKernelGGL((attn_pf_kernel<T, true, 6>), gridp, dim3(512), 0, st, p); break;
        }
        CHECK_LAUNCH("attn_fwd");
        return IDMVTON_OK;
    }
#endif
    if ((tune >> 16) & 0xff) return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: unknown kernel selector in tune");
    dim3 grid;
    attn_grid(p, 32 * nw, grid);
    const dim3 block(nw * 64);
#define ATTN_CASE(NW_, ST_) if (nw == NW_ && stg == ST_) { hipLaunchKernelGGL((attn_kernel<T, MODE, NW_, ST_>), grid, block, 0, st, p); } else
    ATTN_CASE(2, 2) ATTN_CASE(4, 2) ATTN_CASE(8, 2)
    ATTN_CASE(2, 3) ATTN_CASE(4, 3) ATTN_CASE(8, 3)
    ATTN_CASE(2, 4) ATTN_CASE(4, 4) ATTN_CASE(8, 4)
    return idmvton_set_error(IDMVTON_E_ARG, "attn_fwd: unsupported tune (waves=%d stages=%d)", nw, stg);
#undef ATTN_CASE
    CHECK_LAUNCH("attn_fwd");
    return IDMVTON_OK;
}

extern "C" int idmvton_attn_fwd(const idmvton_attn_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "attn_fwd: null args");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "attn_fwd: dtype %d", a->dtype);
    CHECK_ARG(a->mode == IDMVTON_ATTN_SELF || a->mode == IDMVTON_ATTN_CROSS, IDMVTON_E_ARG, "attn_fwd: mode %d", a->mode);
    CHECK_ARG(a->B > 0 && a->heads > 0 && a->Nq > 0, IDMVTON_E_SHAPE, "attn_fwd: B=%d heads=%d Nq=%d", a->B, a->heads, a->Nq);
    CHECK_ARG(a->nseg >= 1 && a->nseg <= 2, IDMVTON_E_SHAPE, "attn_fwd: nseg=%d", a->nseg);
    if (a->mode == IDMVTON_ATTN_CROSS) CHECK_ARG(a->nseg == 2 && a->seg_b0[0] == 0 && a->seg_b0[1] == 0, IDMVTON_E_ARG,
                                                 "attn_fwd: CROSS needs two segments present for every batch");
    CHECK_ARG(a->q && a->out && a->ldq % 8 == 0 && a->ldo % 8 == 0 && ((uintptr_t)a->q & 15) == 0 && ((uintptr_t)a->out & 15) == 0,
              IDMVTON_E_ALIGN, "attn_fwd: q/out alignment (ldq=%d ldo=%d)", a->ldq, a->ldo);
    CHECK_ARG(a->ldq >= a->heads * 64 && a->ldo >= a->heads * 64, IDMVTON_E_SHAPE, "attn_fwd: ldq/ldo < heads*64");
    AttnParams p;
    p.B = a->B; p.heads = a->heads; p.Nq = a->Nq; p.q = a->q; p.ldq = a->ldq; p.out = a->out; p.ldo = a->ldo;
    p.nseg = a->nseg; p.ip_scale = a->ip_scale; p.nqb = 0; p.chunk = p.parts = 1; p.pp_flags = 0; p.pp_thr = 4.f; p.q_prescaled = a->q_prescaled ? 1 : 0;
    const uint64_t qb = ((uint64_t)a->B * a->Nq - 1) * a->ldq * 2 + (uint64_t)a->heads * 128;
    CHECK_ARG(qb < 0x80000000ull, IDMVTON_E_SHAPE, "attn_fwd: Q >= 2 GiB");
    p.qbytes = (uint32_t)qb;
    for (int s = 0; s < 2; ++s) {
        const int ss = s < a->nseg ? s : 0;
        CHECK_ARG(a->k[ss] && a->vt[ss] && a->nk[ss] > 0 && a->seg_b0[ss] >= 0 && a->seg_b0[ss] <= a->B,
                  IDMVTON_E_SHAPE, "attn_fwd: seg %d nk=%d b0=%d", ss, a->nk[ss], a->seg_b0[ss]);
        const int krows = a->k_rows[ss] > 0 ? a->k_rows[ss] : a->nk[ss];
        CHECK_ARG(krows >= a->nk[ss], IDMVTON_E_SHAPE, "attn_fwd: seg %d k_rows=%d < nk", ss, krows);
        CHECK_ARG(a->ldk[ss] % 8 == 0 && a->ldvt[ss] % 16 == 0 && a->ldvt[ss] >= ((a->nk[ss] + 15) & ~15) && a->ldk[ss] >= a->heads * 64 &&
                  ((uintptr_t)a->k[ss] & 15) == 0 && ((uintptr_t)a->vt[ss] & 15) == 0, IDMVTON_E_ALIGN,
                  "attn_fwd: seg %d ldk=%d ldvt=%d (V^T is read in key order: ldvt %% 16 == 0, ldvt >= roundup16(nk))", ss, a->ldk[ss], a->ldvt[ss]);
        const int nb = a->B - a->seg_b0[ss];
        const uint64_t kb = (uint64_t)nb * krows * a->ldk[ss] * 2, vb = (uint64_t)nb * a->heads * 64 * a->ldvt[ss] * 2;
        CHECK_ARG(kb < 0x80000000ull && vb < 0x80000000ull, IDMVTON_E_SHAPE, "attn_fwd: seg %d K/V^T >= 2 GiB", ss);
        p.k[s] = a->k[ss]; p.ldk[s] = a->ldk[ss]; p.kbytes[s] = (uint32_t)kb;
        p.vt[s] = a->vt[ss]; p.ldvt[s] = a->ldvt[ss]; p.vtbytes[s] = (uint32_t)vb;
        p.nk[s] = a->nk[ss]; p.krows[s] = krows; p.seg_b0[s] = a->seg_b0[ss];
    }
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == IDMVTON_BF16)
        return a->mode == IDMVTON_ATTN_SELF ? launch_attn<bf16_t, IDMVTON_ATTN_SELF>(p, a->tune, st) : launch_attn<bf16_t, IDMVTON_ATTN_CROSS>(p, a->tune, st);
    return a->mode == IDMVTON_ATTN_SELF ? launch_attn<f16_t, IDMVTON_ATTN_SELF>(p, a->tune, st) : launch_attn<f16_t, IDMVTON_ATTN_CROSS>(p, a->tune, st);
}
