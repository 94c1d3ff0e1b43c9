// xattn.cuh -- cross-attention fused into the epilogue of its query projection (include/idmvton_hip.h, IDMVTON_EPI_XATTN).
//
// attn2 of a BasicTransformerBlock (ip_adapter/attention_processor.py:1943-1995; GarmentNet: diffusers Attention, one segment) is
// q = to_q(norm2(h)) -> softmax(q k_text^T / 8) v_text + ip_scale * softmax(q k_ip^T / 8) v_ip with 77 + 16 step-invariant keys.
// As separate launches: the projection writes q (7.8 MB at M = 3072), a 14 us attention launch reads it back and writes 7.8 MB
// more -- 60 + 10 launches per TryonNet step, ~1.3 ms of a 43 ms step, for 1.5 GFLOP of matrix work each.  Here the projection's
// accumulators never leave the registers:
//   * a wave of a 64-column (n) sub-tile holds q for ONE head (64 channels) and 32 query rows per accumulator pair, in the layout
//     lane = query row, registers = channels -- which IS the B operand of S^T = K . Q^T (csrc/attention.hip) up to the order of the
//     contraction index: accumulator position P = 16s + 8 (idx >> 2) + 4u + (idx & 3) sits in k-slot (u, idx) of k-step s, while K's
//     16-byte fragment has channel 16s + 8u + idx there.  The HOST permutes to_q's output rows (bits 2 and 3 of the channel index
//     swapped inside every group of 16 -- the same involution as the attention kernel's V^T key order) so that slot and channel agree;
//   * K / V^T fragments (<= 96 + 32 keys x 64 channels per head) are read straight from the step-invariant tables (L2 resident),
//     16 bytes per lane, no LDS; one-pass softmax per segment (every logit of a row is in registers), fp32 statistics;
//   * O^T = V^T . P^T lands in the accumulator layout the projection started with, so the ordinary 16-byte store epilogue writes
//     the attention output where q would have gone.
#pragma once
#include "common.cuh"

struct XAttnParams {
    const void* k[2]; const void* vt[2];                 // K [B][rows][ldk] (head h at columns h*64..), V^T [B][heads*64][ldvt] in key order
    int ldk[2], ldvt[2], nk[2], krows[2];                // krows: rows per batch element in k (>= round32(nk): padded rows are read, then masked)
    int nseg, tokens, vchan;                             // tokens: GEMM rows per batch element (% 32 == 0); vchan: rows of V^T per batch element (= heads * 64)
    float ip_scale;
};

#define XA_NEG (-1.0e30f)
#define XA_NK 16                                         // K fragments per (batch, head): 3 + 1 key blocks x 4 k-steps
#define XA_NV 16                                         // V^T fragments: (6 + 2) 16-key steps x 2 channel halves

// K fragments of this wave's (batch b, head h): issued BEFORE the projection's main loop, so their L2 latency is spent under it
template <typename T>
__device__ __forceinline__ void xattn_load_k(const XAttnParams& xa, const int b, const int h, const int lane, typename VT<T>::v8 (&kf)[XA_NK]) {
    typedef typename VT<T>::v8 v8;
    const int u = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const bool on = sg < xa.nseg;
        const int nkb = on ? (xa.nk[sg] + 31) >> 5 : 0;
        const T* kp = (const T*)xa.k[sg] + ((size_t)b * xa.krows[sg] + l31) * xa.ldk[sg] + h * 64 + 8 * u;
#pragma unroll
        for (int kb = 0; kb < (sg == 0 ? 3 : 1); ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v8 z;
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = (T)0.f;
                kf[(sg == 0 ? kb : 3) * 4 + s] = kb < nkb ? *(const v8*)(kp + (size_t)kb * 32 * xa.ldk[sg] + 16 * s) : z;
            }
    }
}
// V^T fragments (key order: the 8 keys of k-slot half u are one 16-byte read), issued at the top of the epilogue, consumed after the softmax
template <typename T>
__device__ __forceinline__ void xattn_load_v(const XAttnParams& xa, const int b, const int h, const int lane, typename VT<T>::v8 (&vf)[XA_NV]) {
    typedef typename VT<T>::v8 v8;
    const int u = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const bool on = sg < xa.nseg;
        const int nks = on ? (xa.nk[sg] + 15) >> 4 : 0;
        const T* vp = (const T*)xa.vt[sg] + ((size_t)b * xa.vchan + h * 64 + l31) * xa.ldvt[sg] + 8 * u;
#pragma unroll
        for (int ks = 0; ks < (sg == 0 ? 6 : 2); ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                v8 z;
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = (T)0.f;
                vf[((sg == 0 ? ks : 6 + ks)) * 2 + db] = ks < nks ? *(const v8*)(vp + (size_t)db * 32 * xa.ldvt[sg] + 16 * ks) : z;
            }
    }
}

// acc[ni][mi] (ni = 0, 1: the head's two 32-channel halves) of this wave: in = q (accumulators of the projection), out = attention output.
// Every 32-row block mi of the wave belongs to batch element b (the host checks tokens % (32 MI) == 0).
template <typename T, int MI>
__device__ __forceinline__ void xattn_compute(const XAttnParams& xa, const int M, f32x16 (&acc)[2][MI], const int m_wave, const int lane,
                                              const typename VT<T>::v8 (&kf)[XA_NK], const typename VT<T>::v8 (&vf)[XA_NV]) {
    typedef typename VT<T>::v8 v8;
    const int u = lane >> 5;
    const float cs = 0.125f * 1.44269504088896341f;     // softmax scale (d = 64) folded with log2(e), applied to the fp32 logits
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (m_wave + mi * 32 >= M) continue;             // wave-uniform
        v8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) qf[s][i] = (T)acc[s >> 1][mi][8 * (s & 1) + i];
        f32x16 of[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) of[db][r] = 0.f;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            if (sg >= xa.nseg) break;
            constexpr int NKB_MAX = 3;
            const int nkbmax = sg == 0 ? 3 : 1;
            const int nk = xa.nk[sg];
            // ---- S^T = K . Q^T, all keys of the segment ----
            f32x16 sacc[NKB_MAX];
#pragma unroll
            for (int kb = 0; kb < NKB_MAX; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = kb < nkbmax ? 0.f : XA_NEG;
                if (kb < nkbmax) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) sacc[kb] = VT<T>::mfma(kf[(sg == 0 ? kb : 3) * 4 + s], qf[s], sacc[kb]);
                }
            }
            // ---- one-pass softmax over this lane's query row (32 keys per block: 16 here, 16 in lane ^ 32) ----
            float mx = XA_NEG;
#pragma unroll
            for (int kb = 0; kb < NKB_MAX; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= nk) sacc[kb][r] = XA_NEG;
                    mx = fmaxf(mx, sacc[kb][r]);
                }
            mx = xhalf_max(mx) * cs;
            float psum = 0.f;
            v8 pf[2 * NKB_MAX];
#pragma unroll
            for (int kb = 0; kb < NKB_MAX; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], cs, -mx));      // masked keys: exp2(-1e30 * cs - mx) = 0
                    psum += pv;
                    pf[kb * 2 + (r >> 3)][r & 7] = (T)pv;
                }
            const float inv = (sg == 1 ? xa.ip_scale : 1.0f) / xhalf_sum(psum);
            // ---- O^T = V^T . P^T ----
            f32x16 oacc[2];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
            const int nks = (nk + 15) >> 4;              // 16-key steps that hold at least one real key
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                if (ks < (sg == 0 ? 6 : 2) && ks < nks) {
#pragma unroll
                    for (int db = 0; db < 2; ++db) oacc[db] = VT<T>::mfma(vf[(sg == 0 ? ks : 6 + ks) * 2 + db], pf[ks], oacc[db]);
                }
            }
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) of[db][r] = fmaf(oacc[db][r], inv, of[db][r]);
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) acc[db][mi] = of[db];
    }
}
