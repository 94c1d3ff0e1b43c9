// gemm_common.cuh -- shared by the MFMA GEMM kernels (gemm_conv.hip: implicit-GEMM conv + Linear tiles; gemm_lin.hip: the
// hand-scheduled Linear main loop): kernel parameter block and the accumulator epilogue.
#pragma once
#include "common.cuh"
#include "xattn.cuh"

struct GemmParams {
    const void* w; uint32_t w_bytes; int N; int Ktot;
    int nseg; idmvton_seg seg[IDMVTON_MAX_SEG];
    int M, Ho, Wo, Hi, Wi, stride, ups;
    void* out; int ldo;
    const void* bias; const void* rowbias; int rowbias_ld; int rows_per_group;
    const void* res; int ldr;
    int mode;
    void* vt; int vt_n0; int vt_tokens; int vt_perm;
    int colscale_n; float colscale;
    int tiles_m, tiles_n;
    int gm;                                              // grouped raster: m-tiles per group (idmvton_choose_gm; 0 = 1024 rows)
    int wide;                                            // every epilogue operand allows 16-byte accesses at multiples of 8 columns
    int res32, out32;                                    // fp32 residual stream (io_flags): res read / out written as fp32 (wide only)
    int bias32;                                          // bias holds fp32 (plain 16-byte epilogue only): the split-precision VAE path
    int out8; float o8_scale, vt8_scale;                 // IDMVTON_IO_OUT_F8: out / vt are e4m3 bytes (plain 16-byte epilogue; vt in attention_f8.hip's slot order)
    XAttnParams xa;                                      // mode IDMVTON_EPI_XATTN: cross-attention applied to the accumulators (xattn.cuh)
};

// Raster group height for a launch of tiles_m x tiles_n tiles of bm x bn (gemm_conv.hip): the choice that makes the eight XCDs fetch the fewest
// operand panels, see there.
int idmvton_choose_gm(int tiles_m, int tiles_n, int bm, int bn, int K);

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Two accumulator column groups (g = 2gp, 2gp+1: columns 8g + 4u + j) of one 32x32 tile -> 8 CONSECUTIVE columns 16gp + 8u + (0..7)
// of the lane's row: lanes 32..63 of group 2gp trade places with lanes 0..31 of group 2gp+1 (v_permlane32_swap).  The epilogue is
// store-issue bound (one row per lane: every lane's store is its own memory segment), so 16-byte accesses halve its instruction
// count at the same bytes.  Both lanes of a pair (l, l+32) share their row, so they are active together.
__device__ __forceinline__ void swap_cols8(const f32x16& c, const int gp, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[8 * gp + j]), __float_as_uint(c[8 * gp + 4 + j]), false, false);
        v[j] = __uint_as_float(r[0]);
        v[4 + j] = __uint_as_float(r[1]);
    }
}

// Epilogue shared by every main loop: acc[ni][mi] is the wave's (SN x SM) sub-tile as NI x MI 32x32 accumulators (TR: D[m][n]).
template <typename T, int NI, int MI, int SN, int SM, bool TR>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[NI][MI], const int m0, const int n0, const int wn, const int wm,
                                              const int lane) {
    typedef typename VT<T>::v4 v4;
    typedef typename VT<T>::v8 v8;
    const int u = lane >> 5, l31 = lane & 31;
    const T* bias = (const T*)p.bias;
    if constexpr (TR) {
        if (p.out8) {
            // e4m3 V^T in the fp8 attention kernel's slot order: position 64t + 32u + 16kb + 4g + j <-> key 64t + 32kb + 8g + 4u + j.  The
            // lane's 16 accumulator registers 4g + j ARE rows 8g + 4u + j of this 32-row tile (kb = bit 5 of its first row), i.e. 16
            // consecutive positions: one 16-byte store per 32x32 tile and lane.
            uint8_t* vt = (uint8_t*)p.vt;
            const int Cv = p.N - p.vt_n0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = n0 + wn * SN + ni * 32 + l31;
                if (n >= p.N) continue;
                const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int m = m0 + wm * SM + mi * 32;                               // vt_tokens % 64 == 0: one batch element per 32 rows
                    if (m >= p.M) continue;
                    const int b = m / p.vt_tokens;
                    const int tok = m - b * p.vt_tokens;
                    int4 o;
                    int w[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        w[g] = pack4_fp8(clamp448((acc[ni][mi][4 * g] + bv) * p.vt8_scale), clamp448((acc[ni][mi][4 * g + 1] + bv) * p.vt8_scale),
                                         clamp448((acc[ni][mi][4 * g + 2] + bv) * p.vt8_scale), clamp448((acc[ni][mi][4 * g + 3] + bv) * p.vt8_scale));
                    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                    *(int4*)(vt + ((size_t)(b * Cv + n - p.vt_n0) * p.vt_tokens + (tok & ~63) + 32 * u + ((tok >> 1) & 16))) = o;
                }
            }
            return;
        }
        if (p.vt_perm) {
            // key order puts the tokens of accumulator groups g = 2gp, 2gp+1 (rows 8g + 4u + j) next to each other: 16 gp + 8u + 4(g&1) + j
            T* vt = (T*)p.vt;
            const int Cv = p.N - p.vt_n0;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = n0 + wn * SN + ni * 32 + l31;
                if (n >= p.N) continue;
                const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int m = m0 + wm * SM + mi * 32 + 16 * gp;                  // vt_tokens % 16 == 0: one batch per 16 rows
                        if (m >= p.M) continue;
                        const int b = m / p.vt_tokens;
                        const int tok = m - b * p.vt_tokens + 8 * u;
                        v8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (T)(acc[ni][mi][8 * gp + j] + bv);
                        *(v8*)(vt + ((size_t)(b * Cv + n - p.vt_n0) * p.vt_tokens + tok)) = o;
                    }
            }
            return;
        }
        // acc[ni][mi] = D[m][n]: column n = l31, rows m = 8g + 4u + j.  vt[(b*Cv + n - vt_n0)*tokens + tok..tok+3]
        T* vt = (T*)p.vt;
        const int Cv = p.N - p.vt_n0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * SN + ni * 32 + l31;
            if (n >= p.N) continue;
            const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + wm * SM + mi * 32 + 8 * g + 4 * u;
                    if (m >= p.M) continue;
                    const int b = m / p.vt_tokens;
                    int tok = m - b * p.vt_tokens;
                    // attention key order: bits 2 and 3 of the token index swapped inside every group of 16, so that the 8 keys a
                    // half-wave contracts in one PV MFMA (QK^T accumulator rows 8g+4u..+3, g = 0,1) are 16 contiguous bytes of V^T
                    if (p.vt_perm) tok = (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1);
                    v4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (T)(acc[ni][mi][4 * g + j] + bv);
                    *(v4*)(vt + ((size_t)(b * Cv + n - p.vt_n0) * p.vt_tokens + tok)) = o;
                }
        }
        return;
    }
    T* out = (T*)p.out;
    const T* res = (const T*)p.res;
    const T* rowbias = (const T*)p.rowbias;
    if (p.wide) {                                        // block-uniform: 16-byte loads / stores, 8 columns per lane and access
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * SM + mi * 32 + l31;
            if (m >= p.M) continue;
            const T* rb = rowbias ? rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
            if (p.mode == IDMVTON_EPI_GEGLU) {
                if constexpr (NI % 2 == 0) {
#pragma unroll
                    for (int pr = 0; pr < NI / 2; ++pr)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            float h[8], gt[8];
                            swap_cols8(acc[2 * pr][mi], gp, h);
                            swap_cols8(acc[2 * pr + 1][mi], gp, gt);
                            const int nh = n0 + wn * SN + pr * 64 + 16 * gp + 8 * u;   // h rows; gate rows are nh + 32
                            if (nh + 32 >= p.N) continue;
                            const int jo = ((n0 + wn * SN + pr * 64) >> 1) + 16 * gp + 8 * u;
                            if (bias) {
                                const v8 bh = *(const v8*)(bias + nh), bg = *(const v8*)(bias + nh + 32);
#pragma unroll
                                for (int j = 0; j < 8; ++j) { h[j] += (float)bh[j]; gt[j] += (float)bg[j]; }
                            }
                            v8 o;
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = (T)(h[j] * gelu_erf(gt[j]));
                            *(v8*)(out + (size_t)m * p.ldo + jo) = o;
                        }
                }
                continue;
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[8];
                    swap_cols8(acc[ni][mi], gp, v);
                    const int n = n0 + wn * SN + ni * 32 + 16 * gp + 8 * u;
                    if (n >= p.N) continue;
                    if (bias) {
                        if (p.bias32) {                    // block-uniform
                            const float4* bp = (const float4*)((const float*)p.bias + n);
                            const float4 b0 = bp[0], b1 = bp[1];
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        } else {
                            const v8 bb = *(const v8*)(bias + n);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)bb[j];
                        }
                    }
                    if (rb) {
                        const v8 bb = *(const v8*)(rb + n);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += (float)bb[j];
                    }
                    if (n < p.colscale_n) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] *= p.colscale;
                    }
                    if constexpr (NI * MI <= 8)            // (not in the 128x128-per-wave tile: the extra exit from this nest moves its accumulators to scratch)
                    if (p.out8) {                          // block-uniform: e4m3 operands for idmvton_attn_f8, one byte per column
                        int2 o;
                        o.x = pack4_fp8(clamp448(v[0] * p.o8_scale), clamp448(v[1] * p.o8_scale), clamp448(v[2] * p.o8_scale), clamp448(v[3] * p.o8_scale));
                        o.y = pack4_fp8(clamp448(v[4] * p.o8_scale), clamp448(v[5] * p.o8_scale), clamp448(v[6] * p.o8_scale), clamp448(v[7] * p.o8_scale));
                        *(int2*)((uint8_t*)p.out + (size_t)m * p.ldo + n) = o;
                        continue;
                    }
                    if (p.mode == IDMVTON_EPI_GELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
                    } else if (p.mode == IDMVTON_EPI_QUICKGELU) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = v[j] / (1.0f + __expf(-1.702f * v[j]));
                    }
                    if (res) {
                        if (p.res32) {                     // block-uniform: the fp32 residual stream
                            const float4* rp = (const float4*)((const float*)p.res + (size_t)m * p.ldr + n);
                            const float4 r0 = rp[0], r1 = rp[1];
                            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                            v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        } else {
                            const v8 rr = *(const v8*)(res + (size_t)m * p.ldr + n);
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += (float)rr[j];
                        }
                    }
                    if (p.out32) {
                        float4* op = (float4*)((float*)p.out + (size_t)m * p.ldo + n);
                        op[0] = make_float4(v[0], v[1], v[2], v[3]);
                        op[1] = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        v8 o;
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = (T)v[j];
                        *(v8*)(out + (size_t)m * p.ldo + n) = o;
                    }
                }
            }
        }
        return;
    }
    // (128x128-per-wave tiles, NI * MI = 16, exist only with the 16-byte epilogue: this nest no longer unrolls fully there and the
    //  runtime-indexed accumulators would move to scratch -- 1088 bytes per lane measured; launch_gemm falls back to the 8-wave tile)
    if constexpr (NI * MI <= 8)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * SM + mi * 32 + l31;
        if (m >= p.M) continue;
        const T* rb = rowbias ? rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
        if (p.mode == IDMVTON_EPI_GEGLU) {
            if constexpr (NI % 2 == 0) {
#pragma unroll
                for (int pr = 0; pr < NI / 2; ++pr)          // 64-row weight blocks [32 h | 32 gate] of this wave
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nh = n0 + wn * SN + pr * 64 + 8 * g + 4 * u;     // h rows; gate rows are nh + 32
                        if (nh + 32 >= p.N) continue;
                        const int jo = ((n0 + wn * SN + pr * 64) >> 1) + 8 * g + 4 * u;
                        v4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float h = acc[2 * pr][mi][4 * g + j], gt = acc[2 * pr + 1][mi][4 * g + j];
                            if (bias) { h += (float)bias[nh + j]; gt += (float)bias[nh + 32 + j]; }
                            o[j] = (T)(h * gelu_erf(gt));
                        }
                        *(v4*)(out + (size_t)m * p.ldo + jo) = o;
                    }
            }
            continue;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * SN + ni * 32 + 8 * g + 4 * u;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
                if (bias) {
                    const v4 bb = *(const v4*)(bias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)bb[j];
                }
                if (rb) {
                    const v4 bb = *(const v4*)(rb + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)bb[j];
                }
                if (n < p.colscale_n) {                    // e.g. the q columns of a fused QKV projection: softmax scale in fp32
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= p.colscale;
                }
                if (p.mode == IDMVTON_EPI_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
                } else if (p.mode == IDMVTON_EPI_QUICKGELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] / (1.0f + __expf(-1.702f * v[j]));
                }
                if (res) {
                    const v4 rr = *(const v4*)(res + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)rr[j];
                }
                v4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                *(v4*)(out + (size_t)m * p.ldo + n) = o;
            }
    }
}
