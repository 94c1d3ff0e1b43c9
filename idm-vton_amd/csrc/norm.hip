// norm.hip -- HBM-bound normalisation kernels: LayerNorm (one wave per token row, row kept in registers, two-pass
// statistics in fp32, optional second output = GarmentNet feature export) and NHWC GroupNorm(+SiLU) as
// stats (per-channel fp32 partials -> per-(block,batch,group) double partials, deterministic) + finalize + apply (16-byte vector loads/stores, the
// per-thread channel chunk's scale/shift held in registers).  See include/idmvton_hip.h for the reference call sites.
#include <cstdlib>
#include "common.cuh"

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per RPW rows (RPW = 1 is what runs, see launch_ln); lane l owns 8-element chunks l, l+64, l+128, l+192 (C <= 2048) of each.
// 16-byte loads, fp32 math.  The kernel is one latency chain per wave (loads -> reduce -> reduce -> stores) on data the previous GEMM left in
// L2 / Infinity Cache.
template <typename T, int NCH, bool X32, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const idmvton_layernorm_args a) {
    typedef typename VT<T>::v8 v8;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= a.rows) return;
    const int nchunk = a.C >> 3;
    const T* gamma = (const T*)a.gamma;
    const T* beta = (const T*)a.beta;
    v8 gm[NCH], bt_[NCH];                                // gamma / beta issued with the row loads: their L2 latency used to sit behind
#pragma unroll                                           // the two reductions, in front of the stores
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) { gm[i] = *(const v8*)(gamma + c * 8); bt_[i] = *(const v8*)(beta + c * 8); }
    }
    float v[RPW][NCH][8];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < a.rows ? row0 + r : a.rows - 1;       // a ragged last pair recomputes the last row (same values, same address)
        const T* x = (const T*)a.x + (size_t)row * a.ldx;
        const float* xf = (const float*)a.x + (size_t)row * a.ldx;      // X32: the fp32 residual stream
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                if constexpr (X32) {
                    const float4 t0 = *(const float4*)(xf + c * 8), t1 = *(const float4*)(xf + c * 8 + 4);
                    v[r][i][0] = t0.x; v[r][i][1] = t0.y; v[r][i][2] = t0.z; v[r][i][3] = t0.w;
                    v[r][i][4] = t1.x; v[r][i][5] = t1.y; v[r][i][6] = t1.z; v[r][i][7] = t1.w;
                } else {
                    const v8 t = *(const v8*)(x + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[r][i][j] = (float)t[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[r][i][j] = 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[r] += v[r][i][j];
    float mean[RPW], q[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) mean[r] = wave_sum_dpp(s[r]) / (float)a.C;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[r][i][j] - mean[r]; q[r] += d * d; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) rstd[r] = rsqrtf(wave_sum_dpp(q[r]) / (float)a.C + a.eps);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < a.rows ? row0 + r : a.rows - 1;
        T* y = (T*)a.y + (size_t)row * a.ldy;
        T* y2 = a.y2 ? (T*)a.y2 + (size_t)row * a.ldy2 : nullptr;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                v8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (T)((v[r][i][j] - mean[r]) * rstd[r] * (float)gm[i][j] + (float)bt_[i][j]);
                *(v8*)(y + c * 8) = o;
                if (y2) *(v8*)(y2 + c * 8) = o;
            }
        }
    }
}

template <typename T, bool X32, int RPW>
static void launch_ln_n(const idmvton_layernorm_args& a, hipStream_t st) {
    const dim3 grid((a.rows + 4 * RPW - 1) / (4 * RPW)), block(256);
    const int nch = ((a.C >> 3) + 63) / 64;
    if (nch <= 1) hipLaunchKernelGGL((layernorm_kernel<T, 1, X32, RPW>), grid, block, 0, st, a);
    else if (nch == 2) hipLaunchKernelGGL((layernorm_kernel<T, 2, X32, RPW>), grid, block, 0, st, a);
    else if (nch == 3) hipLaunchKernelGGL((layernorm_kernel<T, 3, X32, RPW>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((layernorm_kernel<T, 4, X32, RPW>), grid, block, 0, st, a);
}

template <typename T>
static int launch_ln(const idmvton_layernorm_args& a, hipStream_t st) {
    // one row per wave.  Two rows per wave (both rows' loads in flight before the first reduction) measured the same or slower on every shape
    // of the loop (profiles/r04_ln_rows_per_wave.log: 8.4-8.6 us at 3072 x 1280 either way, 10.5 vs 11.8 us at 9216 x 1280): the launch is a
    // latency floor -- dispatch, one L2 / Infinity-Cache round trip, two wave reductions, the store drain -- not a bandwidth problem
    if (a.x_f32) launch_ln_n<T, true, 1>(a, st);
    else launch_ln_n<T, false, 1>(a, st);
    CHECK_LAUNCH("layernorm");
    return IDMVTON_OK;
}

extern "C" int idmvton_layernorm(const idmvton_layernorm_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "layernorm: null args");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "layernorm: dtype %d", a->dtype);
    CHECK_ARG(a->rows > 0 && a->C > 0 && a->C % 8 == 0 && a->C <= 2048, IDMVTON_E_SHAPE, "layernorm: rows=%d C=%d (C%%8==0, C<=2048)", a->rows, a->C);
    CHECK_ARG(a->x && a->y && a->gamma && a->beta, IDMVTON_E_ARG, "layernorm: null pointer");
    CHECK_ARG(a->ldx % 8 == 0 && a->ldy % 8 == 0 && (!a->y2 || a->ldy2 % 8 == 0), IDMVTON_E_ALIGN, "layernorm: ld alignment");
    CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->y2 | (uintptr_t)a->gamma | (uintptr_t)a->beta) & 15) == 0,
              IDMVTON_E_ALIGN, "layernorm: pointers must be 16-byte aligned");
    return a->dtype == IDMVTON_BF16 ? launch_ln<bf16_t>(*a, (hipStream_t)stream) : launch_ln<f16_t>(*a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ GroupNorm (NHWC)
// Thread -> fixed 8-channel chunk (chunk = tid % (C/8)), striding over pixels.  A chunk may straddle two groups
// (e.g. C=320: 10 channels per group), so statistics are reduced per CHANNEL in LDS and folded to groups at the end.
// The reduction is DETERMINISTIC (no atomics): fixed pixel order per thread, fixed row order across the block's pixel
// lanes, one (sum, sumsq) partial per (block, batch, group) in HBM, folded in block order by gn_finalize_kernel.  Results
// are therefore bit-reproducible run to run (the parity tests compare execution modes bit for bit).
#define GN_MAXC 2560
// 8 consecutive channels of one pixel as fp32: from a 16-bit tensor (one 16-byte load) or, X32, from an fp32 one (two) -- the
// split-precision VAE path keeps its residual stream and convolution outputs in fp32 (include/idmvton_hip.h, IDMVTON_GN_*)
template <typename T, bool X32> struct GnLoad {
    typedef typename VT<T>::v8 v8;
    struct raw { v8 t; };
    static __device__ __forceinline__ raw ld(const void* base, size_t idx) { raw r; r.t = *(const v8*)((const T*)base + idx); return r; }
    static __device__ __forceinline__ float get(const raw& r, int j) { return (float)r.t[j]; }
};
template <typename T> struct GnLoad<T, true> {
    struct raw { float4 a, b; };
    static __device__ __forceinline__ raw ld(const void* base, size_t idx) {
        raw r; const float4* p = (const float4*)((const float*)base + idx); r.a = p[0]; r.b = p[1]; return r;
    }
    static __device__ __forceinline__ float get(const raw& r, int j) {
        return j == 0 ? r.a.x : j == 1 ? r.a.y : j == 2 ? r.a.z : j == 3 ? r.a.w : j == 4 ? r.b.x : j == 5 ? r.b.y : j == 6 ? r.b.z : r.b.w;
    }
};

template <typename T, bool X32>
__global__ __launch_bounds__(256) void gn_stats_kernel(const idmvton_groupnorm_args a, int pix_per_block, int nblk) {
    typedef GnLoad<T, X32> L;
    __shared__ float red_s[GN_MAXC], red_q[GN_MAXC];     // [pixel lane][channel] (tpp * C <= 2048) or [channel] (C > 2048)
    const int b = blockIdx.y;
    const int nchunk = a.C >> 3;
    const int tpp = 256 / nchunk > 0 ? 256 / nchunk : 1;   // pixels processed concurrently per pass
    const int chunk = threadIdx.x % nchunk, psub = threadIdx.x / nchunk;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, a.HW);
    // chunks beyond 256 threads (C > 2048, tpp == 1): each thread also owns chunk + 256
    for (int cb = chunk; cb < nchunk; cb += 256) {
        const int c0 = cb * 8;
        const void* src; int pitch, coff;
        if (c0 < a.C1) { src = a.x; pitch = a.C1; coff = c0; }
        else { src = a.x2; pitch = a.C - a.C1; coff = c0 - a.C1; }
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
        if (psub < tpp) {
            // four pixels' loads in flight per thread (the loop is latency bound: one dependent 16-byte load per iteration
            // otherwise); accumulation order per thread is unchanged, so results are bit-identical to the rolled loop
            int pix = p0 + psub;
            for (; pix + 3 * tpp < p1; pix += 4 * tpp) {
                typename L::raw t[4];
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) t[u4] = L::ld(src, ((size_t)b * a.HW + pix + u4 * tpp) * pitch + coff);
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = L::get(t[u4], j); s[j] += f; q[j] += f * f; }
            }
            for (; pix < p1; pix += tpp) {
                const typename L::raw t = L::ld(src, ((size_t)b * a.HW + pix) * pitch + coff);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = L::get(t, j); s[j] += f; q[j] += f * f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { red_s[psub * a.C + c0 + j] = s[j]; red_q[psub * a.C + c0 + j] = q[j]; }
        }
    }
    __syncthreads();
    // per-channel totals over the pixel lanes, in lane order; column c is touched by exactly one thread
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float ss = red_s[c], qq = red_q[c];
        for (int p = 1; p < tpp; ++p) { ss += red_s[p * a.C + c]; qq += red_q[p * a.C + c]; }
        red_s[c] = ss; red_q[c] = qq;
    }
    __syncthreads();
    const int cpg = a.C / a.groups;
    double* part = a.stats + (size_t)2 * a.B * a.groups;
    for (int g = threadIdx.x; g < a.groups; g += 256) {
        double ds = 0.0, dq = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { ds += (double)red_s[c]; dq += (double)red_q[c]; }
        double* dst = part + (((size_t)b * a.groups + g) * nblk + blockIdx.x) * 2;
        dst[0] = ds; dst[1] = dq;
    }
}

// one wave per (batch, group): block partials summed in a fixed order -> (mean, rstd) as two floats in stats[(b*groups + g)*2]
// (the double-precision division / square root happen ONCE per group here instead of 8 times per thread of the apply kernel)
__global__ __launch_bounds__(64) void gn_finalize_kernel(double* stats, int bg, int nblk, double cnt, double eps) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double* part = stats + (size_t)2 * bg + (size_t)i * nblk * 2;
    double ds = 0.0, dq = 0.0;
    int k = lane;
    for (; k + 3 * 64 < nblk; k += 4 * 64) {              // four independent loads in flight; same summation order
        double ps[4], pq[4];
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) { ps[u4] = part[2 * (k + 64 * u4)]; pq[u4] = part[2 * (k + 64 * u4) + 1]; }
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) { ds += ps[u4]; dq += pq[u4]; }
    }
    for (; k < nblk; k += 64) { ds += part[2 * k]; dq += part[2 * k + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o); dq += __shfl_xor(dq, o); }
    if (lane == 0) {
        const double mean = ds / cnt;
        double var = dq / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        float* o = (float*)(stats + 2 * i);
        o[0] = (float)mean;
        o[1] = (float)(1.0 / sqrt(var + eps));
    }
}

// PREC = the split-precision form (flags = X_F32 | Y_SPLIT | AFFINE_F32): fp32 input, fp32 gamma / beta, output [hi | lo] with pixel pitch 2C
template <typename T, bool PREC>
__global__ __launch_bounds__(256) void gn_apply_kernel(const idmvton_groupnorm_args a, int pix_per_block) {
    typedef typename VT<T>::v8 v8;
    typedef GnLoad<T, PREC> L;
    const int b = blockIdx.y;
    const int nchunk = a.C >> 3;
    const int tpp = 256 / nchunk > 0 ? 256 / nchunk : 1;
    const int chunk = threadIdx.x % nchunk, psub = threadIdx.x / nchunk;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, a.HW);
    const int cpg = a.C / a.groups;
    const int opitch = PREC ? 2 * a.C : a.C;
    for (int cb = chunk; cb < nchunk; cb += 256) {
        const int c0 = cb * 8;
        const void* src; int pitch, coff;
        if (c0 < a.C1) { src = a.x; pitch = a.C1; coff = c0; }
        else { src = a.x2; pitch = a.C - a.C1; coff = c0 - a.C1; }
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j, g = c / cpg;
            const float2 mr = *(const float2*)(a.stats + ((size_t)b * a.groups + g) * 2);      // (mean, rstd) from gn_finalize_kernel
            const float gm = PREC ? ((const float*)a.gamma)[c] : (float)((const T*)a.gamma)[c];
            const float bt = PREC ? ((const float*)a.beta)[c] : (float)((const T*)a.beta)[c];
            sc[j] = mr.y * gm;
            sh[j] = bt - mr.x * sc[j];
        }
        auto emit = [&](const typename L::raw& t, size_t row) {
            T* dst = (T*)a.y + row * opitch + c0;
            v8 o, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = L::get(t, j) * sc[j] + sh[j];
                if (a.silu) f = silu_f(f);
                o[j] = (T)f;
                if constexpr (PREC) lo[j] = (T)(f - (float)o[j]);
            }
            *(v8*)dst = o;
            if constexpr (PREC) *(v8*)(dst + a.C) = lo;
        };
        if (psub < tpp) {
            int pix = p0 + psub;
            for (; pix + 3 * tpp < p1; pix += 4 * tpp) {          // four pixels' loads in flight per thread
                typename L::raw t[4];
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) t[u4] = L::ld(src, ((size_t)b * a.HW + pix + u4 * tpp) * pitch + coff);
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) emit(t[u4], (size_t)b * a.HW + pix + u4 * tpp);
            }
            for (; pix < p1; pix += tpp) {
                const size_t row = (size_t)b * a.HW + pix;
                emit(L::ld(src, row * pitch + coff), row);
            }
        }
    }
}

static void gn_geometry(const idmvton_groupnorm_args& a, int& nblk, int& ppb) {
    // ~8 blocks per CU over the whole launch; each block owns a contiguous pixel slab of one batch element.
    nblk = (2048 + a.B - 1) / a.B;
    ppb = (a.HW + nblk - 1) / nblk;
    const int nchunk = a.C >> 3;
    const int tpp = 256 / nchunk > 0 ? 256 / nchunk : 1;
    if (ppb < tpp * 4) ppb = tpp * 4;
    nblk = (a.HW + ppb - 1) / ppb;
}

template <typename T>
static int launch_gn(const idmvton_groupnorm_args& a, hipStream_t st) {
    int nblk, ppb;
    gn_geometry(a, nblk, ppb);
    const dim3 grid(nblk, a.B), block(256);
    const bool prec = a.flags != 0;
    if (prec) hipLaunchKernelGGL((gn_stats_kernel<T, true>), grid, block, 0, st, a, ppb, nblk);
    else hipLaunchKernelGGL((gn_stats_kernel<T, false>), grid, block, 0, st, a, ppb, nblk);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B * a.groups), dim3(64), 0, st, a.stats, a.B * a.groups, nblk,
                       (double)(a.C / a.groups) * (double)a.HW, (double)a.eps);
    if (prec) hipLaunchKernelGGL((gn_apply_kernel<T, true>), grid, block, 0, st, a, ppb);
    else hipLaunchKernelGGL((gn_apply_kernel<T, false>), grid, block, 0, st, a, ppb);
    CHECK_LAUNCH("groupnorm");
    return IDMVTON_OK;
}

extern "C" int idmvton_groupnorm_stats_doubles(int B, int HW, int C, int groups) {
    if (B <= 0 || HW <= 0 || C < 8 || groups <= 0) return -1;
    idmvton_groupnorm_args a;
    a.B = B; a.HW = HW; a.C = C; a.groups = groups;
    int nblk, ppb;
    gn_geometry(a, nblk, ppb);
    return 2 * B * groups * (1 + nblk);
}

extern "C" int idmvton_groupnorm(const idmvton_groupnorm_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "groupnorm: null args");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "groupnorm: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->HW > 0 && a->C > 0 && a->groups > 0 && a->C % a->groups == 0 && a->C % 8 == 0 && a->C <= GN_MAXC,
              IDMVTON_E_SHAPE, "groupnorm: B=%d HW=%d C=%d groups=%d", a->B, a->HW, a->C, a->groups);
    CHECK_ARG(a->x && a->y && a->gamma && a->beta && a->stats, IDMVTON_E_ARG, "groupnorm: null pointer");
    CHECK_ARG(a->stats_doubles >= idmvton_groupnorm_stats_doubles(a->B, a->HW, a->C, a->groups), IDMVTON_E_SHAPE,
              "groupnorm: stats scratch holds %d doubles, needs %d", a->stats_doubles,
              idmvton_groupnorm_stats_doubles(a->B, a->HW, a->C, a->groups));
    CHECK_ARG(a->C1 > 0 && a->C1 <= a->C && a->C1 % 8 == 0 && (a->C1 == a->C || a->x2), IDMVTON_E_SHAPE, "groupnorm: C1=%d", a->C1);
    constexpr int PREC_FLAGS = IDMVTON_GN_X_F32 | IDMVTON_GN_Y_SPLIT | IDMVTON_GN_AFFINE_F32;
    CHECK_ARG(a->flags == 0 || a->flags == PREC_FLAGS, IDMVTON_E_ARG, "groupnorm: flags=%d (0, or the split-precision form X_F32 | Y_SPLIT | AFFINE_F32 = %d)", a->flags, PREC_FLAGS);
    if (a->flags) CHECK_ARG((((uintptr_t)a->gamma | (uintptr_t)a->beta) & 3) == 0, IDMVTON_E_ALIGN, "groupnorm: fp32 gamma / beta alignment");
    CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->x2 | (uintptr_t)a->y) & 15) == 0, IDMVTON_E_ALIGN, "groupnorm: pointer alignment");
    return a->dtype == IDMVTON_BF16 ? launch_gn<bf16_t>(*a, (hipStream_t)stream) : launch_gn<f16_t>(*a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ row softmax
// In-place softmax(scale * x) over rows of n elements (n % 8 == 0), fp32 statistics, one workgroup per row.
// Used by the VAE mid-block attention (single head, d = 512: unet_block_hacked_tryon.py:585-597, upcast_softmax=True).
// nv = columns that take part (<= n); columns nv..n-1 are written as 0 (key padding up to the next GEMM's K granule)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* x, int n, int ld, float scale, int nv) {
    typedef typename VT<T>::v8 v8;
    __shared__ float red[8];
    T* row = x + (size_t)blockIdx.x * ld;
    const int nchunk = n >> 3;
    const float sl = scale * 1.44269504088896341f;
    float mx = -3.0e38f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const v8 t = *(const v8*)(row + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, c * 8 + j < nv ? (float)t[j] : -3.0e38f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const v8 t = *(const v8*)(row + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += c * 8 + j < nv ? __builtin_amdgcn_exp2f(((float)t[j] - mx) * sl) : 0.f;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const v8 t = *(const v8*)(row + c * 8);
        v8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)(c * 8 + j < nv ? __builtin_amdgcn_exp2f(((float)t[j] - mx) * sl) * inv : 0.f);
        *(v8*)(row + c * 8) = o;
    }
}

// Split-precision form: logits fp32 [rows][ld] (read three times from L2), probabilities out as the pair [hi (n) | lo (n)] of T.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_split_kernel(const float* x, int n, int ld, float scale, T* y, int ldy, int nv) {
    typedef typename VT<T>::v8 v8;
    __shared__ float red[8];
    const float* row = x + (size_t)blockIdx.x * ld;
    T* yr = y + (size_t)blockIdx.x * ldy;
    const int nchunk = n >> 3;
    const float sl = scale * 1.44269504088896341f;
    float mx = -3.0e38f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const float4 t0 = *(const float4*)(row + c * 8), t1 = *(const float4*)(row + c * 8 + 4);
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, c * 8 + j < nv ? v[j] : -3.0e38f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const float4 t0 = *(const float4*)(row + c * 8), t1 = *(const float4*)(row + c * 8 + 4);
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) s += c * 8 + j < nv ? __builtin_amdgcn_exp2f((v[j] - mx) * sl) : 0.f;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = threadIdx.x; c < nchunk; c += 256) {
        const float4 t0 = *(const float4*)(row + c * 8), t1 = *(const float4*)(row + c * 8 + 4);
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        v8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = c * 8 + j < nv ? __builtin_amdgcn_exp2f((v[j] - mx) * sl) * inv : 0.f;
            hi[j] = (T)p;
            lo[j] = (T)(p - (float)hi[j]);
        }
        *(v8*)(yr + c * 8) = hi;
        *(v8*)(yr + n + c * 8) = lo;
    }
}

extern "C" int idmvton_softmax_rows(const idmvton_softmax_args* a, void* stream) {
    CHECK_ARG(a && a->x, IDMVTON_E_ARG, "softmax_rows: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "softmax_rows: dtype %d", a->dtype);
    CHECK_ARG(a->rows > 0 && a->n > 0 && a->n % 8 == 0 && (a->y_split || a->ld % 8 == 0) && a->ld >= a->n && ((uintptr_t)a->x & 15) == 0,
              IDMVTON_E_SHAPE, "softmax_rows: rows=%d n=%d ld=%d", a->rows, a->n, a->ld);
    const dim3 grid(a->rows), block(256);
    CHECK_ARG(a->n_valid >= 0 && a->n_valid <= a->n, IDMVTON_E_SHAPE, "softmax_rows: n_valid=%d n=%d", a->n_valid, a->n);
    const int nv = a->n_valid ? a->n_valid : a->n;
    if (a->y_split) {
        CHECK_ARG(a->ldy % 8 == 0 && a->ldy >= 2 * a->n && ((uintptr_t)a->y_split & 15) == 0 && a->ld % 4 == 0, IDMVTON_E_SHAPE, "softmax_rows: y_split ldy=%d n=%d", a->ldy, a->n);
        if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((softmax_rows_split_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const float*)a->x, a->n, a->ld, a->scale, (bf16_t*)a->y_split, a->ldy, nv);
        else hipLaunchKernelGGL((softmax_rows_split_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, (const float*)a->x, a->n, a->ld, a->scale, (f16_t*)a->y_split, a->ldy, nv);
        CHECK_LAUNCH("softmax_rows");
        return IDMVTON_OK;
    }
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (bf16_t*)a->x, a->n, a->ld, a->scale, nv);
    else hipLaunchKernelGGL((softmax_rows_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, (f16_t*)a->x, a->n, a->ld, a->scale, nv);
    CHECK_LAUNCH("softmax_rows");
    return IDMVTON_OK;
}
