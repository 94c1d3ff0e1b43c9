// elementwise.hip -- small fused HBM/latency-bound ops of the denoising loop body and the pipeline edges, plus the C-ABI
// error plumbing and the hardware layout probes.  See include/idmvton_hip.h for the reference call sites.
#include "common.cuh"
#include <string.h>

static thread_local char g_err[512] = "";
int idmvton_set_error(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
extern "C" const char* idmvton_last_error(void) { return g_err; }
extern "C" int idmvton_abi_version(void) { return 9; }   // 9: idmvton_prefetch removed (both prefetch forms measured <= 0 in round 5 and have no caller); 8: gemm_conv tile_hint variant 6 (8-wave 128x128, 320x256), variant 3 and the LayerNorm-fold fields (rowstats_*, ln_*) removed; 7: IDMVTON_IO_OUT_F8 (gemm_conv writes e4m3 q / k / V^T for idmvton_attn_f8); 6: split-precision VAE path (idmvton_split, GN / softmax / layout flags, IDMVTON_IO_BIAS_F32), IDMVTON_MAX_SEG 24

// ---- TryonNet input: cat([latents]*2 | mask | masked | pose) -> NHWC[cpad] (tryon_pipeline.py:1769,1777) ----
template <typename T>
__global__ void pack_input_kernel(const idmvton_pack_input_args a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (b2, pixel)
    const int total = 2 * a.B * a.hw;
    if (idx >= total) return;
    const int b2 = idx / a.hw, pix = idx - b2 * a.hw;
    const int b = b2 % a.B;                                     // both CFG halves see the same latents
    T* o = (T*)a.out + (size_t)idx * a.cpad;
    const T* cond = (const T*)a.cond + (size_t)idx * 9;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (T)a.latents[((size_t)b * 4 + c) * a.hw + pix];
#pragma unroll
    for (int c = 0; c < 9; ++c) o[4 + c] = cond[c];
    for (int c = 13; c < a.cpad; ++c) o[c] = (T)0.f;
}
extern "C" int idmvton_pack_input(const idmvton_pack_input_args* a, void* stream) {
    CHECK_ARG(a && a->latents && a->cond && a->out, IDMVTON_E_ARG, "pack_input: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "pack_input: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->hw > 0 && a->cpad >= 13, IDMVTON_E_SHAPE, "pack_input: B=%d hw=%d cpad=%d", a->B, a->hw, a->cpad);
    const int total = 2 * a->B * a->hw;
    const dim3 grid((total + 255) / 256), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((pack_input_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((pack_input_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("pack_input");
    return IDMVTON_OK;
}

// ---- CFG combine + scheduler step (tryon_pipeline.py:1814-1823) ----
template <typename T>
__global__ void cfg_step_kernel(const idmvton_cfg_step_args a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (b, pixel)
    if (idx >= a.B * a.hw) return;
    const int b = idx / a.hw, pix = idx - b * a.hw;
    const float c_x = a.coef[0], c_eps = a.coef[1], sigma = a.coef[2], g = a.coef[3];
    const T* eu = (const T*)a.eps_nhwc + ((size_t)b * a.hw + pix) * a.ldc;
    const T* ec = (const T*)a.eps_nhwc + ((size_t)(a.B + b) * a.hw + pix) * a.ldc;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t o = ((size_t)b * 4 + c) * a.hw + pix;
        const float u = (float)eu[c], t = (float)ec[c];
        const float eps = u + g * (t - u);
        float x = c_x * a.latents[o] + c_eps * eps;
        if (a.noise) x += sigma * a.noise[o];
        a.latents[o] = x;
    }
}
extern "C" int idmvton_cfg_step(const idmvton_cfg_step_args* a, void* stream) {
    CHECK_ARG(a && a->eps_nhwc && a->latents && a->coef, IDMVTON_E_ARG, "cfg_step: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "cfg_step: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->hw > 0 && a->ldc >= 4, IDMVTON_E_SHAPE, "cfg_step: B=%d hw=%d ldc=%d", a->B, a->hw, a->ldc);
    const dim3 grid((a->B * a->hw + 255) / 256), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((cfg_step_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((cfg_step_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("cfg_step");
    return IDMVTON_OK;
}

// ---- NCHW fp32 <-> NHWC dtype (channel padded) ----
// flags: IDMVTON_LAYOUT_SPLIT (to_nhwc): dst pixel = [hi (cpad) | lo (cpad)]; IDMVTON_LAYOUT_NHWC_F32 (to_nchw): src NHWC holds fp32
template <typename T>
__global__ void layout_kernel(const idmvton_layout_args a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (b, pixel)
    if (idx >= (size_t)a.B * a.HW) return;
    const int b = (int)(idx / a.HW), pix = (int)(idx - (size_t)b * a.HW);
    if (a.to_nhwc) {
        const float* s = (const float*)a.src;
        if (a.flags & IDMVTON_LAYOUT_SPLIT) {
            T* d = (T*)a.dst + idx * 2 * a.cpad;
            for (int c = 0; c < a.C; ++c) {
                const float f = s[((size_t)b * a.C + c) * a.HW + pix] * a.scale + a.shift;
                const T hi = (T)f;
                d[c] = hi; d[a.cpad + c] = (T)(f - (float)hi);
            }
            for (int c = a.C; c < a.cpad; ++c) { d[c] = (T)0.f; d[a.cpad + c] = (T)0.f; }
            return;
        }
        T* d = (T*)a.dst + idx * a.cpad;
        for (int c = 0; c < a.C; ++c) d[c] = (T)(s[((size_t)b * a.C + c) * a.HW + pix] * a.scale + a.shift);
        for (int c = a.C; c < a.cpad; ++c) d[c] = (T)0.f;
    } else {
        float* d = (float*)a.dst;
        if (a.flags & IDMVTON_LAYOUT_NHWC_F32) {
            const float* s = (const float*)a.src + idx * a.cpad;
            for (int c = 0; c < a.C; ++c) d[((size_t)b * a.C + c) * a.HW + pix] = s[c] * a.scale + a.shift;
            return;
        }
        const T* s = (const T*)a.src + idx * a.cpad;
        for (int c = 0; c < a.C; ++c) d[((size_t)b * a.C + c) * a.HW + pix] = (float)s[c] * a.scale + a.shift;
    }
}
extern "C" int idmvton_layout(const idmvton_layout_args* a, void* stream) {
    CHECK_ARG(a && a->src && a->dst, IDMVTON_E_ARG, "layout: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "layout: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->C > 0 && a->HW > 0 && a->cpad >= a->C, IDMVTON_E_SHAPE, "layout: B=%d C=%d HW=%d cpad=%d", a->B, a->C, a->HW, a->cpad);
    CHECK_ARG((a->flags & ~3) == 0 && !((a->flags & IDMVTON_LAYOUT_SPLIT) && !a->to_nhwc) && !((a->flags & IDMVTON_LAYOUT_NHWC_F32) && a->to_nhwc),
              IDMVTON_E_ARG, "layout: flags=%d (SPLIT goes with to_nhwc, NHWC_F32 with the reverse)", a->flags);
    const size_t total = (size_t)a->B * a->HW;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((layout_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((layout_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("layout");
    return IDMVTON_OK;
}

// ---- fp32 -> [hi | lo] operand pairs of the split-precision GEMMs (include/idmvton_hip.h, idmvton_split) ----
template <typename T>
__global__ __launch_bounds__(256) void split_rows_kernel(const idmvton_split_args a) {
    typedef typename VT<T>::v8 v8;
    const int nchunk = a.cols >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;         // one thread per (row, 8-column chunk)
    if (idx >= (size_t)a.rows * nchunk) return;
    const int row = (int)(idx / nchunk), c = (int)(idx - (size_t)row * nchunk) * 8;
    const float4* p = (const float4*)(a.src + (size_t)row * a.lds + c);
    const float4 t0 = p[0], t1 = p[1];
    const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    v8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (T)v[j]; lo[j] = (T)(v[j] - (float)hi[j]); }
    T* d = (T*)a.dst + (size_t)row * a.ldd + c;
    *(v8*)d = hi;
    if (a.mode == IDMVTON_SPLIT_ACT) *(v8*)(d + a.cols) = lo;
    else { *(v8*)(d + a.cols) = hi; *(v8*)(d + 2 * a.cols) = lo; }
}
// W3T: dst[c][0..rows) = hi(src[.][c]), [rows..2 rows) = the same, [2 rows..3 rows) = lo: 64 x 64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void split_t_kernel(const idmvton_split_args a) {
    typedef typename VT<T>::v8 v8;
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < a.rows && c0 + c < a.cols) ? a.src[(size_t)(r0 + r) * a.lds + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {                  // (column c, 8-row chunk)
        const int c = i >> 3, rc = (i & 7) * 8;
        if (c0 + c >= a.cols || r0 + rc >= a.rows) continue;           // rows % 8 == 0: a chunk is inside or outside as a whole
        v8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = tile[rc + j][c]; hi[j] = (T)f; lo[j] = (T)(f - (float)hi[j]); }
        T* d = (T*)a.dst + (size_t)(c0 + c) * a.ldd + r0 + rc;
        *(v8*)d = hi; *(v8*)(d + a.rows) = hi; *(v8*)(d + 2 * a.rows) = lo;
    }
}
extern "C" int idmvton_split(const idmvton_split_args* a, void* stream) {
    CHECK_ARG(a && a->src && a->dst, IDMVTON_E_ARG, "split: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "split: dtype %d", a->dtype);
    CHECK_ARG(a->mode >= IDMVTON_SPLIT_ACT && a->mode <= IDMVTON_SPLIT_W3T, IDMVTON_E_ARG, "split: mode %d", a->mode);
    CHECK_ARG(a->rows > 0 && a->cols > 0 && a->lds >= a->cols, IDMVTON_E_SHAPE, "split: rows=%d cols=%d lds=%d", a->rows, a->cols, a->lds);
    CHECK_ARG(((uintptr_t)a->dst & 15) == 0 && a->ldd % 8 == 0, IDMVTON_E_ALIGN, "split: dst alignment / ldd=%d", a->ldd);
    hipStream_t st = (hipStream_t)stream;
    if (a->mode == IDMVTON_SPLIT_W3T) {
        CHECK_ARG(a->rows % 8 == 0 && a->ldd >= 3 * a->rows, IDMVTON_E_SHAPE, "split: W3T needs rows %% 8 == 0 and ldd >= 3 rows (rows=%d ldd=%d)", a->rows, a->ldd);
        const dim3 grid((a->cols + 63) / 64, (a->rows + 63) / 64), block(256);
        if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((split_t_kernel<bf16_t>), grid, block, 0, st, *a);
        else hipLaunchKernelGGL((split_t_kernel<f16_t>), grid, block, 0, st, *a);
    } else {
        const int k = a->mode == IDMVTON_SPLIT_ACT ? 2 : 3;
        CHECK_ARG(a->cols % 8 == 0 && a->lds % 4 == 0 && ((uintptr_t)a->src & 15) == 0 && a->ldd >= k * a->cols, IDMVTON_E_SHAPE,
                  "split: cols=%d lds=%d ldd=%d (cols %% 8 == 0, lds %% 4 == 0, ldd >= %d cols)", a->cols, a->lds, a->ldd, k);
        const size_t total = (size_t)a->rows * (a->cols >> 3);
        const dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((split_rows_kernel<bf16_t>), grid, block, 0, st, *a);
        else hipLaunchKernelGGL((split_rows_kernel<f16_t>), grid, block, 0, st, *a);
    }
    CHECK_LAUNCH("split");
    return IDMVTON_OK;
}

// ---- VAE posterior sample (tryon_pipeline.py:255) ----
template <typename T>
__global__ void vae_sample_kernel(const idmvton_vae_sample_args a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.B * a.hw) return;
    const int b = idx / a.hw, pix = idx - b * a.hw;
    const T* m = (const T*)a.moments + (size_t)idx * a.ldm;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const size_t o = ((size_t)b * 4 + c) * a.hw + pix;
        const float mean = (float)m[c];
        const float lv = fminf(fmaxf((float)m[4 + c], -30.f), 20.f);
        a.z[o] = (mean + __expf(0.5f * lv) * a.noise[o]) * a.scale;
    }
}
extern "C" int idmvton_vae_sample(const idmvton_vae_sample_args* a, void* stream) {
    CHECK_ARG(a && a->moments && a->noise && a->z, IDMVTON_E_ARG, "vae_sample: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "vae_sample: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->hw > 0 && a->ldm >= 8, IDMVTON_E_SHAPE, "vae_sample: B=%d hw=%d ldm=%d", a->B, a->hw, a->ldm);
    const dim3 grid((a->B * a->hw + 255) / 256), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((vae_sample_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((vae_sample_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("vae_sample");
    return IDMVTON_OK;
}

// ---- hardware layout probes: one wave, one instruction, raw per-lane operands in, raw per-lane results out ----
// which = 0: mfma_f32_32x32x16_bf16   a,b: [64 lanes][8] bf16 ; c: [64 lanes][16] f32
// which = 1: mfma_f32_32x32x16_f16
// which = 2: mfma_scale_f32_32x32x64_f8f6f4, A and B fp8 e4m3 (cbsz = blgp = 0), a,b: [64 lanes][32] bytes, unit E8M0 scales (127)
// which = 3: the same with scale A = 2^-3 (E8M0 byte 124) and scale B = 2^1 (128): D = (A . B) * 2^-2
typedef __attribute__((ext_vector_type(8))) int i32x8;
__global__ void probe_mfma_kernel(int which, const void* a, const void* b, float* c) {
    const int lane = threadIdx.x;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (which == 0) acc = VT<bf16_t>::mfma(((const bf16x8*)a)[lane], ((const bf16x8*)b)[lane], acc);
    else if (which == 2) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(((const i32x8*)a)[lane], ((const i32x8*)b)[lane], acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    else if (which == 3) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(((const i32x8*)a)[lane], ((const i32x8*)b)[lane], acc, 0, 0, 0, 0x7c7c7c7c, 0, (int)0x80808080);
    else acc = VT<f16_t>::mfma(((const f16x8*)a)[lane], ((const f16x8*)b)[lane], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[lane * 16 + r] = acc[r];
}
extern "C" int idmvton_probe_mfma(int which, const void* a, const void* b, float* c, void* stream) {
    CHECK_ARG(a && b && c && which >= 0 && which <= 3, IDMVTON_E_ARG, "probe_mfma: bad args");
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, which, a, b, c);
    CHECK_LAUNCH("probe_mfma");
    return IDMVTON_OK;
}

// ---- struct-size self description: lets the host-side binding verify its mirror of include/idmvton_hip.h ----
extern "C" int idmvton_sizeof(const char* name) {
#define SZ(n) if (!strcmp(name, #n)) return (int)sizeof(n);
    SZ(idmvton_seg) SZ(idmvton_gemm_conv_args) SZ(idmvton_attn_args) SZ(idmvton_layernorm_args)
    SZ(idmvton_groupnorm_args) SZ(idmvton_pack_input_args) SZ(idmvton_cfg_step_args) SZ(idmvton_layout_args)
    SZ(idmvton_vae_sample_args) SZ(idmvton_softmax_args) SZ(idmvton_attn_small_args) SZ(idmvton_attn_f8_args) SZ(idmvton_quant_f8_args) SZ(idmvton_split_args) SZ(idmvton_xattn)
#undef SZ
    return -1;
}
