// attention_f8.hip -- fp8 (OCP e4m3) self-attention on the block-scaled MFMA of gfx950, v_mfma_scale_f32_32x32x64_f8f6f4
// (BASELINE.json configs[4]: "fp16 + fp8 MFMA attention"; call site replaced: ip_adapter/attention_processor.py:258-260).
//
// Same swapped formulation as attention.hip (S^T = K.Q^T, O^T = V^T.P^T: a lane owns one query row, softmax is lane-local), with the
// K = 64 scaled MFMA: one instruction contracts the whole head dimension (d = 64) of a 32-key block, and one instruction contracts a
// whole 64-key tile for a 32-wide slice of d -- 4 MFMAs per 64-key tile instead of 16, at twice the bf16 rate.  Operand layout of the
// instruction (verified on hardware by tests/test_probe_gpu.py, probe 2): lane l supplies 32 consecutive k-slots, k = 32*(l>>5) + 0..31,
// of row (l & 31): 32 bytes = 8 VGPRs.  So
//   * Q, K stay in their natural [token][64 d] layout (one 32-byte load per lane and fragment),
//   * the PV contraction's k-slot <-> key assignment is chosen to be the one the S^T accumulator already has: lane (q, u) holds the
//     scores of keys 32 kb + 8 g + 4 u + j in register 16 kb + 4 g + j, and that register index IS its slot number, so P goes from the
//     accumulator through one cvt_pk_fp8 per pair straight into the B operand -- no shuffle;
//   * V^T is stored [channel][position] with position 64 t + 32 u + 16 kb + 4 g + j  <->  key 64 t + 32 kb + 8 g + 4 u + j
//     (idmvton_quant_f8 mode 1 produces it from the 16-bit V^T).
// Quantisation scales are powers of two and ride on the instruction's E8M0 scale operands (no per-element multiply): the caller
// multiplies q, k, v by 2^eq, 2^ek, 2^ev before the conversion (idmvton_quant_f8 `scale`) and passes qk_scale_exp = -(eq + ek),
// v_scale_exp = -ev; P (in [0, 1]) is converted as P * 2^8 -- e4m3 would flush every probability below 2^-9 otherwise, i.e. most of
// a 3072-key row -- and the PV MFMA's B scale carries the 2^-8.
// No LDS: each wave streams its own K / V^T fragments (fragment-shaped 32-byte loads, next tile prefetched into registers).  This
// variant exists for configs[4] and for the measurement behind DESIGN.md section 6 (the softmax VALU work per key is unchanged, so the
// halved matrix time does not shorten the loop); the bf16 / fp16 kernels of attention.hip remain the default.
#include "common.cuh"

typedef __attribute__((ext_vector_type(8))) int i32x8;

struct AttnF8Params {
    int B, heads, Nq;
    const uint8_t* q; int ldq;
    void* out; int ldo;
    int nseg;
    const uint8_t* k[2]; int ldk[2];
    const uint8_t* vt[2]; int ldvt[2];
    int nk[2]; int krows[2]; int seg_b0[2];
    int sc_qk, sc_v;                                   // E8M0 bytes (127 + exponent), replicated in all four bytes
    int nqb;
};

#define NEG_BIG_F8 (-1.0e30f)

__device__ __forceinline__ i32x8 load32(const uint8_t* p) {
    const int4 a = *(const int4*)p, b = *(const int4*)(p + 16);
    i32x8 r;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_f8_kernel(const AttnF8Params p) {
    typedef typename VT<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int u = lane >> 5, l31 = lane & 31;
    // block -> (batch, head, q-block): longest work (conditional batches: both segments present) first
    const int per = p.heads * p.nqb;
    const int b = p.B - 1 - blockIdx.x / per;
    const int rem = blockIdx.x % per;
    const int h = rem / p.nqb, qb = rem - h * p.nqb;
    const int q_row = qb * 128 + wave * 32 + l31;
    const int q_ld = q_row < p.Nq ? q_row : p.Nq - 1;
    const i32x8 qf = load32(p.q + ((size_t)b * p.Nq + q_ld) * p.ldq + h * 64 + 32 * u);

    const bool pres0 = p.nseg > 0 && b >= p.seg_b0[0];
    const bool pres1 = p.nseg > 1 && b >= p.seg_b0[1];
    const int nt0 = pres0 ? (p.nk[0] + 63) >> 6 : 0;
    const int nt1 = pres1 ? (p.nk[1] + 63) >> 6 : 0;
    const int nt = nt0 + nt1;

    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = NEG_BIG_F8, l_run = 0.f;             // l_run in units of 2^-8 (it sums the scaled probabilities)
    {
        int nz = 0;                                     // closed form for absent (all-zero) segments: nk keys, logit 0, value 0
        if (p.nseg > 0 && !pres0) nz += p.nk[0];
        if (p.nseg > 1 && !pres1) nz += p.nk[1];
        if (nz > 0) { m_run = 0.f; l_run = u == 0 ? 256.f * (float)nz : 0.f; }
    }

    i32x8 kf[2], vf[2];
    auto load_tile = [&](int t, i32x8 (&kk)[2], i32x8 (&vv)[2]) {
        const int sg = t < nt0 ? 0 : 1;
        const int kt = sg ? t - nt0 : t;
        const int bsg = b - p.seg_b0[sg];
        const uint8_t* kp = p.k[sg] + (size_t)bsg * p.krows[sg] * p.ldk[sg] + h * 64 + 32 * u;
        const uint8_t* vp = p.vt[sg] + ((size_t)bsg * p.heads * 64 + h * 64) * p.ldvt[sg] + kt * 64 + 32 * u;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            int key = kt * 64 + kb * 32 + l31;
            key = key < p.nk[sg] ? key : p.nk[sg] - 1;            // masked below; V^T positions beyond nk are zero-filled
            kk[kb] = load32(kp + (size_t)key * p.ldk[sg]);
            vv[kb] = load32(vp + (size_t)(kb * 32 + l31) * p.ldvt[sg]);
        }
    };
    if (nt > 0) load_tile(0, kf, vf);
    for (int t = 0; t < nt; ++t) {
        i32x8 kn[2], vn[2];
        if (t + 1 < nt) load_tile(t + 1, kn, vn);                 // next tile's fragments in flight behind this tile's math
        const int sg = t < nt0 ? 0 : 1;
        const int kt = sg ? t - nt0 : t;
        const int valid = p.nk[sg] - kt * 64;
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            sacc[kb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[kb], qf, z, 0, 0, 0, p.sc_qk, 0, 0x7f7f7f7f);
        }
        if (valid < 64) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * u;
                    if (key >= valid) sacc[kb][r] = NEG_BIG_F8;
                }
        }
        float mx = NEG_BIG_F8;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run, mx);
        float psum = 0.f;
        i32x8 pf;                                                  // slot 16 kb + r  <-  P of register r of key block kb, times 2^8
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { e[j] = __builtin_amdgcn_exp2f(sacc[kb][4 * r4 + j] - m_new + 8.0f); psum += e[j]; }
                pf[kb * 4 + r4] = pack4_fp8(e[0], e[1], e[2], e[3]);
            }
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        m_run = m_new;
        l_run += psum;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            oacc[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf[db], pf, oacc[db], 0, 0, 0, p.sc_v, 0, 0x77777777);   // B scale 2^-8
        if (t + 1 < nt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { kf[i] = kn[i]; vf[i] = vn[i]; }
        }
    }
    const float lt = xhalf_sum(l_run);
    const float inv = lt > 0.f ? 256.0f / lt : 0.f;
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

extern "C" int idmvton_attn_f8(const idmvton_attn_f8_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "attn_f8: null args");
    CHECK_ARG(a->out_dtype == IDMVTON_F16 || a->out_dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "attn_f8: out_dtype %d", a->out_dtype);
    CHECK_ARG(a->B > 0 && a->heads > 0 && a->Nq > 0 && a->nseg >= 1 && a->nseg <= 2, IDMVTON_E_SHAPE, "attn_f8: B=%d heads=%d Nq=%d nseg=%d", a->B, a->heads, a->Nq, a->nseg);
    CHECK_ARG(a->q8 && a->out && a->ldq % 16 == 0 && a->ldq >= a->heads * 64 && a->ldo % 8 == 0 && a->ldo >= a->heads * 64 &&
              ((uintptr_t)a->q8 & 15) == 0 && ((uintptr_t)a->out & 15) == 0, IDMVTON_E_ALIGN, "attn_f8: q8 / out alignment (ldq=%d ldo=%d)", a->ldq, a->ldo);
    CHECK_ARG(a->qk_scale_exp >= -100 && a->qk_scale_exp <= 100 && a->v_scale_exp >= -100 && a->v_scale_exp <= 100, IDMVTON_E_ARG, "attn_f8: scale exponents");
    AttnF8Params p;
    p.B = a->B; p.heads = a->heads; p.Nq = a->Nq; p.q = (const uint8_t*)a->q8; p.ldq = a->ldq; p.out = a->out; p.ldo = a->ldo; p.nseg = a->nseg;
    for (int s = 0; s < 2; ++s) {
        const int ss = s < a->nseg ? s : 0;
        CHECK_ARG(a->k8[ss] && a->vt8[ss] && a->nk[ss] > 0 && a->seg_b0[ss] >= 0 && a->seg_b0[ss] <= a->B, IDMVTON_E_SHAPE, "attn_f8: seg %d", ss);
        const int krows = a->k_rows[ss] > 0 ? a->k_rows[ss] : a->nk[ss];
        CHECK_ARG(krows >= a->nk[ss] && a->ldk[ss] % 16 == 0 && a->ldk[ss] >= a->heads * 64 && a->ldvt[ss] % 64 == 0 && a->ldvt[ss] >= ((a->nk[ss] + 63) & ~63) &&
                  ((uintptr_t)a->k8[ss] & 15) == 0 && ((uintptr_t)a->vt8[ss] & 15) == 0, IDMVTON_E_ALIGN,
                  "attn_f8: seg %d ldk=%d ldvt=%d (V^T rows hold whole 64-key tiles: ldvt %% 64 == 0, ldvt >= roundup64(nk), zero-filled beyond nk)", ss, a->ldk[ss], a->ldvt[ss]);
        p.k[s] = (const uint8_t*)a->k8[ss]; p.ldk[s] = a->ldk[ss]; p.vt[s] = (const uint8_t*)a->vt8[ss]; p.ldvt[s] = a->ldvt[ss];
        p.nk[s] = a->nk[ss]; p.krows[s] = krows; p.seg_b0[s] = a->seg_b0[ss];
    }
    auto rep = [](int e) { const int b = (127 + e) & 0xff; return b | (b << 8) | (b << 16) | (b << 24); };
    p.sc_qk = rep(a->qk_scale_exp);
    p.sc_v = rep(a->v_scale_exp);
    p.nqb = (a->Nq + 127) / 128;
    const dim3 grid(a->B * a->heads * p.nqb), block(256);
    if (a->out_dtype == IDMVTON_BF16) hipLaunchKernelGGL((attn_f8_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_f8_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, p);
    CHECK_LAUNCH("attn_f8");
    return IDMVTON_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// idmvton_quant_f8: 16-bit -> e4m3 with a power-of-two scale (saturating at +-448).
//   mode 0: dst[r][c] = e4m3(src[r*lds + c] * scale), c < cols (cols % 16 == 0)                                (Q, K)
//   mode 1: src = V^T in the 16-bit kernels' key order ([rows][cols positions], bits 2,3 of the key swapped per 16), dst = V^T in the
//           fp8 kernel's slot order per 64-key tile (see the file header), zero-filled from cols to ldd        (V^T)
// One thread per 16 destination bytes.
template <typename T>
__global__ __launch_bounds__(256) void quant_f8_kernel(const idmvton_quant_f8_args a) {
    const int chunks = a.ldd >> 4;                                 // 16-byte chunks per destination row (mode 0: only cols/16 are written)
    const int nch = a.mode == 0 ? a.cols >> 4 : chunks;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.rows * nch) return;
    const int r = (int)(idx / nch), c = (int)(idx - (long)r * nch);
    const T* src = (const T*)a.src + (size_t)r * a.lds;
    float v[16];
    if (a.mode == 0) {
        typedef typename VT<T>::v8 v8;
        const v8 x0 = *(const v8*)(src + c * 16), x1 = *(const v8*)(src + c * 16 + 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (float)x0[j]; v[8 + j] = (float)x1[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pos = c * 16 + j;                            // destination position: 64 t + 32 u + 16 kb + 4 g + jj
            const int t = pos >> 6, uu = (pos >> 5) & 1, kb = (pos >> 4) & 1, g = (pos >> 2) & 3, jj = pos & 3;
            const int key = 64 * t + 32 * kb + 8 * g + 4 * uu + jj;
            const int sp = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);      // where the 16-bit V^T keeps that key
            v[j] = key < a.cols ? (float)src[sp] : 0.f;
        }
    }
    int4 o;
    int w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        w[q] = pack4_fp8(clamp448(v[4 * q] * a.scale), clamp448(v[4 * q + 1] * a.scale), clamp448(v[4 * q + 2] * a.scale), clamp448(v[4 * q + 3] * a.scale));
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *(int4*)((uint8_t*)a.dst + (size_t)r * a.ldd + c * 16) = o;
}

extern "C" int idmvton_quant_f8(const idmvton_quant_f8_args* a, void* stream) {
    CHECK_ARG(a && a->src && a->dst, IDMVTON_E_ARG, "quant_f8: null pointer");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "quant_f8: dtype %d", a->dtype);
    CHECK_ARG(a->mode == 0 || a->mode == 1, IDMVTON_E_ARG, "quant_f8: mode %d", a->mode);
    CHECK_ARG(a->rows > 0 && a->cols > 0 && a->cols % 16 == 0 && a->lds >= a->cols && a->lds % 8 == 0 && a->ldd % 16 == 0 &&
              a->ldd >= (a->mode == 0 ? a->cols : ((a->cols + 63) & ~63)) && (a->mode == 0 || a->ldd % 64 == 0) &&
              ((uintptr_t)a->src & 15) == 0 && ((uintptr_t)a->dst & 15) == 0 && a->scale > 0.f,
              IDMVTON_E_SHAPE, "quant_f8: rows=%d cols=%d lds=%d ldd=%d", a->rows, a->cols, a->lds, a->ldd);
    const long n = (long)a->rows * (a->mode == 0 ? a->cols >> 4 : a->ldd >> 4);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((quant_f8_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((quant_f8_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("quant_f8");
    return IDMVTON_OK;
}
