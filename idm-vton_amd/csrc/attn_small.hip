// attn_small.hip -- attention for the ONE-OFF conditioning encoders: the CLIP text towers (77 tokens,
// causal, head_dim 64) and the CLIP-H vision tower (257 tokens, head_dim 80) of src/tryon_pipeline.py:460-507,511-743.
// These run once per pipeline call on a few hundred tokens (~0.01 TFLOP of attention in total), need a causal mask and a head
// dimension the d=64 flash kernel does not have, and are latency- not throughput-bound: one wave per query row, fp32 arithmetic,
// scores in LDS, no MFMA.  (The denoising loop's attention is csrc/attention.hip.)
#include "common.cuh"

#define AS_MAXK 1024
#define AS_MAXD 128

template <typename T>
__global__ __launch_bounds__(256) void attn_small_kernel(const idmvton_attn_small_args a) {
    __shared__ float sq[4][AS_MAXD];
    __shared__ float sp[4][AS_MAXK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    const int b = blockIdx.y / a.heads, h = blockIdx.y - b * a.heads;
    if (row >= a.Lq) return;                             // whole wave; no block barrier below
    const T* q = (const T*)a.q + ((size_t)b * a.Lq + row) * a.ldq + h * a.d;
    const T* k = (const T*)a.k + (size_t)b * a.Lk * a.ldk + h * a.d;
    const T* v = (const T*)a.v + (size_t)b * a.Lk * a.ldv + h * a.d;
    for (int c = lane; c < a.d; c += 64) sq[wave][c] = (float)q[c] * a.scale;
    __builtin_amdgcn_wave_barrier();
    const int nk = a.causal ? min(a.Lk, row + 1 + (a.Lk - a.Lq)) : a.Lk;      // keys this row may see
    float mx = -3.0e38f;
    for (int j = lane; j < nk; j += 64) {
        const T* kr = k + (size_t)j * a.ldk;
        float s = 0.f;
        for (int c = 0; c < a.d; c += 2) s += sq[wave][c] * (float)kr[c] + sq[wave][c + 1] * (float)kr[c + 1];
        sp[wave][j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) {
        const float p = __expf(sp[wave][j] - mx);
        sp[wave][j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / sum;
    T* o = (T*)a.out + ((size_t)b * a.Lq + row) * a.ldo + h * a.d;
    for (int c = lane; c < a.d; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < nk; ++j) acc += sp[wave][j] * (float)v[(size_t)j * a.ldv + c];
        o[c] = (T)(acc * inv);
    }
}

extern "C" int idmvton_attn_small(const idmvton_attn_small_args* a, void* stream) {
    CHECK_ARG(a != nullptr, IDMVTON_E_ARG, "attn_small: null args");
    CHECK_ARG(a->dtype == IDMVTON_F16 || a->dtype == IDMVTON_BF16, IDMVTON_E_DTYPE, "attn_small: dtype %d", a->dtype);
    CHECK_ARG(a->B > 0 && a->heads > 0 && a->Lq > 0 && a->Lk > 0 && a->Lk <= AS_MAXK && a->d >= 2 && a->d <= AS_MAXD && a->d % 2 == 0,
              IDMVTON_E_SHAPE, "attn_small: B=%d heads=%d Lq=%d Lk=%d (<= %d) d=%d (even, <= %d)", a->B, a->heads, a->Lq, a->Lk, AS_MAXK, a->d, AS_MAXD);
    CHECK_ARG(a->q && a->k && a->v && a->out, IDMVTON_E_ARG, "attn_small: null pointer");
    CHECK_ARG(a->ldq >= a->heads * a->d && a->ldk >= a->heads * a->d && a->ldv >= a->heads * a->d && a->ldo >= a->heads * a->d,
              IDMVTON_E_SHAPE, "attn_small: leading dimensions < heads*d");
    CHECK_ARG(!a->causal || a->Lk >= a->Lq, IDMVTON_E_ARG, "attn_small: causal needs Lk >= Lq");
    const dim3 grid((a->Lq + 3) / 4, a->B * a->heads), block(256);
    if (a->dtype == IDMVTON_BF16) hipLaunchKernelGGL((attn_small_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((attn_small_kernel<f16_t>), grid, block, 0, (hipStream_t)stream, *a);
    CHECK_LAUNCH("attn_small");
    return IDMVTON_OK;
}
