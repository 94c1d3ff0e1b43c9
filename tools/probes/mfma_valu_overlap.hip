// Probe: do MFMA and VALU / transcendental work of the two waves of one SIMD overlap, and does it depend on where the MFMA accumulator lives
// (arch VGPRs, "-amdgpu-mfma-vgpr-form", vs AGPRs)?  One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run `nm` MFMAs per iteration,
// waves 4-7 (their SIMD partners) run `nv` VALU ops per iteration; either side can be switched off.  Also the single-wave forms (one wave issuing
// both, interleaved 1 MFMA : k VALU).  Prints cycles per iteration (s_memtime) for every combination.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o .ab_r06/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// mode bits: 1 = MFMA waves active, 2 = VALU waves active; accA: accumulators in AGPRs; op: 0 v_exp_f32, 1 v_fma_f32, 2 v_max3, 3 v_cvt_pk + v_mov mix
template <int ACCA, int OP>
__global__ __launch_bounds__(512, 2) void probe(int mode, int iters, float* out, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = 0.001f * (lane + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ACCA) { MFMA_A(c0, a, b); MFMA_A(c1, a, b); MFMA_A(c2, a, b); MFMA_A(c3, a, b); }
                    else { MFMA_V(c0, a, b); MFMA_V(c1, a, b); MFMA_V(c2, a, b); MFMA_V(c3, a, b); }
                }
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                        else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[r]));
                        else if (OP == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(x[(r + 1) & 15]), "v"(x[(r + 2) & 15]));
                        else asm volatile("v_mov_b32 %0, %1" : "=v"(x[r]) : "v"(x[(r + 1) & 15]));
                    }
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + x[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

// single wave per SIMD (256 threads), 16 MFMAs interleaved with k VALU ops each
template <int ACCA, int OP, int K>
__global__ __launch_bounds__(256, 1) void probe1(int iters, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = 0.001f * (lane + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x16& c = (q & 3) == 0 ? c0 : (q & 3) == 1 ? c1 : (q & 3) == 2 ? c2 : c3;
            if (ACCA) MFMA_A(c, a, b); else MFMA_V(c, a, b);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int r = (q * K + k) & 15;
                if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[r]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + x[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// two waves per SIMD, BOTH running the interleaved stream (16 MFMAs, K fillers each); OP2: 0 exp, 1 fma, 2 = K/2 exp + K/2 fma
template <int OP, int K>
__global__ __launch_bounds__(512, 2) void probe2(int iters, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = 0.001f * (lane + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x16& c = (q & 3) == 0 ? c0 : (q & 3) == 1 ? c1 : (q & 3) == 2 ? c2 : c3;
            MFMA_V(c, a, b);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int r = (q * K + k) & 15;
                if (OP == 0 || (OP == 2 && (k & 1))) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[r]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + x[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
// two waves per SIMD, both VALU only: 64 ops per iteration each
template <int OPA, int OPB>
__global__ __launch_bounds__(512, 2) void probe3(int iters, float* out, unsigned long long* cyc) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float x[16];
    for (int r = 0; r < 16; ++r) x[r] = 0.001f * (lane + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 64; ++q) { if (OPA == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[q & 15])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[q & 15])); }
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 64; ++q) { if (OPB == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[q & 15])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[q & 15])); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += x[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    unsigned long long h[256 * 16];
    const int iters = 200;
    auto run2 = [&](const char* name, auto kern) {
        for (int mode = 1; mode <= 3; ++mode) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0, v = 0, e = 0;
            for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < 8; ++w) { (w < 4 ? m : v) += h[(bI * 8 + w) * 2]; e += h[(bI * 8 + w) * 2 + 1]; }
            printf("%-28s mode %d (%s): mfma waves %7.1f cyc/iter (16 MFMA)  valu waves %7.1f cyc/iter (64 ops)  all-done %7.1f\n", name, mode,
                   mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", m / 1024 / iters, v / 1024 / iters, e / 2048 / iters);
        }
    };
    run2("acc VGPR, partner v_exp", probe<0, 0>);
    run2("acc AGPR, partner v_exp", probe<1, 0>);
    run2("acc VGPR, partner v_fma", probe<0, 1>);
    run2("acc AGPR, partner v_fma", probe<1, 1>);
    run2("acc VGPR, partner v_max3", probe<0, 2>);
    run2("acc VGPR, partner v_mov", probe<0, 3>);
    auto run1 = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, out, cyc);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 256 * 4 * 8, hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < 1024; ++i) m += h[i];
        printf("%-44s %7.1f cyc per 16 MFMA = %5.1f per MFMA\n", name, m / 1024 / iters, m / 1024 / iters / 16);
    };
    run1("one wave/SIMD, acc VGPR, 0 fillers", probe1<0, 1, 0>);
    run1("one wave/SIMD, acc VGPR, 2 v_exp per MFMA", probe1<0, 0, 2>);
    run1("one wave/SIMD, acc AGPR, 2 v_exp per MFMA", probe1<1, 0, 2>);
    run1("one wave/SIMD, acc VGPR, 4 v_exp per MFMA", probe1<0, 0, 4>);
    run1("one wave/SIMD, acc AGPR, 4 v_exp per MFMA", probe1<1, 0, 4>);
    run1("one wave/SIMD, acc VGPR, 4 v_fma per MFMA", probe1<0, 1, 4>);
    run1("one wave/SIMD, acc AGPR, 4 v_fma per MFMA", probe1<1, 1, 4>);
    run1("one wave/SIMD, acc VGPR, 8 v_fma per MFMA", probe1<0, 1, 8>);
    run1("one wave/SIMD, acc AGPR, 8 v_fma per MFMA", probe1<1, 1, 8>);
    auto run3 = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, out, cyc);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[bI * 8 + w];
        printf("%-52s waves 0-3 %7.1f  waves 4-7 %7.1f cyc/iter\n", name, m / 1024 / iters, v / 1024 / iters);
    };
    run3("two waves/SIMD: 64 v_exp || 64 v_exp", probe3<0, 0>);
    run3("two waves/SIMD: 64 v_fma || 64 v_fma", probe3<1, 1>);
    run3("two waves/SIMD: 64 v_exp || 64 v_fma", probe3<0, 1>);
    run3("two waves/SIMD, each 16 x (MFMA + 2 v_exp)", probe2<0, 2>);
    run3("two waves/SIMD, each 16 x (MFMA + 2 v_fma)", probe2<1, 2>);
    run3("two waves/SIMD, each 16 x (MFMA + 4 v_fma)", probe2<1, 4>);
    run3("two waves/SIMD, each 16 x (MFMA + 2 exp + 2 fma)", probe2<2, 4>);
    run3("two waves/SIMD, each 16 x (MFMA + 3 exp + 3 fma)", probe2<2, 6>);
    run3("two waves/SIMD, each 16 x (MFMA + 4 exp + 4 fma)", probe2<2, 8>);
    run3("two waves/SIMD, each 16 x (MFMA + 8 v_fma)", probe2<1, 8>);
    return 0;
}
