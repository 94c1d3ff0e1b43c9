// bw_probe.hip -- per-CU operand-delivery microbenchmark (NOT part of the product library): how many bytes per clock can one
// CU pull from an L2-resident buffer (a) by LDS-DMA (`buffer_load ... lds`, what the GEMM / attention kernels use) and
// (b) by plain 16-byte vector loads into VGPRs, for 4 / 8 / 16 waves per CU and different numbers of loads in flight.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC bw_probe.hip -o bw_probe.so ; run: tools/gpu_bw_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Each workgroup streams its own `region` bytes (1 KiB per wave-instruction), `passes` times.
template <int MODE, int DEPTH>
__global__ void bw_kernel(const char* base, uint32_t region, int passes, uint32_t* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[65536];
    const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* mine = base + (size_t)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, region, 0x00020000);
    const uint32_t pieces = region >> 10;                 // 1 KiB pieces
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        for (uint32_t i = wave; i < pieces; i += nw * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint32_t piece = i + d * nw;
                const uint32_t off = (piece < pieces ? piece : i) * 1024u + lane * 16u;
                if (MODE == 0) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + ((wave * DEPTH + d) & 63) * 1024), 16, off, 0, 0, 0);
                } else {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    acc ^= v;
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (MODE == 1 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

extern "C" int bw_probe(int mode, int depth, int waves, int blocks, const void* base, uint32_t region, int passes, void* sink,
                        void* stream) {
    hipStream_t st = (hipStream_t)stream;
    dim3 g(blocks), b(waves * 64);
#define L(M, D) hipLaunchKernelGGL((bw_kernel<M, D>), g, b, 0, st, (const char*)base, region, passes, (uint32_t*)sink)
    if (mode == 0) { if (depth == 1) L(0, 1); else if (depth == 2) L(0, 2); else if (depth == 4) L(0, 4); else if (depth == 8) L(0, 8); else return -1; }
    else { if (depth == 1) L(1, 1); else if (depth == 2) L(1, 2); else if (depth == 4) L(1, 4); else if (depth == 8) L(1, 8); else return -1; }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
