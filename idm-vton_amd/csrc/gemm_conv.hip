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
// Pipeline: 2 LDS buffers, one barrier per 64-deep K tile: wait(tile t) -> barrier -> issue DMA(tile t+1) -> MFMA(tile t).
#include "common.cuh"

struct GemmParams {
    const void* w; uint32_t w_bytes; int N; int Ktot;
    int nseg; idmvton_seg seg[IDMVTON_MAX_SEG];
    int M, Ho, Wo, Hi, Wi, stride, ups;
    void* out; int ldo;
    const void* bias; const void* rowbias; int rowbias_ld; int rows_per_group;
    const void* res; int ldr;
    int mode;
    void* vt; int vt_n0; int vt_tokens;
    int tiles_m, tiles_n;
};

// TR = this n-tile is written transposed (V^T epilogue): the MFMA operands are swapped so the accumulator is D[m][n].
template <typename T, int BN, int BM, bool TR>
__device__ __forceinline__ void gemm_body(const GemmParams& p, char* smem, const int m0, const int n0) {
    typedef typename VT<T>::v8 v8;
    typedef typename VT<T>::v4 v4;
    constexpr int NI = BN / 64, MI = BM / 64;          // 32x32 MFMA tiles per wave along n / m
    constexpr int WBYTES = BN * 128, XBYTES = BM * 128; // one LDS buffer of each operand
    constexpr int WI = BN / 32, XI = BM / 32;           // DMA instructions per wave per tile
    char* sW = smem;
    char* sX = smem + 2 * WBYTES;

    const int lane = threadIdx.x & 63;
    const int wave = uniform(threadIdx.x >> 6);
    const int wn = wave >> 1, wm = wave & 1;
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
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int R = (wave * XI + i) * 8 + lrow;
        const int m = m0 + R;
        x_c8[i] = (lslot ^ ((R >> 1) & 7)) * 8;
        if (m < p.M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            x_pix[i] = b * p.Hi * p.Wi; x_oy[i] = oy * p.stride; x_ox[i] = ox * p.stride;
        } else {
            x_pix[i] = 0; x_oy[i] = -(1 << 28); x_ox[i] = 0;   // fails every bounds test -> zeros
        }
    }
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
    const int hin = p.ups ? 2 * p.Hi : p.Hi, win = p.ups ? 2 * p.Wi : p.Wi;

    int si = 0, kseg = 0;                                // K-segment cursor of the NEXT tile to issue
    auto issue = [&](int t, int buf) {
        const idmvton_seg sg = p.seg[si];
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(sg.ptr, sg.bytes);
        char* dW = sW + buf * WBYTES + wave * (WI * 1024);
        char* dX = sX + buf * XBYTES + wave * (XI * 1024);
#pragma unroll
        for (int i = 0; i < WI; ++i) dma16(rs_w, dW + i * 1024, w_off[i] + (uint32_t)t * 128u);
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
    };

    // ---- fragment read addresses (row*128 and the row's swizzle key) ----
    int a_row[NI], a_swz[NI], b_row[MI], b_swz[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { const int r = wn * (BN / 2) + ni * 32 + l31; a_row[ni] = r * 128; a_swz[ni] = (r >> 1) & 7; }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) { const int r = wm * (BM / 2) + mi * 32 + l31; b_row[mi] = r * 128; b_swz[mi] = (r >> 1) & 7; }

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    const int nt = p.Ktot >> 6;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const char* bW = sW + (t & 1) * WBYTES;
        const char* bX = sX + (t & 1) * XBYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v8 a[NI], b[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) a[ni] = *(const v8*)(bW + a_row[ni] + (((2 * s + u) ^ a_swz[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) b[mi] = *(const v8*)(bX + b_row[mi] + (((2 * s + u) ^ b_swz[mi]) << 4));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[ni][mi] = TR ? VT<T>::mfma(b[mi], a[ni], acc[ni][mi]) : VT<T>::mfma(a[ni], b[mi], acc[ni][mi]);
        }
    }

    // ---- epilogue ----
    const T* bias = (const T*)p.bias;
    if constexpr (TR) {
        // acc[ni][mi] = D[m][n]: column n = l31, rows m = 8g + 4u + j.  vt[(b*Cv + n - vt_n0)*tokens + tok..tok+3]
        T* vt = (T*)p.vt;
        const int Cv = p.N - p.vt_n0;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + wn * (BN / 2) + ni * 32 + l31;
            if (n >= p.N) continue;
            const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + wm * (BM / 2) + mi * 32 + 8 * g + 4 * u;
                    if (m >= p.M) continue;
                    const int b = m / p.vt_tokens, tok = m - b * p.vt_tokens;
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
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * (BM / 2) + mi * 32 + l31;
        if (m >= p.M) continue;
        const T* rb = rowbias ? rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
        if (p.mode == IDMVTON_EPI_GEGLU) {
            if constexpr (NI == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nh = n0 + wn * 64 + 8 * g + 4 * u;          // h rows; gate rows are nh + 32
                    if (nh + 32 >= p.N) continue;
                    const int jo = ((n0 + wn * 64) >> 1) + 8 * g + 4 * u;
                    v4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float h = acc[0][mi][4 * g + j], gt = acc[1][mi][4 * g + j];
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
                const int n = n0 + wn * (BN / 2) + ni * 32 + 8 * g + 4 * u;
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
                if (p.mode == IDMVTON_EPI_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
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

template <typename T, int BN, int BM>
__global__ __launch_bounds__(256, (BN + BM >= 256 ? 2 : 3)) void gemm_conv_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * (BN + BM) * 128];
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    if (p.vt != nullptr && n0 >= p.vt_n0) gemm_body<T, BN, BM, true>(p, smem, m0, n0);   // block-uniform
    else gemm_body<T, BN, BM, false>(p, smem, m0, n0);
}

template <typename T>
static int launch_gemm(const GemmParams& p0, int bn, int bm, hipStream_t st) {
    GemmParams p = p0;
    p.tiles_n = (p.N + bn - 1) / bn;
    p.tiles_m = (p.M + bm - 1) / bm;
    const dim3 grid(p.tiles_n * p.tiles_m), block(256);
    if (bn == 128 && bm == 128) hipLaunchKernelGGL((gemm_conv_kernel<T, 128, 128>), grid, block, 0, st, p);
    else if (bn == 128 && bm == 64) hipLaunchKernelGGL((gemm_conv_kernel<T, 128, 64>), grid, block, 0, st, p);
    else if (bn == 64 && bm == 64) hipLaunchKernelGGL((gemm_conv_kernel<T, 64, 64>), grid, block, 0, st, p);
    else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unsupported tile %dx%d", bn, bm);
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
    CHECK_ARG(a->w && ((uintptr_t)a->w & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: weight pointer");
    const uint64_t wbytes = (uint64_t)a->N * a->Ktot * 2;
    CHECK_ARG(wbytes + (uint64_t)128 * a->Ktot * 2 < 0xFFFFFFFFull, IDMVTON_E_SHAPE, "gemm_conv: weight too large");
    const bool geglu = a->mode == IDMVTON_EPI_GEGLU;
    CHECK_ARG(a->mode == IDMVTON_EPI_NONE || a->mode == IDMVTON_EPI_GELU || geglu, IDMVTON_E_ARG, "gemm_conv: mode %d", a->mode);
    if (geglu) CHECK_ARG(a->N % 64 == 0 && !a->res && !a->rowbias && !a->vt, IDMVTON_E_ARG, "gemm_conv: GEGLU needs N%%64==0, no res/rowbias/vt");
    CHECK_ARG(a->out || (a->vt && a->vt_n0 == 0), IDMVTON_E_ARG, "gemm_conv: null out");
    if (a->out) CHECK_ARG(a->ldo % 4 == 0 && ((uintptr_t)a->out & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: out alignment");
    if (a->res) CHECK_ARG(a->ldr % 4 == 0 && ((uintptr_t)a->res & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: res alignment");
    if (a->bias) CHECK_ARG(((uintptr_t)a->bias & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: bias alignment");
    if (a->rowbias) CHECK_ARG(a->rowbias_ld % 4 == 0 && a->rows_per_group > 0 && ((uintptr_t)a->rowbias & 7) == 0,
                              IDMVTON_E_ALIGN, "gemm_conv: rowbias");
    if (a->vt) CHECK_ARG(a->vt_tokens > 0 && a->vt_tokens % 4 == 0 && a->M % a->vt_tokens == 0 && a->vt_n0 % 64 == 0 &&
                         a->vt_n0 >= 0 && a->vt_n0 < a->N && ((uintptr_t)a->vt & 7) == 0,
                         IDMVTON_E_ARG, "gemm_conv: vt_tokens=%d vt_n0=%d", a->vt_tokens, a->vt_n0);

    GemmParams p;
    p.w = a->w; p.w_bytes = (uint32_t)wbytes; p.N = a->N; p.Ktot = a->Ktot;
    p.nseg = a->nseg;
    for (int s = 0; s < IDMVTON_MAX_SEG; ++s) p.seg[s] = a->seg[s < a->nseg ? s : a->nseg - 1];
    p.M = a->M; p.Ho = a->Ho; p.Wo = a->Wo; p.Hi = a->Hi; p.Wi = a->Wi; p.stride = a->stride; p.ups = a->ups ? 1 : 0;
    p.out = a->out; p.ldo = a->ldo; p.bias = a->bias; p.rowbias = a->rowbias; p.rowbias_ld = a->rowbias_ld;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1; p.res = a->res; p.ldr = a->ldr; p.mode = a->mode;
    p.vt = a->vt; p.vt_n0 = a->vt_n0; p.vt_tokens = a->vt_tokens > 0 ? a->vt_tokens : 4;
    p.tiles_m = p.tiles_n = 0;

    // Tile choice: largest tile that still gives >= 2 workgroups per CU (256 CUs); GEGLU needs 64-row wave tiles (BN=128).
    int bn = 128, bm = 128;
    if (a->tile_hint) { bn = a->tile_hint >> 16; bm = a->tile_hint & 0xffff; }
    else {
        auto tiles = [&](int n_, int m_) { return (long)((a->N + n_ - 1) / n_) * ((a->M + m_ - 1) / m_); };
        if (tiles(128, 128) >= 512) { bn = 128; bm = 128; }
        else if (tiles(128, 64) >= 384 || geglu) { bn = 128; bm = 64; }
        else { bn = 64; bm = 64; }
    }
    if (geglu) CHECK_ARG(bn == 128, IDMVTON_E_ARG, "gemm_conv: GEGLU needs BN=128");
    if (a->vt) CHECK_ARG(a->vt_n0 % bn == 0, IDMVTON_E_ARG, "gemm_conv: vt_n0 %% BN != 0");
    hipStream_t st = (hipStream_t)stream;
    return a->dtype == IDMVTON_BF16 ? launch_gemm<bf16_t>(p, bn, bm, st) : launch_gemm<f16_t>(p, bn, bm, st);
}
