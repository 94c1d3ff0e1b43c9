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
// Pipeline, two forms (template parameter ST = LDS stages):
//   ST == 2 : wait(tile t, vmcnt 0) -> __syncthreads -> issue DMA(tile t+1) -> MFMA(tile t)      (latency hidden by 2-3 blocks/CU)
//   ST >= 3 : ring of ST buffers, DMA runs ST-1 tiles ahead and stays in flight ACROSS the barrier: counted
//             `s_waitcnt vmcnt((ST-2)*loads_per_tile)` (only tile t must have landed) -> raw s_barrier -> issue
//             DMA(tile t+ST-1) into the buffer tile t-1 was read from (every wave is past compute(t-1) once it has passed
//             this barrier) -> MFMA(tile t).  Used with 8-wave 256x128 tiles (144 KiB LDS, 48 B/clk/CU of operand traffic
//             instead of 64 for 128x128) and with small tiles where 1-2 blocks per CU cannot hide the HBM/L2 latency.
// LIN = plain Linear (one K segment, no spatial gather): the activation operand's DMA offsets are precomputed like the
// weight's, so issuing a tile costs one add per DMA instead of the ~12 VALU of the conv gather.
// This file is the C entry point and the tile dispatch; the main loop lives in gemm_body.cuh and the tile configurations are instantiated by
// gemm_tiles_*.hip (one family per translation unit, so the library builds in parallel).
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_body.cuh"

// gemm_lin.hip: the hand-scheduled Linear main loop (variant 5: 256x256 as 8 waves x 128x64, 256x192 as 8 waves x 64x96)
int launch_gemm_lin(const GemmParams& p, bool bf16, int bm, int form, int grid_cap, hipStream_t st);

// ---- raster group height ------------------------------------------------------------------------------------------------------------
// Workgroup ids are remapped so that XCD x runs the x-th eighth of the tile order (xcd_remap), and the order is a grouped raster: groups of GM
// m-tiles, inside a group n-tile major.  Each XCD has a private 4 MiB L2, so it fetches (from HBM / the Infinity Cache) every A row-panel
// (BM x K) and every W row-panel (BN x K) its tiles touch: the launch costs sum_x (|M_x| BM + |N_x| BN) K 2 bytes of operand traffic whatever the
// algorithmic size is.  The product runs GM = 1024 / BM (return value 0).  Round 6 measured the alternative: walking the
// order for every GM and taking the one with the smallest sum, subject to a group's A panels fitting a share of the L2 (IDMVTON_GM_MODEL=1,
// cap IDMVTON_GM_CAP_KB, default 2048).  It does what it says to the counters on the single-round launches (3072 x 1280 x 1280: 46 -> 37 MB,
// 9216 x 1280 x 1280: 90 -> 74 MB, ff2 171 -> 152 MB; with a 6 MiB cap the many-round launches thrash instead: 9216 x 10240 477 -> 685 MB) and
// the pipeline gets SLOWER with every cap tried (38.7 -> 39.3 / 39.4 / 39.7 ms per denoising step at 4 / 1 / 2 MiB, same box, interleaved:
// profiles/r06_gemm_traffic_table.txt): these launches are bound by the per-CU LDS fill path, not by L2 misses, and the tile table was
// tuned on the 1024-row order.  So the model stays a measurement switch.  IDMVTON_GM=<n> forces a value.
#include <map>
#include <mutex>
#include <tuple>
int idmvton_choose_gm(int tiles_m, int tiles_n, int bm, int bn, int K) {
    static const char* env = getenv("IDMVTON_GM");
    if (env) return atoi(env);
    static const bool model = getenv("IDMVTON_GM_MODEL") && atoi(getenv("IDMVTON_GM_MODEL")) != 0;
    if (!model) return 0;
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, int> cache;
    const auto key = std::make_tuple(tiles_m, tiles_n, bm, bn, K);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int T = tiles_m * tiles_n, q = T / 8, r = T % 8;
    static const long cap = getenv("IDMVTON_GM_CAP_KB") ? atol(getenv("IDMVTON_GM_CAP_KB")) << 10 : (2l << 20);
    long best = -1; int best_gm = 0;
    for (int gm = 1; gm <= tiles_m; ++gm) {
        if (gm > 1 && (long)gm * bm * K * 2 > cap) break;
        long cost = 0;
        for (int x = 0; x < 8; ++x) {
            const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, n = x < r ? q + 1 : q;
            int m_lo = tiles_m, m_hi = -1;               // an XCD's run is contiguous in the order: its m-tiles and n-tiles are counted exactly
            std::vector<char> ms(tiles_m, 0), ns(tiles_n, 0);
            int cm = 0, cn = 0;
            for (int wg = lo; wg < lo + n; ++wg) {
                const int width = gm * tiles_n, grp = wg / width, rem = wg - grp * width, first = grp * gm;
                const int gsz = tiles_m - first < gm ? tiles_m - first : gm;
                const int tn = rem / gsz, tm = first + (rem - tn * gsz);
                if (!ms[tm]) { ms[tm] = 1; ++cm; }
                if (!ns[tn]) { ns[tn] = 1; ++cn; }
            }
            (void)m_lo; (void)m_hi;
            cost += (long)cm * bm + (long)cn * bn;
        }
        if (best < 0 || cost < best) { best = cost; best_gm = gm; }
    }
    cache[key] = best_gm;
    return best_gm;
}

static int launch_gemm(const GemmParams& p0, bool bf16, int variant, int bn, int bm, bool lin, hipStream_t st) {
    GemmParams p = p0;
    int form = 0;
    int grid_cap = 0;                                        // tests only (bit 14 of the BM field): 5 persistent workgroups, so small shapes walk several tiles each
    if (variant == 5) { form = bm & 15; grid_cap = (bm & 0x4000) ? 5 : 0; bm &= ~(15 | 0x4000); }   // placement form: low nibble of the BM field
    p.tiles_n = (p.N + bn - 1) / bn;
    p.tiles_m = (p.M + bm - 1) / bm;
    int rc = 1;                                          // 1 = the family has no such tile
    if (p.mode == IDMVTON_EPI_XATTN) {                   // tiles whose waves own 64 columns: 128x64, 128x128 (2x2 waves), 128x256 (2x4)
        if (bm != 64 && p.xa.tokens % 64 != 0) {         // a wave's rows must lie in one batch element: 64-row waves need tokens % 64 == 0
            bm = 64;
            p.tiles_m = (p.M + bm - 1) / bm;
        }
        if (gemm_tiles_xattn(p, bf16, bn, bm, variant == 6 && bn == 128 && bm == 128, st))
            return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN runs on the 128x64, 128x128 and 128x256 tiles (got %dx%d)", bn, bm);
        CHECK_LAUNCH("gemm_conv");
        return IDMVTON_OK;
    }
    if (variant == 0) rc = gemm_tiles_v0(p, bf16, bn, bm, lin, st);
    else if (variant == 1) rc = gemm_tiles_v1(p, bf16, bn, bm, lin, st);
    else if (variant == 2) rc = gemm_tiles_v2(p, bf16, bn, bm, lin, st);
    else if (variant == 5) {                             // hand-scheduled Linear loop; anything it does not cover runs on the 8-wave ring tile
        if (!(bn == 256 && (bm == 256 || bm == 192))) return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: variant 5 is the 256x256 / 256x192 tile");
        if (lin) rc = launch_gemm_lin(p, bf16, bm, form, grid_cap, st);
        else if (bm == 256) rc = gemm_tiles_v1(p, bf16, 256, 256, lin, st);
        else {                                           // 256x192 exists only hand-scheduled: the same launch on the 128x256 ring tile
            p.tiles_n = (p.N + 127) / 128; p.tiles_m = (p.M + 255) / 256;
            rc = gemm_tiles_v1(p, bf16, 128, 256, lin, st);
        }
    } else if (variant == 6) {
        // More waves per workgroup on the tiles that run ONE workgroup per CU (gemm_tiles_w8.hip): 128x128 as 2 x 4 waves of 64x32 (two waves per SIMD
        // on one shared k-tile: somebody to cover the fragment-read / DMA / barrier waits of the 4-wave tile), 320x192 as 12 waves of 160x32 (N = 320 in
        // ONE weight tile -- the 128-column tiles compute 384 columns for it -- and M = 49152 gives exactly 256 tiles), 256x192 as 12 waves of 64x64,
        // 256x256 / 128x256 as 16 waves.  The 320-column tile takes no V^T part and only the 16-byte epilogue, no GEGLU pairing (gemm_common.cuh):
        // launches it cannot take run on the plain ring tile.
        // (Measured with them and removed, profiles/r05_tune_report_v1_new_tiles.json: 128x128 with INTRA-WORKGROUP SPLIT-K -- two 4-wave groups
        //  on alternate k-tiles, two LDS rings, partial sums exchanged through LDS -- 6-20 % SLOWER than the 4-wave ring tile on every shape.)
        const int form6 = bm & 15;                       // low nibble of the BM field: form of the 128x128 tile (gemm_tiles_w8.hip)
        bm &= ~15;
        p.tiles_m = (p.M + bm - 1) / bm;
        if (bn == 320 && (p.vt || !p.wide || p.out8 || p.mode == IDMVTON_EPI_GEGLU)) {
            const int fbn = bn == 320 ? 128 : bn, fbm = bn == 320 ? 256 : bm;
            p.tiles_n = (p.N + fbn - 1) / fbn;
            p.tiles_m = (p.M + fbm - 1) / fbm;
            rc = gemm_tiles_v1(p, bf16, fbn, fbm, lin, st);
        } else rc = gemm_tiles_w8(p, bf16, bn, bm, form6, lin, st);
    } else return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: unknown variant %d", variant);
    if (rc) return idmvton_set_error(IDMVTON_E_ARG, "gemm_conv: variant %d has no %dx%d tile", variant, bn, bm);
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
    if (a->ups) CHECK_ARG(a->stride == 1 && a->Ho <= 2 * a->Hi && a->Ho >= 2 * a->Hi - 1 && a->Wo <= 2 * a->Wi && a->Wo >= 2 * a->Wi - 1, IDMVTON_E_SHAPE,
                          "gemm_conv: ups=1 needs stride 1 and an output grid of 2Hi x 2Wi or one short of it (Ho=%d Hi=%d Wo=%d Wi=%d)", a->Ho, a->Hi, a->Wo, a->Wi);
    CHECK_ARG(a->w && ((uintptr_t)a->w & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: weight pointer");
    const uint64_t wbytes = (uint64_t)a->N * a->Ktot * 2;
    CHECK_ARG(wbytes + (uint64_t)128 * a->Ktot * 2 < 0xFFFFFFFFull, IDMVTON_E_SHAPE, "gemm_conv: weight too large");
    const bool geglu = a->mode == IDMVTON_EPI_GEGLU, xattn = a->mode == IDMVTON_EPI_XATTN;
    CHECK_ARG(a->mode == IDMVTON_EPI_NONE || a->mode == IDMVTON_EPI_GELU || a->mode == IDMVTON_EPI_QUICKGELU || geglu || xattn, IDMVTON_E_ARG, "gemm_conv: mode %d", a->mode);
    CHECK_ARG(a->colscale_n >= 0 && a->colscale_n % 4 == 0 && !(geglu && a->colscale_n), IDMVTON_E_ARG, "gemm_conv: colscale_n=%d", a->colscale_n);
    if (geglu) CHECK_ARG(a->N % 64 == 0 && !a->res && !a->rowbias && !a->vt, IDMVTON_E_ARG, "gemm_conv: GEGLU needs N%%64==0, no res/rowbias/vt");
    CHECK_ARG(a->out || (a->vt && a->vt_n0 == 0), IDMVTON_E_ARG, "gemm_conv: null out");
    if (a->out) CHECK_ARG(a->ldo % 4 == 0 && ((uintptr_t)a->out & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: out alignment");
    if (a->res) CHECK_ARG(a->ldr % 4 == 0 && ((uintptr_t)a->res & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: res alignment");
    if (a->bias) CHECK_ARG(((uintptr_t)a->bias & 7) == 0, IDMVTON_E_ALIGN, "gemm_conv: bias alignment");
    if (a->rowbias) CHECK_ARG(a->rowbias_ld % 4 == 0 && a->rows_per_group > 0 && ((uintptr_t)a->rowbias & 7) == 0,
                              IDMVTON_E_ALIGN, "gemm_conv: rowbias");
    if (a->vt && a->vt_perm) CHECK_ARG(((uintptr_t)a->vt & 15) == 0, IDMVTON_E_ALIGN, "gemm_conv: vt pointer not 16-byte aligned");
    if (a->vt && a->vt_perm) CHECK_ARG(a->vt_tokens % 16 == 0, IDMVTON_E_ARG, "gemm_conv: vt_perm needs vt_tokens %% 16 == 0 (got %d)", a->vt_tokens);
    if (a->vt) CHECK_ARG(a->vt_tokens > 0 && a->vt_tokens % 4 == 0 && a->M % a->vt_tokens == 0 && a->vt_n0 % 64 == 0 &&
                         a->vt_n0 >= 0 && a->vt_n0 < a->N && ((uintptr_t)a->vt & 7) == 0,
                         IDMVTON_E_ARG, "gemm_conv: vt_tokens=%d vt_n0=%d", a->vt_tokens, a->vt_n0);

    GemmParams p;
    memset(&p.xa, 0, sizeof(p.xa));
    if (xattn) {
        const idmvton_xattn* x = a->xattn;
        CHECK_ARG(x != nullptr && (x->nseg == 1 || x->nseg == 2), IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN needs `xattn` with 1 or 2 key segments");
        CHECK_ARG(a->N % 64 == 0 && !a->bias && !a->res && !a->rowbias && !a->vt && !a->colscale_n && !a->io_flags &&
                  a->nseg == 1 && a->Ho == 1 && a->Hi == 1 && a->Wo == a->M && a->Wi == a->M && a->out && a->ldo % 8 == 0 && ((uintptr_t)a->out & 15) == 0,
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_EPI_XATTN is a plain Linear with N %% 64 == 0 and nothing else in its epilogue");
        CHECK_ARG(x->tokens > 0 && x->tokens % 32 == 0 && a->M % x->tokens == 0, IDMVTON_E_SHAPE, "gemm_conv: xattn.tokens=%d (rows per batch element, a multiple of 32 dividing M=%d)", x->tokens, a->M);
        for (int s = 0; s < x->nseg; ++s) {
            const int nk_max = s == 0 ? 96 : 32;         // xattn.cuh holds 3 K blocks / 6 V^T steps for segment 0 (text) and 1 / 2 for segment 1 (image prompt)
            CHECK_ARG(x->k[s] && x->vt[s] && x->nk[s] > 0 && x->nk[s] <= nk_max && x->k_rows[s] >= ((x->nk[s] + 31) & ~31) && x->ldk[s] >= a->N && x->ldk[s] % 8 == 0 &&
                      x->ldvt[s] >= ((x->nk[s] + 15) & ~15) && x->ldvt[s] % 8 == 0 && (((uintptr_t)x->k[s] | (uintptr_t)x->vt[s]) & 15) == 0, IDMVTON_E_SHAPE,
                      "gemm_conv: xattn segment %d: nk=%d (<= %d) k_rows=%d (>= round32(nk)) ldk=%d ldvt=%d (>= round16(nk))", s, x->nk[s], nk_max, x->k_rows[s], x->ldk[s], x->ldvt[s]);
            p.xa.k[s] = x->k[s]; p.xa.vt[s] = x->vt[s]; p.xa.ldk[s] = x->ldk[s]; p.xa.ldvt[s] = x->ldvt[s]; p.xa.nk[s] = x->nk[s]; p.xa.krows[s] = x->k_rows[s];
        }
        p.xa.nseg = x->nseg; p.xa.tokens = x->tokens; p.xa.vchan = a->N; p.xa.ip_scale = x->ip_scale;
    } else CHECK_ARG(a->xattn == nullptr, IDMVTON_E_ARG, "gemm_conv: `xattn` without mode IDMVTON_EPI_XATTN");
    p.w = a->w; p.w_bytes = (uint32_t)wbytes; p.N = a->N; p.Ktot = a->Ktot;
    p.nseg = a->nseg;
    for (int s = 0; s < IDMVTON_MAX_SEG; ++s) p.seg[s] = a->seg[s < a->nseg ? s : a->nseg - 1];
    p.M = a->M; p.Ho = a->Ho; p.Wo = a->Wo; p.Hi = a->Hi; p.Wi = a->Wi; p.stride = a->stride; p.ups = a->ups ? 1 : 0;
    p.out = a->out; p.ldo = a->ldo; p.bias = a->bias; p.rowbias = a->rowbias; p.rowbias_ld = a->rowbias_ld;
    p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1; p.res = a->res; p.ldr = a->ldr; p.mode = a->mode;
    p.vt = a->vt; p.vt_n0 = a->vt_n0; p.vt_tokens = a->vt_tokens > 0 ? a->vt_tokens : 4; p.vt_perm = a->vt_perm ? 1 : 0;
    p.colscale_n = a->colscale_n; p.colscale = a->colscale;
    p.tiles_m = p.tiles_n = 0; p.gm = 0;
    {
        auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
        const int n_out = geglu ? a->N / 2 : a->N;
        p.wide = n_out % 8 == 0 && a->N % 8 == 0 && a->colscale_n % 8 == 0 && (!a->out || (a->ldo % 8 == 0 && a16(a->out))) &&
                 (!a->res || (a->ldr % 8 == 0 && a16(a->res))) && (!a->bias || a16(a->bias)) &&
                 (!a->rowbias || (a->rowbias_ld % 8 == 0 && a16(a->rowbias))) && (!a->vt || a->vt_n0 % 8 == 0);
    }

    // Tile choice: tile_hint (variant<<28 | BN<<16 | BM) from the caller's tuning table; GEGLU needs 64-row wave tiles (BN >= 128).
    int variant = 1, bn = 64, bm = 64;
    static const bool env_narrow = getenv("IDMVTON_EPILOGUE_8B") != nullptr;   // measurement only (A/B of the whole pipeline)
    if ((a->tile_hint & 0x8000) || env_narrow) p.wide = 0;                      // measurement only: force the 8-byte epilogue
    p.res32 = (a->io_flags & IDMVTON_IO_RES_F32) ? 1 : 0;
    p.out32 = (a->io_flags & IDMVTON_IO_OUT_F32) ? 1 : 0;
    p.bias32 = (a->io_flags & IDMVTON_IO_BIAS_F32) ? 1 : 0;
    if (p.bias32) CHECK_ARG(a->bias && p.wide && !geglu && !a->vt && ((uintptr_t)a->bias & 15) == 0, IDMVTON_E_ARG,
                            "gemm_conv: IDMVTON_IO_BIAS_F32 needs a 16-byte aligned bias and the plain 16-byte epilogue (no GEGLU, no vt)");
    CHECK_ARG((a->io_flags & ~15) == 0, IDMVTON_E_ARG, "gemm_conv: io_flags=%d", a->io_flags);
    p.out8 = (a->io_flags & IDMVTON_IO_OUT_F8) ? 1 : 0;
    p.o8_scale = a->f8_out_scale; p.vt8_scale = a->f8_vt_scale;
    if (p.out8) {
        CHECK_ARG(a->io_flags == IDMVTON_IO_OUT_F8 && a->mode == IDMVTON_EPI_NONE && !a->res && !a->rowbias && p.wide,
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_OUT_F8 needs the plain 16-byte epilogue (no activation / residual / rowbias / fp32 IO)");
        CHECK_ARG((!a->out || a->f8_out_scale > 0.f) && (!a->vt || (a->f8_vt_scale > 0.f && a->vt_tokens % 64 == 0 && ((uintptr_t)a->vt & 15) == 0)),
                  IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_OUT_F8: scales > 0, vt_tokens %% 64 == 0 (got %d), 16-byte aligned vt", a->vt_tokens);
    }
    if (p.res32 || p.out32) {
        CHECK_ARG((a->io_flags & ~7) == 0 && !geglu && !a->vt, IDMVTON_E_ARG, "gemm_conv: io_flags=%d (fp32 res / out: no GEGLU, no vt)", a->io_flags);
        CHECK_ARG(!p.res32 || a->res, IDMVTON_E_ARG, "gemm_conv: IDMVTON_IO_RES_F32 without res");
        CHECK_ARG(p.wide, IDMVTON_E_ALIGN, "gemm_conv: the fp32 residual stream needs the 16-byte epilogue (N, ldo, ldr multiples of 8, 16-byte aligned pointers)");
    }
    if (a->tile_hint) { variant = (a->tile_hint >> 28) & 0xf; bn = (a->tile_hint >> 16) & 0xfff; bm = a->tile_hint & 0x7fff; }
    else {
        // No hint: the largest ring tile that still gives every CU a tile (measured rule, profiles/r01_tune_report_*.json:
        // operand delivery per CU is the bound, so arithmetic intensity per tile wins until the grid no longer fills 256 CUs).
        static const int cand[5][2] = {{256, 256}, {128, 256}, {128, 128}, {128, 64}, {64, 64}};
        for (int i = 0; i < 5; ++i) {
            const int n_ = cand[i][0], m_ = cand[i][1];
            if (geglu && n_ < 128) continue;
            if (xattn && (n_ != 128)) continue;          // waves must own 64 columns (one head): the 128-column tiles
            if (a->vt && a->vt_n0 % n_ != 0) continue;
            // a tile wider than the (64-rounded) problem wastes its surplus columns' MFMAs: N = 128 on the 256-column tile ran the
            // VAE's full-resolution 128-channel convolutions at half rate (profiles/r03_v2_prof_default_kernel_stats_by_grid.txt)
            if (i < 4 && (n_ > ((a->N + 63) & ~63) || m_ > ((a->M + 63) & ~63))) continue;
            bn = n_; bm = m_;
            if ((long)((a->N + n_ - 1) / n_) * ((a->M + m_ - 1) / m_) >= 200) break;
        }
    }
    if (geglu) CHECK_ARG(bn >= 128, IDMVTON_E_ARG, "gemm_conv: GEGLU needs BN >= 128 (64-row wave tiles)");
    if (a->vt) CHECK_ARG(a->vt_n0 % bn == 0, IDMVTON_E_ARG, "gemm_conv: vt_n0 %% BN != 0");
    const idmvton_seg& s0 = a->seg[0];
    const bool lin = a->nseg == 1 && a->Ho == 1 && a->Hi == 1 && a->Wo == a->M && a->Wi == a->M && a->stride == 1 &&
                     !a->ups && s0.dy == 0 && s0.dx == 0 && (uint64_t)a->M * s0.pitch * 2 < 0x80000000ull;
    // no hint and the heuristic chose the 256x256 tile for a plain Linear: the hand-scheduled loop (gemm_lin.hip) won on every such shape it
    // was measured on (profiles/r04_gemm_probe_*.log: +5..11 % over the compiler-scheduled tile)
    if (!a->tile_hint && bn == 256 && bm == 256 && lin) { variant = 5; bm = 257; }
    else if (!a->tile_hint && !xattn) {
        // ... and the round-5 measurements for everything else the table has no entry for (profiles/r05_tune_report_v{1..4}*.json: every one of
        // these won or tied on each shape it was tried on): a convolution on the 256x256 tile -> its 16-wave form; the one-workgroup-per-CU 128x128
        // tile -> its 8-wave form; N = 320 (SDXL's first level) -> the 320-column tile when the launch can take it and fills the chip.
        if (bn == 256 && bm == 256) variant = 6;
        else if (bn == 128 && bm == 128) variant = 6;
        if (a->N > 256 && a->N <= 320 && p.wide && !a->vt && !geglu && !p.out8 && (a->M + 191) / 192 >= 200) { variant = 6; bn = 320; bm = 192; }
    }
    hipStream_t st = (hipStream_t)stream;
    return launch_gemm(p, a->dtype == IDMVTON_BF16, variant, bn, bm, lin, st);
}
