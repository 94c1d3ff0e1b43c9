/* idmvton_hip.h -- C ABI of libidmvton_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * The reference (yisol/IDM-VTON) has no FFI of its own: its hot path is Python calling ATen/cuDNN/cuBLAS ops through
 * diffusers.  Each entry point below therefore cites the reference *call site(s)* whose arithmetic
 * it replaces.  Conventions (SURVEY.md 8b "C ABI a native replacement must export"):
 *
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm `tensor.data_ptr()`);
 *   - `int idmvton_<op>(const idmvton_<op>_args*, void* stream)` returns 0 or a negative IDMVTON_E_* code;
 *     `idmvton_last_error()` gives a thread-local message; no exceptions, no allocation, no ownership transfer;
 *   - launches are asynchronous on `stream` (a hipStream_t), no hidden syncs => hipGraph-capturable; stateless.
 *   - activations are NHWC / token-major ([rows][channels], channels contiguous); "ld*" are in ELEMENTS.
 */
#ifndef IDMVTON_HIP_H
#define IDMVTON_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { IDMVTON_OK = 0, IDMVTON_E_SHAPE = -1, IDMVTON_E_DTYPE = -2, IDMVTON_E_ALIGN = -3, IDMVTON_E_LAUNCH = -4,
       IDMVTON_E_ARG = -5 };
enum { IDMVTON_F16 = 0, IDMVTON_BF16 = 1, IDMVTON_F32 = 2, IDMVTON_F8E4M3 = 3 /* OCP e4m3: idmvton_attn_f8 / idmvton_quant_f8 operands */ };

const char* idmvton_last_error(void);
int idmvton_abi_version(void);
int idmvton_sizeof(const char* struct_name); /* sizeof() of an args struct by name, -1 if unknown (binding self-check) */

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_gemm_conv : out[M][N] = epilogue( sum_k X[m][k] * W[n][k] )        (MFMA implicit GEMM, fp32 accumulate)
 *
 * Replaces every nn.Linear / nn.Conv2d on the hot path:
 *   to_q/to_k/to_v/to_out   ip_adapter/attention_processor.py:238,245-246,266,1943,1957-1958,1978-1979,1998
 *   GEGLU proj + ff.net.2   src/attentionhacked_tryon.py:656-667 (diffusers GEGLU: h*gelu(gate))
 *   proj_in / proj_out      src/transformerhacked_tryon.py:342-346,423
 *   ResnetBlock2D conv1/conv2/conv_shortcut/time_emb_proj, Downsample2D, Upsample2D convs
 *                           (constructed at src/unet_block_hacked_tryon.py:1068-1079,1113-1115,2258-2269,2301)
 *   conv_in / conv_out      src/unet_hacked_tryon.py:1245,1386 ; VAE convs (src/tryon_pipeline.py:924,1646,1876)
 *
 * X is a virtual [M][Ktot] matrix assembled from `nseg` K-segments.  Segment s contributes `len` (multiple of 64)
 * consecutive k; row m = (b, oy, ox) of the OUTPUT grid reads input pixel
 *     (iy, ix) = (oy*stride + dy, ox*stride + dx)        [>>1 each when ups=1: fused nearest upsample to the Ho x Wo grid, which is
 *                                                          2Hi x 2Wi or one short of it (diffusers `upsample_size`, odd skip sizes); stride 1]
 * of the NHWC tensor `ptr` ([B][Hi][Wi][pitch]) at channels [coff, coff+len); out-of-image taps read 0.
 * A Linear is one segment with Ho=Hi=1, Wo=Wi=M.  A 3x3 conv is 9 segments; channel-concatenated inputs
 * (torch.cat at unet_block_hacked_tryon.py:2346,2482) are extra segments on a second pointer; the 1x1 conv_shortcut
 * is fused as extra centre-tap segments with its weights appended along K.
 * W is [N][Ktot] (K contiguous: nn.Linear layout; conv weights pre-permuted to [Cout][ky][kx][Cin]).
 * Epilogue: (+ bias[n] + rowbias[(m / rows_per_group)*rowbias_ld + n]) [* colscale for n < colscale_n] + res[m*ldr + n]; mode GEGLU multiplies the two
 * 32-row halves of each 64-row weight block (weights pre-interleaved [32 h | 32 gate]) -> out has N/2 columns.
 * Columns n >= vt_n0 (when vt != NULL) are written TRANSPOSED to vt[(b*(N-vt_n0) + n-vt_n0)*vt_tokens + tok] with
 * (b, tok) = divmod(m, vt_tokens): the V^T layout the attention kernel consumes.
 * ------------------------------------------------------------------------------------------------------------- */
#define IDMVTON_MAX_SEG 24 /* 9 taps x {[hi|lo] pair, hi again} of the split-precision VAE convolutions + shortcut segments */
typedef struct {
    const void* ptr; /* NHWC tensor base                           */
    uint32_t bytes;  /* size of that tensor in bytes (bounds check) */
    int32_t pitch;   /* channels per pixel of that tensor           */
    int32_t coff;    /* first channel read                          */
    int32_t len;     /* channels read (multiple of 64)              */
    int32_t dy, dx;  /* tap offset, padding already folded in       */
} idmvton_seg;

enum { IDMVTON_EPI_NONE = 0, IDMVTON_EPI_GEGLU = 1, IDMVTON_EPI_GELU = 2 /* gelu_erf(acc+bias+rowbias) then +res */,
       IDMVTON_EPI_QUICKGELU = 3 /* x*sigmoid(1.702x): CLIP-L text MLP (transformers hidden_act "quick_gelu") */,
       IDMVTON_EPI_XATTN = 4 /* the GEMM is attn2.to_q and its epilogue IS the cross-attention: see idmvton_xattn */ };
/* mode IDMVTON_EPI_XATTN: out = softmax(q k_0^T / 8) v_0 [+ ip_scale * softmax(q k_1^T / 8) v_1], q = X . W^T never written to memory.
 * Replaces attn2 of every BasicTransformerBlock: to_q + the two SDPAs + their sum (ip_adapter/attention_processor.py:1943,1970-1995; GarmentNet:
 * diffusers Attention with the text keys only, nseg = 1) in ONE launch -- the step-invariant K / V^T tables (77 text + 16 image tokens) are read
 * by the epilogue straight from memory.  Head h = columns [64h, 64h + 64) of W's N outputs.  W's rows must be in the ACCUMULATOR ORDER: inside every
 * group of 16 output channels, bits 2 and 3 of the channel index swapped (row p of W = to_q row (p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1)); K
 * and V^T are the tables idmvton_attn_fwd takes (K [B][k_rows][ldk], V^T [B][N][ldvt] in key order).  Keys >= nk are masked; rows nk..k_rows-1 of K
 * must be readable (k_rows >= round32(nk)).  tokens = GEMM rows per batch element (a multiple of 32).  No bias / residual / colscale. */
typedef struct {
    int32_t nseg;
    const void* k[2]; int32_t ldk[2]; int32_t k_rows[2];
    const void* vt[2]; int32_t ldvt[2];
    int32_t nk[2];               /* nk[0] <= 96 (text), nk[1] <= 32 (image prompt): the fragment slots xattn.cuh holds per segment */
    int32_t tokens; float ip_scale;
} idmvton_xattn;
enum { IDMVTON_IO_RES_F32 = 1, IDMVTON_IO_OUT_F32 = 2, IDMVTON_IO_BIAS_F32 = 4, IDMVTON_IO_OUT_F8 = 8 };
typedef struct {
    int32_t dtype;               /* IDMVTON_F16 | IDMVTON_BF16 (X, W, out, bias, res, rowbias all this type; see io_flags) */
    const void* w; int32_t N; int32_t Ktot;
    int32_t nseg; idmvton_seg seg[IDMVTON_MAX_SEG];
    int32_t M, Ho, Wo, Hi, Wi, stride, ups;
    void* out; int32_t ldo;
    const void* bias;
    const void* rowbias; int32_t rowbias_ld; int32_t rows_per_group;
    const void* res; int32_t ldr;
    int32_t mode;
    void* vt; int32_t vt_n0; int32_t vt_tokens;
    int32_t tile_hint;           /* 0 = auto; else (variant<<28)|(BN<<16)|BM: variant 0 = 2-stage 4-wave tiles 128x128,
                                    128x64, 64x64; variant 1 = LDS-ring tiles 256x256, 128x256 (8 waves), 128x128, 128x64, 64x64; variant 2 =
                                    the 128x256 and 64x64 ring tiles with register-prefetched fragments; variant 6 = more waves per workgroup on the tiles that
                                    run one workgroup per CU (csrc/gemm_tiles_w8.hip): 8-wave 128x128 (64x32 per wave; low nibble of BM: 1 = prefetched
                                    fragments), 12-wave 320x192 (N = 320 in ONE weight tile; no V^T part / GEGLU: such launches fall back to the
                                    variant-1 128x256 tile) and 256x192, 16-wave 256x256 and 128x256; variant 5 = the hand-scheduled Linear loop (csrc/gemm_lin.hip): 256x256 (low nibble
                                    of the BM field = placement form 0 | 1; bit 14 of it, tests only: 5 persistent workgroups) and 256x192 -- a PERSISTENT kernel:
                                    min(tiles, CUs) workgroups walk the tile raster.  Bit 15
                                    (0x8000) forces the 8-byte epilogue (measurement only; default: 16-byte accesses when every
                                    epilogue operand is 16-byte aligned with strides / N multiples of 8).
                                    Filled from the per-shape tuning table (idm-vton_amd/tune_gfx950.json). */
    int32_t vt_perm;             /* 1: write vt in the attention kernel's KEY ORDER: inside every group of 16 tokens, bits 2 and 3 of
                                    the token index are swapped (position p holds token (p&~12)|((p&4)<<1)|((p&8)>>1)), which makes
                                    the 8 keys one half-wave contracts per PV MFMA one aligned 16-byte read.  Needs vt_tokens % 16 == 0.
                                    0: plain transpose (the VAE mid block uses V^T as a GEMM weight). */
    int32_t io_flags;            /* IDMVTON_IO_RES_F32: `res` holds fp32 (ldr in fp32 elements); IDMVTON_IO_OUT_F32: `out` is written as fp32 (ldo in
                                    fp32 elements).  The fp32 RESIDUAL STREAM of a Transformer2DModel: the block-to-block hidden state
                                    (src/attentionhacked_tryon.py:348,384,412: three `+ hidden_states` per block, 210 per TryonNet forward) is
                                    kept in fp32 between the to_out / ff.net.2 epilogues and the next LayerNorm instead of being rounded to
                                    16 bits after every add; both need the 16-byte epilogue (all strides / N multiples of 8).  0 = off.
                                    IDMVTON_IO_BIAS_F32: `bias` holds fp32 (plain 16-byte epilogue only).  With all three, a GEMM whose operands are
                                    bf16 [hi | lo] pairs (idmvton_split, idmvton_groupnorm's IDMVTON_GN_Y_SPLIT) carries fp32-equivalent values end
                                    to end: the SPLIT-PRECISION path of the VAE, which the reference runs in fp32 (src/tryon_pipeline.py:1076-1093,
                                    1868-1880 upcast_vae / force_upcast): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) keeps 16 mantissa bits, and
                                    x.w = hi.w_hi + lo.w_hi + hi.w_lo (+ O(2^-17)) is three K segments of ONE launch: activations [hi | lo] (K-segments
                                    (coff 0, len 2C) and (coff 0, len C)) against weights laid out [w_hi | w_hi | w_lo]; two when w_lo == 0. */
    const idmvton_xattn* xattn;  /* mode IDMVTON_EPI_XATTN only (host pointer, read during the call) */
    int32_t colscale_n; float colscale; /* columns n < colscale_n (multiple of 4) of `out` are multiplied by colscale after bias / rowbias and
                                    before the activation / residual, in fp32 (0: off).  Used to hand the attention kernel a q that is
                                    already scaled by softmax_scale * log2(e) (idmvton_attn_args.q_prescaled). */
    float f8_out_scale, f8_vt_scale; /* io_flags & IDMVTON_IO_OUT_F8 (ABI 7): the projection that feeds idmvton_attn_f8 writes its operands itself -- `out`
                                    (ldo in BYTES) receives e4m3(clamp(+-448, value * f8_out_scale)) after colscale, one byte per column, and `vt`
                                    receives e4m3(clamp(value * f8_vt_scale)) as [B][N - vt_n0][vt_tokens] bytes in the fp8 kernel's SLOT ORDER (position
                                    64t + 32u + 16kb + 4g + j holds key 64t + 32kb + 8g + 4u + j: exactly what idmvton_quant_f8 mode 1 produces from the
                                    16-bit V^T; vt_perm is ignored).  Scales are powers of two (they ride on the MFMA's E8M0 operands).  Needs the plain
                                    16-byte epilogue (no GEGLU / residual / fp32 IO), vt_tokens % 64 == 0, 16-byte aligned
                                    pointers.  One rounding from the fp32 accumulator instead of two (16-bit, then e4m3), no idmvton_quant_f8 launches. */
} idmvton_gemm_conv_args;
int idmvton_gemm_conv(const idmvton_gemm_conv_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_attn_fwd : flash attention, head_dim 64, fp32 online softmax, softmax scale 1/8.
 *
 * Replaces F.scaled_dot_product_attention at ip_adapter/attention_processor.py:258-260 as used by
 *   TryonNet attn1  (src/attentionhacked_tryon.py:334-348): keys = [own tokens ; garment tokens], only the first N
 *                   query rows are computed (the reference discards rows N..2N at :348);
 *   GarmentNet attn1 (src/attentionhacked_garmnet.py:331-342): one key segment.
 * and the two SDPAs + sum at ip_adapter/attention_processor.py:1970-1995 (mode CROSS: out = O_text + ip_scale*O_ip).
 *
 * q   : [B][Nq][ldq]   head h at columns [h*64, h*64+64)
 * k_s : [B_s][k_rows_s][ldk_s] (same head packing; first nk_s rows of each batch are keys)
 * vt_s: V transposed, [B_s][heads*64][ldvt_s] in the kernel's KEY ORDER (gemm_conv's vt_perm = 1: bits 2 and 3 of the key
 *       index swapped inside every group of 16 keys); ldvt >= roundup16(nk), positions of keys >= nk finite
 * Segment s serves batches b >= seg_b0[s], reading its batch (b - seg_b0[s]).  For b < seg_b0[s] the segment's keys
 * are the all-zero garment features of the CFG-unconditional half (src/tryon_pipeline.py:1796): K=V=0 exactly (to_k /
 * to_v have no bias), handled in closed form as `Nk_s` keys with logit 0 and value 0.
 * mode SELF : one softmax over all segments.   mode CROSS: segment 0 and segment 1 are separate softmaxes.
 * ------------------------------------------------------------------------------------------------------------- */
enum { IDMVTON_ATTN_SELF = 0, IDMVTON_ATTN_CROSS = 1 };
typedef struct {
    int32_t dtype; int32_t mode;
    int32_t B, heads, Nq;
    const void* q; int32_t ldq;
    void* out; int32_t ldo;
    int32_t nseg;
    const void* k[2]; int32_t ldk[2];
    const void* vt[2]; int32_t ldvt[2];
    int32_t nk[2];               /* keys per segment (masked beyond nk)                          */
    int32_t k_rows[2];           /* rows per batch element in k[s] (>= nk; 0 means nk)           */
    int32_t seg_b0[2];
    float ip_scale;
    int32_t tune;                /* 0 = auto; else (flags<<24)|(kernel<<16)|(stages<<8)|waves.  kernel 0: attn_kernel, one barrier per 64-key
                                    tile, stages 2 (two-buffer) | 3 | 4 (LDS ring), waves 2|4|8 (any mode).  kernel 2|3: the ping-pong kernel
                                    (SELF mode, 8 waves, stages 2|3; 2 = two workgroups per CU, 3 = one per CU with every fragment of a block
                                    prefetched); flags bit0 = pair waves (w, w^1) instead of (w, w+4), bit1 = no s_setprio, bits 2..3 =
                                    deferred-rescale threshold selector {0: 4, 1: 0 (exact skip only), 2: 8, 3: 2} in log2 units.
                                    kernel 7|8: the ping-pong kernel with each MFMA block's fragments read from LDS one phase early (3 stages;
                                    7 = row sums on the matrix pipe); needs q_prescaled.  kernel 16: the software-pipelined kernel (every wave
                                    overlaps the exponentials of tile i-1 with the MFMAs of PV(i-2) and QK^T(i); exponentials are speculative
                                    and a row-sum bound replaces the row max), waves 8 | 4 = 256 | 128 query rows per workgroup, flags bits
                                    2..3 = row-sum limit selector {0: 512, 1: 32, 2: 8192, 3: 128}; SELF mode, needs q_prescaled.  auto picks
                                    kernel 16 for SELF launches with a pre-multiplied q and >= 128 query rows */
    int32_t q_prescaled;         /* 1: q is already multiplied by softmax_scale * log2(e) = 0.125 * 1.4426950408889634 (gemm_conv's colscale
                                    applies it in fp32 in the projection epilogue, same single rounding as an unscaled q) */
} idmvton_attn_args;
int idmvton_attn_fwd(const idmvton_attn_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_attn_f8 / idmvton_quant_f8 : the fp8 (OCP e4m3) variant of the self-attention above, on the block-scaled MFMA
 * v_mfma_scale_f32_32x32x64_f8f6f4 (BASELINE.json configs[4]: "fp16 + fp8 MFMA attention").  SELF semantics only (one softmax over
 * up to two key segments, closed form for the absent all-zero segment); the cross-attention (93 keys) stays 16-bit.
 *   q8  : [B][Nq][ldq] bytes, head h at bytes [h*64, h*64+64): e4m3(q * softmax_scale * log2(e) * 2^eq)
 *   k8_s: [B_s][k_rows_s][ldk_s] bytes: e4m3(k * 2^ek)
 *   vt8_s: [B_s][heads*64][ldvt_s] bytes: e4m3(v * 2^ev), position 64t + 32u + 16kb + 4g + j of a row holds key 64t + 32kb + 8g + 4u + j
 *          (the k-slot order of the PV contraction; idmvton_quant_f8 mode 1 builds it from the 16-bit key-ordered V^T), whole 64-key
 *          tiles per row, zero-filled beyond nk.
 * qk_scale_exp = -(eq + ek), v_scale_exp = -ev: applied by the MFMA's E8M0 scale operands.  Output in out_dtype (f16 / bf16).
 * Accuracy: e4m3 carries 3 mantissa bits; on N(0,1) operands the output is within 6e-2 .. 1e-1 max-rel of fp32 SDPA on the unquantised
 * operands and 2e-2 of fp32 SDPA on the dequantised ones (tests/kernel_checks.py: check_attn_f8; stated tolerances 1.2e-1 / 3e-2).
 * idmvton_quant_f8: mode 0 rows (dst[r][c] = e4m3(src[r*lds + c] * scale), cols % 16 == 0), mode 1 the V^T re-order above;
 * saturating at +-448; lds in source elements, ldd in bytes.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t out_dtype; int32_t B, heads, Nq;
    const void* q8; int32_t ldq;
    void* out; int32_t ldo;
    int32_t nseg;
    const void* k8[2]; int32_t ldk[2];
    const void* vt8[2]; int32_t ldvt[2];
    int32_t nk[2]; int32_t k_rows[2]; int32_t seg_b0[2];
    int32_t qk_scale_exp; int32_t v_scale_exp;
} idmvton_attn_f8_args;
int idmvton_attn_f8(const idmvton_attn_f8_args* a, void* stream);
typedef struct {
    int32_t dtype; int32_t mode; int32_t rows, cols;
    const void* src; int32_t lds;
    void* dst; int32_t ldd;
    float scale;
} idmvton_quant_f8_args;
int idmvton_quant_f8(const idmvton_quant_f8_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_attn_small : softmax(scale * q k^T [+ causal mask]) v for the one-off conditioning encoders -- the CLIP text towers
 * (transformers CLIPTextModel / CLIPTextModelWithProjection as called at src/tryon_pipeline.py:511-743: 77 tokens, causal) and
 * the CLIP-H vision tower (:460-482: 257 tokens, head_dim 80).  Any even head_dim <= 128, Lk <= 1024, fp32 arithmetic; one wave
 * per query row.  q/k/v/out: [B][L][ld*], head h at columns [h*d, h*d + d).  causal: row i sees keys j <= i + (Lk - Lq).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype; int32_t B, heads, Lq, Lk, d;
    const void* q; int32_t ldq;
    const void* k; int32_t ldk;
    const void* v; int32_t ldv;
    void* out; int32_t ldo;
    float scale; int32_t causal;
} idmvton_attn_small_args;
int idmvton_attn_small(const idmvton_attn_small_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_layernorm : y = LN(x)*gamma+beta over the last dim, fp32 statistics (eps inside sqrt), optional second copy.
 * Replaces nn.LayerNorm at src/attentionhacked_tryon.py:199,229,256 (eps 1e-5 :147); `y2` is the GarmentNet feature
 * export of src/attentionhacked_garmnet.py:321-322 fused into the same pass.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype; int32_t rows, C;
    const void* x; int32_t ldx;
    const void* gamma; const void* beta; float eps;
    void* y; int32_t ldy;
    void* y2; int32_t ldy2;
    int32_t x_f32;               /* 1: x is fp32 (ldx in fp32 elements): the fp32 residual stream written by gemm_conv's IDMVTON_IO_OUT_F32 */
} idmvton_layernorm_args;
int idmvton_layernorm(const idmvton_layernorm_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_groupnorm : NHWC GroupNorm (+ optional SiLU), fp32/fp64 statistics.  Three launches inside one call:
 * stats (per-block sum / sum-of-squares partials, no atomics: bit-reproducible), finalize (block-order fold), apply.  The input may be the channel-concat
 * of two tensors (x: C1 channels, x2: C-C1 channels) -- the torch.cat of skip connections is never materialised.
 * Replaces nn.GroupNorm (+F.silu) in diffusers ResnetBlock2D (eps 1e-5; src/unet_hacked_tryon.py:325,606),
 * Transformer2DModel.norm (eps 1e-6; src/transformerhacked_tryon.py:148,329), conv_norm_out (:1383-1385), VAE norms.
 * `stats` is caller-owned scratch of idmvton_groupnorm_stats_doubles(B, HW, C, groups) doubles (its capacity goes in
 * `stats_doubles`); it needs no initialisation.
 * ------------------------------------------------------------------------------------------------------------- */
enum { IDMVTON_GN_X_F32 = 1 /* x (and x2) hold fp32 */, IDMVTON_GN_Y_SPLIT = 2 /* y is [B][HW][2C] = [hi | lo], hi = dtype(v), lo = dtype(v - hi) */,
       IDMVTON_GN_AFFINE_F32 = 4 /* gamma / beta hold fp32 */ };
typedef struct {
    int32_t dtype; int32_t B, HW, C, groups;
    const void* x; int32_t C1;  /* channels taken from x  (== C when x2 is NULL) */
    const void* x2;
    const void* gamma; const void* beta; float eps; int32_t silu;
    void* y;                    /* [B][HW][C]  ([B][HW][2C] with IDMVTON_GN_Y_SPLIT) */
    double* stats; int32_t stats_doubles;
    int32_t flags;              /* IDMVTON_GN_*: the split-precision VAE path (see idmvton_gemm_conv_args.io_flags); 0 = the 16-bit form */
} idmvton_groupnorm_args;
int idmvton_groupnorm(const idmvton_groupnorm_args* a, void* stream);
int idmvton_groupnorm_stats_doubles(int B, int HW, int C, int groups);

/* ---------------------------------------------------------------------------------------------------------------
 * Small fused elementwise ops of the loop body (src/tryon_pipeline.py:1769-1823).
 * ------------------------------------------------------------------------------------------------------------- */
/* Build the TryonNet input (tryon_pipeline.py:1769,1777): NHWC [2B][hw][cpad] = cat([latents]*2 | mask | masked | pose),
 * channels [13, cpad) zero.  latents: fp32 NCHW [B][4][hw]; cond: dtype NHWC [2B][hw][9] (step-invariant). */
typedef struct {
    int32_t dtype; int32_t B, hw, cpad;
    const float* latents; const void* cond; void* out;
} idmvton_pack_input_args;
int idmvton_pack_input(const idmvton_pack_input_args* a, void* stream);

/* CFG combine + scheduler step (tryon_pipeline.py:1814-1823; diffusers DDPMScheduler.step / DDIM eta=0):
 * eps = e_u + g*(e_c - e_u);  latents = c_x*latents + c_eps*eps + sigma*noise.   eps_nhwc: dtype NHWC [2B][hw][ldc]
 * (channels 0..3 used); latents/noise: fp32 NCHW [B][4][hw]; coef: DEVICE pointer to {c_x, c_eps, sigma, g}. */
typedef struct {
    int32_t dtype; int32_t B, hw, ldc;
    const void* eps_nhwc; float* latents; const float* noise; const float* coef;
} idmvton_cfg_step_args;
int idmvton_cfg_step(const idmvton_cfg_step_args* a, void* stream);

/* Layout / dtype conversion at the pipeline edges: NCHW fp32 <-> NHWC dtype with channel padding. */
enum { IDMVTON_LAYOUT_SPLIT = 1 /* to_nhwc: dst is NHWC [2*cpad] = [hi | lo] */, IDMVTON_LAYOUT_NHWC_F32 = 2 /* to_nchw: src NHWC holds fp32 */ };
typedef struct {
    int32_t dtype; int32_t B, C, HW, cpad; int32_t to_nhwc; /* 1: src fp32 NCHW -> dst dtype NHWC[cpad]; 0: reverse */
    const void* src; void* dst; float scale; float shift;    /* dst = src*scale + shift */
    int32_t flags;                                           /* IDMVTON_LAYOUT_*: edges of the split-precision VAE path */
} idmvton_layout_args;
int idmvton_layout(const idmvton_layout_args* a, void* stream);

/* Diagonal-Gaussian posterior sample of the VAE (tryon_pipeline.py:255):
 * z = (mean + exp(0.5*clamp(logvar,-30,20))*noise) * scale ; moments: dtype NHWC [B][hw][ldm] (mean 0..3, logvar 4..7),
 * noise fp32 NCHW [B][4][hw], z fp32 NCHW [B][4][hw]. */
typedef struct {
    int32_t dtype; int32_t B, hw, ldm;
    const void* moments; const float* noise; float* z; float scale;
} idmvton_vae_sample_args;
int idmvton_vae_sample(const idmvton_vae_sample_args* a, void* stream);

/* In-place row softmax(scale*x), fp32 statistics.  VAE mid-block attention (1 head x 512:
 * src/unet_block_hacked_tryon.py:585-597, upcast_softmax=True) = gemm_conv (QK^T) + softmax_rows + gemm_conv (PV). */
typedef struct {
    int32_t dtype; int32_t rows, n, ld;
    void* x; float scale;
    void* y_split; int32_t ldy;  /* NULL: in place, x of `dtype`.  Else x holds fp32 [rows][ld] (read only) and the probabilities are written as the
                                    pair [rows][ldy] = [hi (n) | lo (n)] of `dtype`: the A operand of the split-precision P.V product */
    int32_t n_valid;             /* 0 = n.  Else the softmax runs over the first n_valid (<= n) columns and columns n_valid..n-1 are written as 0:
                                    a key count that is not a multiple of 64 (latent H*W of an odd size) is padded up to the GEMM's K granule */
} idmvton_softmax_args;
int idmvton_softmax_rows(const idmvton_softmax_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * idmvton_split : fp32 -> bf16 / fp16 [hi | lo] pairs, the operands of the split-precision GEMMs (idmvton_gemm_conv_args.io_flags).
 * The reference keeps these tensors in fp32 (upcast VAE: src/tryon_pipeline.py:1076-1093,1868-1880); here a value x travels as
 * hi = dtype(x), lo = dtype(x - hi).  src: fp32 [rows][cols], row stride lds (elements).
 *   mode IDMVTON_SPLIT_ACT : dst [rows][ldd] = [hi (cols) | lo (cols)]                       (activation side: K-segments 2C + C)
 *   mode IDMVTON_SPLIT_W3  : dst [rows][ldd] = [hi | hi | lo]                                (weight side of an activation x activation product)
 *   mode IDMVTON_SPLIT_W3T : dst [cols][ldd] = the W3 form of src^T (V^T of the VAE mid-block attention), ldd >= 3*rows
 * ------------------------------------------------------------------------------------------------------------- */
enum { IDMVTON_SPLIT_ACT = 0, IDMVTON_SPLIT_W3 = 1, IDMVTON_SPLIT_W3T = 2 };
typedef struct {
    int32_t dtype; int32_t mode; int32_t rows, cols;
    const float* src; int32_t lds;
    void* dst; int32_t ldd;
} idmvton_split_args;
int idmvton_split(const idmvton_split_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The path's one collective: start-up broadcast of a packed weight arena from `root` to every rank over RCCL / xGMI.
 * No reference counterpart: every process of the reference loads every checkpoint itself (inference.py:232-274) and there are no
 * cross-image collectives.  One process per GPU; rank 0 calls idmvton_rccl_unique_id and ships the 128 bytes to the other ranks by
 * any side channel (the Python host uses the torch.distributed store); every rank then builds the communicator and calls
 * idmvton_rccl_bcast_arena on its own copy of the arena (in place; <= chunk_bytes per ncclBroadcast, 0 = 512 MiB), asynchronously
 * on `stream`.  librccl.so is dlopen'ed on first use: single-GPU users never load it.
 * ------------------------------------------------------------------------------------------------------------- */
int idmvton_rccl_unique_id(void* id128);                                  /* out: 128 bytes (ncclUniqueId) */
int idmvton_rccl_comm_init(const void* id128, int rank, int world, void** comm_out);
int idmvton_rccl_bcast_arena(void* comm, void* buf, uint64_t bytes, int root, uint64_t chunk_bytes, void* stream);
int idmvton_rccl_comm_destroy(void* comm);

/* Hardware layout probes (tests/test_probe_gpu.py): run one MFMA / LDS-transpose instruction on caller data. */
int idmvton_probe_mfma(int which, const void* a, const void* b, float* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif
