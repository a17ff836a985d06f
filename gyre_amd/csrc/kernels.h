// Host-side launchers of the hand-written gfx950 kernels.  All enqueue on `st`
// and return 0 / negative gyre_status (message via gyre_last_error()).
#pragma once
#include "common.h"

// ---- layout / small ops (kernels_elem.hip) ----------------------------------
int launch_nchw_to_nhwc(hipStream_t st, const void* x, int dtype, int B, int C, int HW, int Cpad, bf16_t* y);
int launch_add_nchw_into_nhwc(hipStream_t st, const void* r, int dtype, int B, int C, int HW, int Cpad, bf16_t* y);
int launch_ctx_to_bf16(hipStream_t st, const void* x, int dtype, size_t n, bf16_t* y);
// dst[o][0:inner] = dst[o][inner:2*inner] = src[o][0:inner] (bytes, multiples of 16), o < outer
int launch_dup_batch(hipStream_t st, const void* src, void* dst, size_t outer, size_t inner_bytes);
int launch_add_f32(hipStream_t st, float* y, const float* x, size_t n);
// debug (GYRE_VERIFY_HINTS=1): flag |= 1 if t[b] != t[0] for some b, |= 2 if the two halves of x (bytes_per_half each) differ
int launch_verify_hints(hipStream_t st, const int64_t* t, int B, int check_t, const void* x, size_t bytes_per_half, int check_pairs, int* flag);
int launch_timestep_embedding(hipStream_t st, const int64_t* t, int B, int dim, int flip, float shift, float* out);
// out[b][n] = sum_k f(x[b][k]) * W[n][k] + bias[n]; x f32 [B][K], W bf16 [N][K], out f32 [B][ldo].  act_in_silu: f = SiLU,
// and for B > 4 x is OVERWRITTEN with SiLU(x) (its only use on the time-embedding path; B <= 4 applies it on load)
int launch_rowvec_linear(hipStream_t st, float* x, int B, int K, const bf16_t* W, const float* bias, int N,
                         int act_in_silu, float* out, int ldo);
// out[b] = Linear(timestep_embedding(t[b])) - one launch for B <= 4 (emb_scratch [B][dim] floats is used above that)
int launch_timestep_linear(hipStream_t st, const int64_t* t, int B, int dim, int flip, float shift, float* emb_scratch,
                           const bf16_t* W, const float* bias, int N, float* out, int ldo);

// GroupNorm over NHWC bf16, optional second source for the skip-concat case:
// channels [0,C1) come from x (pixel stride C1), [C1,C) from x2 (pixel stride C-C1).
struct GnParams {
    const bf16_t* x; const bf16_t* x2; int C1;
    int B, HW, C, G;
    const float* gamma; const float* beta; float eps; int silu;
    float* partial;      // [B][nchunks][G][2]
    float* scale_shift;  // [B][2][C]   (a = rstd*gamma, b = beta - mean*a)
    int nchunks;
    bf16_t* y;           // [B][HW][C]
    float* mean_rstd = nullptr;  // optional [B][G][2] (backward pass)
    // statistics left by the producers of x / x2 (GemmParams::colstat_out): [B][cs_chunks][C_src / cs_unit][2] per source.
    // When cs_x is set (and cs_x2 for a dual-source call) launch_groupnorm_stats is not needed: launch_groupnorm_apply
    // finishes the group statistics from them in its prologue (cs_unit divides C1, C - C1 and C / G).
    const float* cs_x = nullptr; int cs_x_chunks = 0;
    const float* cs_x2 = nullptr; int cs_x2_chunks = 0;
    int cs_unit = 0;
};
size_t gn_workspace_bytes(int B, int HW, int C, int G);
int gn_pick_chunks(int B, int HW, int C);
// 0 = off; n > 0: plan every split-K factor as if the batch held n samples (results then do not depend on B)
int gemm_set_batch_invariant(int canonical_samples);
int gemm_get_batch_invariant();
// everything thread-local the planner's choices depend on (tuning flags, forced config, batch-invariant size, packed-weight
// scratch) folded into one value: callers that memoise plans key on it
long gemm_planner_state();
int launch_groupnorm_stats(hipStream_t st, const GnParams& p);   // partial sums + finalize -> scale_shift
int launch_groupnorm_apply(hipStream_t st, const GnParams& p);   // y = act(a*x+b)
bool gn_use_small(int HW, int C, int C1, int G);                 // one-launch path for small feature maps
bool gn_accepts_colstats(const GnParams& p);                     // cs_unit / shapes fit the producer-statistics form of the apply
int launch_groupnorm_small(hipStream_t st, const GnParams& p);
// GroupNorm (affine only) folded into the Linear / 1x1 conv W[N][C] (+ bias) that consumes it: per sample b the scaled weights
// Wf[b][N][C] (bf16) and the bias row bf[b][N] (fp32); statistics from p.cs_x (producer column statistics) or p.partial
// (launch_groupnorm_stats).  Consumed by a GEMM with GemmParams::w_sample_stride = N * C and rowbias = bf.
int launch_gn_fold(hipStream_t st, const GnParams& p, const bf16_t* W, const float* bias, int N, bf16_t* Wf, float* bf);
int launch_layernorm(hipStream_t st, const bf16_t* x, int M, int C, const float* gamma, const float* beta, float eps,
                     bf16_t* y);
// stats[m] = (rstd, rstd * mean) of row m: the input of a GEMM with the LayerNorm folded in (GemmParams::ln_stats)
int launch_layernorm_stats(hipStream_t st, const bf16_t* x, int M, int C, float eps, float* stats);

// weight repack: OIHW (any dtype) -> [O][KH][KW][Ipad] bf16 ; linear [O][I] -> bf16 (optionally GEGLU-interleaved)
int launch_nhwc_to_nchw_f32(hipStream_t st, const bf16_t* x, int B, int C, int HW, int Cpad, float* y);
int launch_repack_conv(hipStream_t st, const void* w, int dtype, int O, int I, int KH, int KW, int Ipad, bf16_t* out,
                       float scale = 1.f);
int launch_repack_linear(hipStream_t st, const void* w, int dtype, int O, int I, int geglu_interleave, bf16_t* out);
int launch_cast_f32(hipStream_t st, const void* w, int dtype, size_t n, int geglu_interleave, float* out, float scale = 1.f);
int launch_copy_probe(hipStream_t st, const void* src, void* dst, size_t bytes);

// ---- MFMA GEMM / implicit-GEMM conv (kernels_gemm.hip) -------------------------
enum { GEMM_LINEAR = 0, GEMM_CONV3 = 1 };
enum { OUT_BF16 = 0, OUT_NCHW = 1, OUT_BF16_T = 2 };
struct GemmParams {
    // A operand: [M][K] rows (linear) or NHWC image gathered through a 3x3 window (conv)
    const bf16_t* A = nullptr; const bf16_t* A2 = nullptr; int C1 = 0;  // dual source split along K / channel
    int lda = 0, lda2 = 0;      // row (pixel) stride of each source, elements
    int mode = GEMM_LINEAR;
    int Hi = 0, Wi = 0, Cin = 0;  // conv: SOURCE spatial dims (before the fused 2x upsample) and channels (multiple of 8)
    int Ho = 0, Wo = 0, stride = 1, pad = 1, ups = 0;
    // B operand: W[N][K] bf16, K contiguous
    const bf16_t* W = nullptr; int K = 0; int N = 0; int M = 0;
    // epilogue
    const float* bias = nullptr;      // [N] (GEGLU: [2N] interleaved like the weight rows)
    const float* rowbias = nullptr;   // [M / rows_per_sample][ld_rowbias], per-sample per-channel (time embedding)
    int rows_per_sample = 1; int ld_rowbias = 0;
    const bf16_t* residual = nullptr; int ldr = 0;
    int geglu = 0;                    // W has 2*N_out rows interleaved in 16-row value/gate groups; N == 2*N_out
    void* out = nullptr; int ldc = 0; int out_mode = OUT_BF16; int out_dtype = 1;
    int tokens_per_batch = 0; int ldt = 0;  // OUT_BF16_T: out[(b*N + col)*ldt + tok]
    // fused Q|K|V projection (8-wave kernels, OUT_BF16): columns >= vt_col0 are written transposed to
    // vt_out[(b*(N - vt_col0) + col - vt_col0)*ldt + tok] instead of `out` (V^T for the attention kernel)
    bf16_t* vt_out = nullptr; int vt_col0 = 0;
    const bf16_t* zero_page = nullptr;      // >= 16 zero bytes in global memory (filled in by launch_gemm)
    int force_cfg = 0;                      // tests/tuning: 0 auto, else tile-config id (see launch_gemm)
    int debug = 0;                          // tuning ablations: bit0 = no operand loads in the K loop, bit1 = no MFMAs
    int Hup = 0, Wup = 0;         // ups: logical size of the upsampled image (0 = 2*Hi x 2*Wi); < 2x crops the last row/col
    int samples = 0;              // batch entries folded into M (0 = unknown); used by the batch-invariant planner
    float* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;  // fp32 partial slabs [splits][M][N] (see gemm_plan)
    // batched form (4-wave kernels only): problem b uses A + b*bsA, W + b*bsW, out + b*bsC (element strides); no bias /
    // residual / split-K.  Used for the per-sample token-similarity matrix of ToMe.
    int batch = 1; size_t bsA = 0, bsW = 0, bsC = 0;
    // LayerNorm folded into the GEMM (8-wave kernels, linear mode, one source, no split-K): A holds the RAW rows, W the
    // gamma-scaled weights W'[n][k] = W[n][k] * gamma[k], bias[n] = sum_k beta[k] W[n][k] (+ the layer's bias),
    // ln_colsum[n] = sum_k W'[n][k] (launch_ln_fold) and ln_stats[m] = (rstd, rstd * mean) of row m
    // (launch_layernorm_stats, one streaming read of the rows).  The epilogue applies
    // out = rstd * acc - rstd * mean * colsum + bias: the normalised tensor is never written or re-read.
    // (Accumulating the row sums inside the K loop from the operand fragments - v_dot2c_f32_bf16, no extra pass - was built
    // and measured: every N tile repeats the sums at ~20 cycles per packed pair, slower than the separate statistics pass
    // for every layer of the UNet; profiles/README.md.)
    const float* ln_colsum = nullptr; const float* ln_stats = nullptr;
    // Statistics straight from the kernel that PRODUCED the rows: rowstat_out (8-wave kernels, staged epilogue, no GEGLU /
    // split-K / fused V^T; gemm_rowstat_parts() > 0) receives [tiles_n][M][2] = per row the sum and the sum of squares of the
    // bf16-rounded outputs of each N tile, reduced in a fixed order (lanes -> waves -> tile).  A consumer with the folded
    // LayerNorm takes them as ln_parts / ln_nparts instead of ln_stats and finishes mean / rstd in its epilogue
    // (variance as E[x^2] - mean^2 in fp32): the separate statistics pass disappears.
    float* rowstat_out = nullptr;
    const float* ln_parts = nullptr; int ln_nparts = 0; float ln_eps = 1e-5f;
    // GroupNorm statistics straight from the kernel that PRODUCED the tensor (the column counterpart of rowstat_out):
    // colstat_out [M / colstat_rows][N / colstat_unit][2] = per block of colstat_rows consecutive output rows and per unit of
    // colstat_unit consecutive channels the sum and the sum of squares of the bf16-rounded outputs, reduced in a fixed order
    // (rows of a 16- / 32-row slab -> slabs -> wave tiles -> channels of the unit; no atomics).  colstat_rows is dictated by
    // the kernel the planner picks: gemm_colstat_rows(p) (0: this problem cannot; the consumer then runs its own statistics
    // pass).  The row blocks never straddle a sample (rows_per_sample % colstat_rows == 0 is part of the condition), so a
    // GroupNorm over the tensor - alone or as one source of a skip concat - finishes mean / rstd from these partials
    // (GnParams::cs_x / cs_x2) and the statistics pass over the tensor disappears.
    float* colstat_out = nullptr; int colstat_unit = 0;
    // A-resident kernel (kernels_gemm_ar.hip, tile config 30: K = 320 / 640 linear problems).  It reads W from a fragment-ordered
    // copy: ar_ok = the caller can provide one (the planner only then picks the config; the model runtime packs per handle on
    // first use and sets w_packed, tests pass a scratch buffer through gyre_debug_set_ar_workspace); no_ar = the caller wants a
    // feature that kernel lacks from this launch (column statistics for a GroupNorm)
    const void* w_packed = nullptr; int ar_ok = 0, no_ar = 0;
    // BLOCKED copy of W for the LDS-DMA tile kernels (tile configs 4 - 8, 12, 20 - 24, 32; round 5): 1-KiB blocks of 8 weight rows x
    // 64 k, block (n >> 3, k >> 6) at ((n >> 3) * (K / 64) + (k >> 6)) * 512 elements, row-major inside.  A wave's LDS-DMA request
    // (8 rows x 128 B) then reads ONE contiguous KiB instead of eight lines 2 K bytes apart: tools/ubench/l2_bw.hip measures 82 - 125
    // GB/s per CU from L2 for requests that stay inside 16 KB and 19 - 60 for requests spread over more 4-KiB pages, which is what
    // every weight request with K >= 1280 (and every 3x3 conv's: K = 9 Cin) was.  K % 64 == 0, N % 8 == 0; kernels that do not
    // know the layout ignore the field and read W.  (launch_w_block makes the copy; the model runtime caches one per weight.)
    const bf16_t* W_blk = nullptr;
    // per-SAMPLE weights (8-wave tile configs 4 - 8, linear mode): rows of sample b = row / rows_per_sample read W + b *
    // w_sample_stride (elements); row blocks never straddle a sample (gemm_per_sample_w_ok).  The GroupNorm folded into a
    // Transformer2D's proj_in (launch_gn_fold); the per-sample bias travels as `rowbias`
    size_t w_sample_stride = 0;
    // 1x1 SHORTCUT folded into a 3x3 convolution as extra K steps (round 6; pipelined 256x320 tile only, stride 1, pad 1, no upsample):
    //   out = conv3x3(A | A2) + (S | S2) Wsc^T + bias        (the resnet's conv2 + conv_shortcut of a channel-changing / concat block)
    // Behind the nine taps of every 64-channel chunk of the conv input the K loop walks sc_K more channels of a second image pair
    // (same H x W as the output, split at sc_C1 like A | A2) through the CENTRE tap.  W holds [N][9 Cin + sc_K] rows (the conv's rows
    // followed by the shortcut's: launch_concat_rows), K = 9 Cin + sc_K, bias = the sum of the two biases.  gemm_conv_shortcut_ok(p)
    // says whether launch_gemm would run `p` on a kernel that knows the form.
    const bf16_t* sc_A = nullptr; const bf16_t* sc_A2 = nullptr; int sc_lda = 0, sc_lda2 = 0, sc_C1 = 0, sc_K = 0;
    // conv: circular instead of zero padding along x (bit 0) / y (bit 1) - the reference's request option "tiling"
    // (unified_pipeline.py:1671-1712 patches every Conv2d's own padding to F.pad(mode="circular")); 4-wave tile configs only
    int wrap = 0;
};
// rows per colstat_out row block launch_gemm would use for `p` (p.colstat_unit and p.rows_per_sample set); 0: unsupported
int gemm_colstat_rows(const GemmParams& p);
// number of N tiles (= partial sums per row) launch_gemm would emit into rowstat_out for `p`; 0: this problem cannot
int gemm_rowstat_parts(const GemmParams& p);
// true when launch_gemm would run `p` on a kernel that reads per-sample weights (w_sample_stride; rows_per_sample set)
bool gemm_per_sample_w_ok(const GemmParams& p);
// true when launch_gemm would run `p` (sc_* set) on the kernel that folds the 1x1 shortcut into the 3x3 convolution
bool gemm_conv_shortcut_ok(const GemmParams& p);
// out[n][0:K1] = a[n][0:K1], out[n][K1:K1+K2] = b[n][0:K2] (bf16 rows; K1, K2 multiples of 8): the folded conv + shortcut weight
int launch_concat_rows(hipStream_t st, const bf16_t* a, int K1, const bf16_t* b, int K2, int N, bf16_t* out);
// true when launch_gemm would run `p` (ln_colsum set or not) on a kernel that supports the folded LayerNorm
bool gemm_ln_fusable(const GemmParams& p);
// W'[n][k] = bf16(W[n][k] * gamma[k]); colsum[n] = sum_k W'[n][k]; bias_out[n] = sum_k beta[k] * W[n][k] + (bias ? bias[n] : 0)
int launch_ln_fold(hipStream_t st, const bf16_t* W, int N, int K, const float* gamma, const float* beta, const float* bias,
                   bf16_t* Wf, float* colsum, float* bias_out);
// A-resident kernel (tile config 30): bytes of / conversion into the fragment-ordered weight copy it reads (GemmParams::w_packed)
size_t gemm_ar_packed_bytes(int N, int K);
int launch_ar_pack(hipStream_t st, const bf16_t* W, int N, int K, void* out);
int launch_w_block(hipStream_t st, const bf16_t* W, int N, int K, bf16_t* out);   // out: N * K elements (GemmParams::W_blk)
bool gemm_w_block_wanted(const GemmParams& p);      // the planner's kernel for `p` reads a blocked copy and the shape gains from one
// Pure function of the problem shape: tile configuration, K splits and the split-K workspace it needs.
// The caller allocates `ws_bytes` (or passes none: the launch then falls back to a single split).
struct GemmPlan { int cfg; int splits; size_t ws_bytes; };
// 3x3 / stride 1 / zero padding 1 convolution of NHWC bf16 x [B][H][W][C] into O <= 16 channels, written as NCHW of the given
// runtime dtype (kernels_conv_out.hip): W [O][3][3][C] bf16, bias [O] or null.  conv_out_supports: C % 64 == 0 and the weights fit LDS.
bool conv_out_supports(int C, int O);
int launch_conv_out(hipStream_t st, const bf16_t* x, int B, int H, int W, int C, const bf16_t* Wt, const float* bias, int O,
                    void* out, int out_dtype);
GemmPlan gemm_plan(const GemmParams& p);
int launch_gemm(hipStream_t st, const GemmParams& p);


// ---- flash-style attention (kernels_attn.hip) ------------------------------------
struct AttnParams {
    const bf16_t* q; int ldq;   // [B][Nq][ldq], head h at column h*D
    const bf16_t* k; int ldk;   // [B][Nk][ldk]
    const bf16_t* vt; int ldvt; // [B][H*D][ldvt] (V transposed: token index contiguous)
    bf16_t* o; int ldo;         // [B][Nq][ldo]
    int B, H, Nq, Nk, D;
    int k_prescaled = 0;        // K already carries log2(e)/sqrt(D) (folded into the to_k weights at repack time)
    int always_check = 0;       // tuning (k_attn3): keep the per-tile overflow check instead of the optimistic first pass
    unsigned* redo_counter = nullptr;   // incremented by every workgroup that repeats its pass (per device; gyre_debug_attn_redo_count)
};
int launch_attention(hipStream_t st, const AttnParams& p);

// ---- fused cross-attention block: to_q (+ folded LayerNorm) -> attention over the cached text keys -> to_out + residual (kernels_xattn.hip) ----
struct XattnParams {
    const bf16_t* x; int ldx;                 // [M][C] rows BEFORE the LayerNorm (also the residual)
    const bf16_t* wq;                         // gamma-folded to_q weights [C][C] (launch_ln_fold)
    const float* q_colsum; const float* q_bias;       // [C] colsum of W', folded bias
    const float* ln_parts; int ln_nparts; const float* ln_stats; float ln_eps;
    const bf16_t* k; const bf16_t* vt; int ldvt;     // context cache: K [B][Nk][C] (prescaled), V^T [B][C][ldvt]
    const bf16_t* wo; const float* bo;       // to_out [C][C], [C]
    bf16_t* out; int ldo;                     // [M][C]
    float* rowstat_out;                       // [M][2] (one partial per row) or null
    int M, rows_per_sample, Nk, heads;
};

bool xattn_supports(int C, int heads, int Nq, int Nk, int M);
int launch_xattn(hipStream_t st, const XattnParams& p, int C);

// ---- input-gradient kernels of the CLIP-guided mode (kernels_bwd.hip) ---------------------------------------------
struct GnBwdParams {
    const bf16_t* x; const bf16_t* x2; int C1;       // forward input (x2 = second source of the skip concat, or null)
    int B, HW, C, G;
    const float* gamma; const float* beta; float eps; int silu;
    const bf16_t* dy;                                // [B][HW][C]
    const bf16_t* addend;                            // optional [B][HW][C1], added to dx (identity-shortcut gradient)
    bf16_t* dx; bf16_t* dx2;                         // [B][HW][C1], [B][HW][C - C1]
    // filled in by launch_groupnorm_bwd from its workspace
    float* partial = nullptr; const float* fwd_partial = nullptr; int nchunks = 0;
};
size_t gn_bwd_workspace_bytes(int B, int HW, int C, int G);
int launch_groupnorm_bwd(hipStream_t st, GnBwdParams p, void* ws);
int launch_layernorm_bwd(hipStream_t st, const bf16_t* x, const bf16_t* dy, int M, int C, const float* gamma, float eps,
                         const bf16_t* addend, bf16_t* dx);
// pre: [M][2F] GEGLU pre-activation in the interleaved column order of the packed weight; dy: [M][F]; dpre: [M][2F]
int launch_geglu_bwd(hipStream_t st, const bf16_t* pre, const bf16_t* dy, size_t M, int F, bf16_t* dpre);
int launch_pool2_sum(hipStream_t st, const bf16_t* du, int B, int H, int W, int Hu, int Wu, int C, bf16_t* dx);
int launch_zero_stuff2(hipStream_t st, const bf16_t* dy, int B, int Ho, int Wo, int H, int W, int C, bf16_t* dz);
int launch_add_bf16(hipStream_t st, bf16_t* y, const bf16_t* x, size_t n);
int launch_conv_weight_t(hipStream_t st, const bf16_t* w, int O, int I, bf16_t* wt);   // [O][9][I] -> [I][9][O], window rotated
int launch_transpose(hipStream_t st, const bf16_t* in, int ld_in, int R, int C, bf16_t* out, int ld_out, int batch,
                     size_t bs_in, size_t bs_out);
struct AttnBwdParams {
    const bf16_t* q; int ldq;      // [B][Nq][ldq], head h at column h*D
    const bf16_t* k; int ldk;      // [B][Nk][ldk]
    const bf16_t* v; int ldv;      // [B][Nk][ldv]  (row-major, unlike the forward kernel's V^T)
    const bf16_t* o; int ldo;      // forward output [B][Nq][ldo]
    const bf16_t* d_o; int lddo;   // its gradient
    const bf16_t* kt; int ldkt;    // K transposed [B][H*D][ldkt], ldkt >= Nk rounded up to 32, pad columns zero
    const bf16_t* qt; const bf16_t* d_ot; int ldqt;   // Q^T, dO^T [B][H*D][ldqt] (only read when dk != null)
    bf16_t* dq; int lddq;
    bf16_t* dk; int lddk; bf16_t* dv; int lddv;       // dk == null: queries only (cross-attention: the context gets no gradient)
    void* stats;                   // attn_bwd_stats_bytes(B, H, Nq)
    int B, H, Nq, Nk, D;
    int k_prescaled;               // K carries log2(e)/sqrt(D) (see AttnParams)
    // filled in by launch_attention_bwd
    float* lse = nullptr; float* delta = nullptr; int NqPad = 0; float alpha = 0.f, beta = 0.f;
};
size_t attn_bwd_stats_bytes(int B, int H, int Nq);
bool attn_bwd_needs_transposes(int D);     // false: the LDS-tiled kernels (D <= 160) ignore kt / qt / d_ot
int launch_attention_bwd(hipStream_t st, AttnBwdParams p);

// ---- ToMe: bipartite soft matching + merge of self-attention K / V tokens (kernels_tome.hip) ---------------------------
struct TomeParams {
    const bf16_t* k; int ldk;     // [B][N][ldk]  keys (row-major; any column offset already applied)
    const bf16_t* v; int ldv;     // [B][N][ldv]  values (row-major)
    int B, N, C, r;               // r tokens of the "a" half (even tokens) are merged into their best "b" (odd) match
    bf16_t* k_out;                // [B][N - r][C]
    bf16_t* vt_out; int ldvt;     // [B][C][ldvt]  merged values, transposed (what the attention kernel streams)
    void* ws; size_t ws_bytes;    // tome_workspace_bytes(B, N, C)
    int* order_out = nullptr;     // optional [B][N/2]: a-token indices by descending best-match score
    int* node_idx_out = nullptr;  // optional [B][N/2]: best b match of every a token
    bf16_t* vrows_out = nullptr;  // optional [B][N - r][C]: merged values row-major (the backward pass reads rows)
    int* dstlist_out = nullptr;   // optional [B][N/2]: entry k < r = b token the k-th ranked a token was merged into
};
size_t tome_workspace_bytes(int B, int N, int C);
int tome_effective_r(int N, int r);       // min(r, N / 2), 0 when N < 2
int launch_tome_merge(hipStream_t st, const TomeParams& p);
// adjoint of the merge for one merged tensor: dy [B][N - r][C] -> dx [B][N][ldx] (every original token receives the gradient
// of the row it went into, divided by that row's token count); order / dstlist from launch_tome_merge, inv: int scratch [B][N/2]
int launch_tome_unmerge(hipStream_t st, const bf16_t* dy, int B, int N, int C, int r, const int* order, const int* dstlist,
                        int* inv, bf16_t* dx, int ldx);
