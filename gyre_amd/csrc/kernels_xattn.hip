// Cross-attention of a BasicTransformerBlock as ONE kernel per block of BM rows (round 6):
//
//   out[m][:] = ( softmax_s( Q[m] . K[b][s] ) V[b][s] ) Wo^T + bo + x[m][:]        Q = LayerNorm(x) Wq^T,  b = sample of row m
//
// i.e. attn2.to_q (with norm2 folded in, as the three-launch chain folds it) -> attention over the Nk = 77 k cached text keys ->
// attn2.to_out + bias + residual, with the row statistics the next folded LayerNorm (norm3 -> GEGLU FF1) needs.  The three
// launches it replaces (k_gemm8<..., LNF>, k_attn2, k_gemm8<..., RS>) move x, Q, Q, O, O, x, out through HBM - 294 MB at the
// 64x64 level of SD1.5 at batch 16 for 84 MB that must move (x in, out out) - and each of them is a short-K streaming launch that
// runs at half the achievable HBM rate (five K steps between a cold prologue and a store tail): 104 us for the chain, measured
// launch by launch (profiles/r05_seq_b16.txt).  Here Q and O never leave the CU:
//
//   LDS (exactly 160 KB): XO = BM rows x 2C bytes (x, then Q, then O, XOR-swizzled 16-byte slots like every operand tile of this
//   library) + a two-stage ring of weight K steps (C rows x 128 B), reused by the per-head K / V^T tiles of the attention phase.
//   phase 0  x block -> XO by LDS-DMA (source-side swizzle); per-row LayerNorm statistics from the producer's partial sums
//   phase 1  Q = x W'q^T: 8 waves as 2 x 4 (BM = 128) on the BM x C tile, v_mfma_f32_16x16x32, weights streamed through the ring,
//            activations read from XO; folded-LayerNorm epilogue (same arithmetic as k_gemm8's LNF form: bit-identical Q) -> XO
//   phase 2  per head: K_h (Nk x D, prescaled by log2e / sqrt(D) at repack time) and V_h^T staged once per workgroup; a wave owns
//            16 query rows: S^T = K_h Q_h^T so the softmax axis is lane-local (+ two shuffles), ONE exact pass (Nk <= 80: no
//            running maximum), P feeds O^T = V_h^T P^T without leaving registers (the key order of the P registers is mirrored on
//            the V^T fragment reads), O_h overwrites Q_h in XO
//   phase 3  out = O Wo^T + bo + x: the same GEMM with XO as the activation operand; rounded rows staged through XO, 16-byte
//            coalesced stores, per-row (sum, sum of squares) of the rounded outputs for the consumer's LayerNorm
//
// Measured (SD1.5, batch 16, 64x64 level: M = 65536 rows, 512 workgroups = two rounds on 256 CUs): 74 - 82 us per launch against
// 104 us for the chain; UNet call 16.81 -> 16.68 ms.  Per workgroup (cycle stamps, tools/r06_xattn_phases.py): x block + LayerNorm
// partials 11 k cycles (80 KB per CU from the Infinity Cache: ~6 k at its 30 GB/s per CU), to_q loop 11.6 k (6.4 k of MFMA issue),
// attention 37 k - 2200 cycles per (16 rows, head) unit for 300 cycles of MFMA, the rate of the stand-alone k_attn2: this phase is
// bound by the length of its dependency chain at two waves per SIMD, two heads per iteration or one - to_out loop 9.5 k, epilogue 6 k.
// What the fusion removes is the memory passes of the two projections, not the attention's own time.
//
// Replaces the CrossAttention module call of diffusers' BasicTransformerBlock (attn2) inside the UNet the reference calls at
// gyre/pipeline/unet/core.py:262-274 (text conditioning: `encoder_hidden_states`), for the shape of SD1.x's 64x64 level (C = 320, 8 heads,
// one 77-token text chunk, batch >= 8); everything else keeps the three launches.
#include "gemm_shared.h"
#include <atomic>

namespace {
__device__ __forceinline__ void xa_dma16(unsigned dst, const void* src) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
}
typedef __attribute__((ext_vector_type(4))) unsigned xa_u32x4;
}  // namespace

// C channels, D = C / 8 head dim, BM rows per workgroup (BM * C = 40960: 128 rows at C = 320, 64 at C = 640), 8 waves as 2 x 4.
// The two projections run in NP = C / 320 passes of 320 output columns each (so that a weight K step is always 320 rows x 128 B and
// two of them fit beside XO); a pass's results wait in registers (packed) until the last pass has read its operand out of XO.
template <int C, int D, int BM>
__global__ __launch_bounds__(512) void k_xattn(XattnParams p, unsigned long long* stamps) {
#define XA_STAMP(i_) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (i_)] = __builtin_readcyclecounter(); } while (0)
    constexpr int H = C / D;
    constexpr int NH = 320, NP = C / NH;                   // output columns per pass, passes
    constexpr int WM = 2, WN = 4;
    constexpr int TM = BM / WM, TN = NH / WN, MI = TM / 16, NI = TN / 16;
    constexpr int ROWB = 2 * C, ROWS = ROWB / 16;          // XO row: bytes, 16-byte slots
    constexpr int XO_BYTES = BM * ROWB, STAGE = NH * 128;  // weight K step: 320 rows x 128 B
    constexpr int NKC = C / 64;                            // K steps of a pass
    constexpr int KSQ = (D + 31) / 32;                     // k32 steps of Q K^T
    constexpr int NKEY = 80, NKF = NKEY / 16;              // keys a staged head holds (Nk <= 80: one 77-token text chunk), 16-key fragments
    constexpr int DV = (D + 15) / 16;                      // 16-row fragments of V_h^T
    constexpr int KROW = DV * 32 + 16, VROW = NKEY * 2 + 16;  // K_h row (DV * 16 dims, +16 B: odd slot count), V_h^T row
    constexpr int HEAD_BYTES = NKEY * KROW + DV * 16 * VROW;
    constexpr bool DBUF = 4 * HEAD_BYTES <= 2 * STAGE;     // room for two pairs of head tiles: the next pair lands while this one is read
    constexpr int REGION = DBUF ? STAGE : 0;
    static_assert(C % NH == 0 && H == 8 && TM % 16 == 0 && TN % 16 == 0 && XO_BYTES + 2 * STAGE <= 160 * 1024 && 2 * HEAD_BYTES <= 2 * STAGE &&
                  (!DBUF || 2 * HEAD_BYTES <= STAGE), "tile / LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xo = smem;
    char* ring = smem + XO_BYTES;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int b = m0 / p.rows_per_sample;

    // ---- phase 0: x block -> XO (swizzled), first weight step of to_q -> ring ------------------------------------------------------
    {
        // a wave instruction fills 1 KiB of XO lane-linearly: lane L of instruction t -> byte (t * 8 + wave) * 1024 + L * 16
        constexpr int NT = XO_BYTES / 8192;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned off = (unsigned)((t * 8 + wave) * 1024 + lane * 16);
            const unsigned row = off / ROWB, ps = (off - row * ROWB) >> 4;
            const unsigned ls = (ps & ~7u) | ((ps & 7u) ^ (row & 7u));
            xa_dma16(lds0 + (unsigned)((t * 8 + wave) * 1024), p.x + (size_t)(m0 + row) * p.ldx + ls * 8);     // (M0 = the wave's base: the lane offset is implicit)
        }
    }
    XA_STAMP(0);
    const int r0 = tid >> 3;                       // row of this lane inside a 64-row staging granule
    const int kvs = (tid & 7) ^ (r0 & 7);          // global 16-byte k-vector this lane fetches into slot (tid & 7)
    // K step kc of the weight rows of pass np -> ring slot
    auto issue_w = [&](const bf16_t* W, int np, int kc, int slot) {
        const unsigned dst = lds0 + XO_BYTES + slot * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NH / 64; ++i)
            xa_dma16(dst + i * 8192, W + (size_t)(np * NH + r0 + 64 * i) * C + kc * 64 + kvs * 8);
    };
    issue_w(p.wq, 0, 0, 0);

    // per-row LayerNorm statistics of the rows this lane's accumulators belong to (k_gemm8's LNF arithmetic)
    float lrstd[MI], lrmu[MI];
    {
        const float invk = 1.0f / (float)C;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * TM + i * 16 + fr;
            float2 rs;
            if (p.ln_nparts > 0) {
                // every partial of the row requested before the first add (a dependent load per partial cost ~4 us per workgroup)
                float2 v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) if (t < p.ln_nparts) v[t] = ((const float2*)p.ln_parts)[(size_t)t * p.M + m];
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) if (t < p.ln_nparts) { su += v[t].x; sq += v[t].y; }
                for (int t = 8; t < p.ln_nparts; ++t) {
                    const float2 w2 = ((const float2*)p.ln_parts)[(size_t)t * p.M + m];
                    su += w2.x; sq += w2.y;
                }
                const float mean = su * invk;
                const float rstd = 1.0f / sqrtf(fmaxf(sq * invk - mean * mean, 0.f) + p.ln_eps);
                rs = make_float2(rstd, rstd * mean);
            } else {
                rs = ((const float2*)p.ln_stats)[m];
            }
            lrstd[i] = rs.x; lrmu[i] = rs.y;
        }
    }

    f32x4_t acc[MI][NI];
    // One projection: for every pass np, acc = XO[BM][C] * W[np * 320 .. + 320][C]^T, then `finish(np)`.  Weights through the two-stage
    // ring; the first K step of pass 0 has been requested into slot `first` by the caller; `pre(np)` runs in front of a pass's K loop
    // (the place for requests that should be in flight under it).
    auto projection = [&](const bf16_t* W, int first, auto&& pre, auto&& finish) {
        int slot = first;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int np = 0; np < NP; ++np) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            pre(np);
            for (int kc = 0; kc < NKC; ++kc) {
                if (kc + 1 < NKC) issue_w(W, np, kc + 1, slot ^ 1);
                else if (np + 1 < NP) issue_w(W, np + 1, 0, slot ^ 1);
                const uint4* a = (const uint4*)xo;
                const uint4* bw = (const uint4*)(ring + slot * STAGE);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8_t af[MI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int r = wm * TM + i * 16 + fr;
                        af[i] = __builtin_bit_cast(bf16x8_t, a[r * ROWS + kc * 8 + ((ks * 4 + fq) ^ (r & 7))]);
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int r = wn * TN + j * 16 + fr;
                        const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, bw[r * 8 + ((ks * 4 + fq) ^ (r & 7))]);
#pragma unroll
                        for (int i = 0; i < MI; ++i) acc[i][j] = GYRE_MFMA_16x16x32(bf, af[i], acc[i][j], 0, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                slot ^= 1;
            }
            finish(np);
        }
    };
    // a lane's 4 consecutive channels of row r (packed) -> XO (swizzled)
    auto xo_store4 = [&](int r, int n, uint2 v) {
        const int ls = n >> 3, ps = (ls & ~7) | ((ls & 7) ^ (r & 7));
        *(uint2*)(xo + r * ROWB + ps * 16 + (n & 7) * 2) = v;
    };
    uint2 pk[NP][MI][NI];                                  // a projection's results, packed, until its last pass is done with XO
    auto flush_pk = [&]() {
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) xo_store4(wm * TM + i * 16 + fr, np * NH + wn * TN + j * 16 + 4 * fq, pk[np][i][j]);
    };

    // ---- phase 1: Q = LayerNorm(x) Wq^T -> XO ---------------------------------------------------------------------------------------------
    XA_STAMP(1);
    float4 ccs[NI], cbb[NI];                               // column constants of the pass's fragments, in flight under its K loop
    projection(p.wq, 0,
        [&](int np) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = np * NH + wn * TN + j * 16 + 4 * fq;
                ccs[j] = *(const float4*)(p.q_colsum + n); cbb[j] = *(const float4*)(p.q_bias + n);
            }
        },
        [&](int np) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float4 cs = ccs[j], bb = cbb[j];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    pk[np][i][j] = make_uint2(pack_bf16x2(fmaf(acc[i][j][0], lrstd[i], fmaf(-lrmu[i], cs.x, bb.x)), fmaf(acc[i][j][1], lrstd[i], fmaf(-lrmu[i], cs.y, bb.y))),
                                              pack_bf16x2(fmaf(acc[i][j][2], lrstd[i], fmaf(-lrmu[i], cs.z, bb.z)), fmaf(acc[i][j][3], lrstd[i], fmaf(-lrmu[i], cs.w, bb.w))));
            }
        });
    XA_STAMP(2);
    flush_pk();                                            // (the last barrier of the projection: every wave is done reading x from XO)

    // ---- phase 2: attention, two heads per iteration -----------------------------------------------------------------------------------------
    // The pair of heads of iteration p + 1 is fetched into registers BEFORE pair p is computed and written to the ring AFTER it, so the
    // fetch latency hides under the pair's work: K_h rows [NKEY][KROW] (keys >= Nk and dims >= D zero), V_h^T rows [DV * 16][VROW]
    // (keys >= Nk zero: the cached V^T is padded with unspecified values and 0 * NaN is NaN).  D = 40: two pairs fit the ring, the
    // next one is written while this one is read (one barrier per pair); D = 80: one pair, written between two barriers.
    // A wave owns one 16-row fragment: at BM = 128 (8 fragments) for both heads of a pair, at BM = 64 (4 fragments) for one of them.
    constexpr int RF = BM / 16, HG = 8 / RF;               // row fragments; head groups among the waves (1 or 2)
    static_assert(RF * HG == 8 && (HG == 1 || HG == 2) && H % 2 == 0, "8 waves = row fragments x head groups");
    constexpr int KV = DV * 2;                             // 16-byte vectors per staged K row
    constexpr int VV = NKEY / 8;                           // 16-byte vectors per staged V^T row
    constexpr int NQK = (NKEY * KV + 511) / 512, NQV = (DV * 16 * VV + 511) / 512;     // vectors per thread, head and operand
    // per-thread fetch descriptors (head 0; head h adds h * D channels / rows)
    const bf16_t* k_src[NQK]; int k_dst[NQK]; bool k_live[NQK], k_load[NQK];
    const bf16_t* v_src[NQV]; int v_dst[NQV]; bool v_live[NQV], v_load[NQV];
    unsigned vmask[NQV][4];                                // keys of a V^T vector at or past Nk are cleared
#pragma unroll
    for (int q = 0; q < NQK; ++q) {
        const int v = tid + 512 * q, sk = v / KV, c8 = v - sk * KV;       // key, 8-dim vector
        k_live[q] = v < NKEY * KV; k_load[q] = k_live[q] && sk < p.Nk && c8 * 8 < D;
        k_src[q] = p.k + ((size_t)b * p.Nk + (k_load[q] ? sk : 0)) * C + c8 * 8;
        k_dst[q] = sk * KROW + c8 * 16;
    }
#pragma unroll
    for (int q = 0; q < NQV; ++q) {
        const int v = tid + 512 * q, d = v / VV, s8 = v - d * VV;          // dim row, 8-key vector
        v_live[q] = v < DV * 16 * VV; v_load[q] = v_live[q] && d < D && s8 * 8 < p.ldvt;
        v_src[q] = p.vt + ((size_t)b * C + (v_load[q] ? d : 0)) * p.ldvt + s8 * 8;
        v_dst[q] = NKEY * KROW + d * VROW + s8 * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nvalid = p.Nk - s8 * 8;
            vmask[q][e] = 2 * e >= nvalid ? 0u : (2 * e + 1 >= nvalid ? 0xffffu : 0xffffffffu);
        }
    }
    xa_u32x4 hk[2][NQK], hv[2][NQV];
    auto fetch_pair = [&](int pr) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int h = 2 * pr + e;
#pragma unroll
            for (int q = 0; q < NQK; ++q) {
                hk[e][q] = xa_u32x4{0u, 0u, 0u, 0u};
                if (k_load[q]) hk[e][q] = *(const xa_u32x4*)(k_src[q] + h * D);
            }
#pragma unroll
            for (int q = 0; q < NQV; ++q) {
                hv[e][q] = xa_u32x4{0u, 0u, 0u, 0u};
                if (v_load[q]) hv[e][q] = *(const xa_u32x4*)(v_src[q] + (size_t)h * D * p.ldvt);
            }
        }
    };
    auto store_pair = [&](int pr) {
        char* base = ring + (pr & 1) * REGION;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int q = 0; q < NQK; ++q)
                if (k_live[q]) *(xa_u32x4*)(base + e * HEAD_BYTES + k_dst[q]) = hk[e][q];
#pragma unroll
            for (int q = 0; q < NQV; ++q)
                if (v_live[q]) *(xa_u32x4*)(base + e * HEAD_BYTES + v_dst[q]) =
                    xa_u32x4{hv[e][q].x & vmask[q][0], hv[e][q].y & vmask[q][1], hv[e][q].z & vmask[q][2], hv[e][q].w & vmask[q][3]};
        }
    };
    const int qr = (wave % RF) * 16 + fr;                  // this lane's query row inside the block
    const int hg = wave / RF;                              // which head of a pair this wave takes (HG == 2)
    // one (16-row fragment, head) unit out of the staged tiles at kb / vb
    auto attend = [&](int h, const char* kb, const char* vb) {
        // Q_h fragments: lane (query fr, dims ks * 32 + fq * 8 .. + 8); dims past D are zeroed (they belong to the next head)
        bf16x8_t qf[KSQ];
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) {
            const int ls = (h * D + ks * 32 + fq * 8) >> 3;
            xa_u32x4 raw = {0u, 0u, 0u, 0u};
            if (ks * 32 + fq * 8 < D) raw = *(const xa_u32x4*)(xo + qr * ROWB + (((ls & ~7) | ((ls & 7) ^ (qr & 7))) << 4));
            qf[ks] = __builtin_bit_cast(bf16x8_t, raw);
        }
        // S^T = K_h Q_h^T: NKF 16-key fragments; lane: query fr, keys 16 jk + 4 fq + e.  Dims past D carry zeros on BOTH sides (the K row ends
        // at DV * 16 dims: a read behind it sees the next row / the pad, replaced by zero here; the Q side is zeroed above)
        f32x4_t s[NKF];
#pragma unroll
        for (int jk = 0; jk < NKF; ++jk) {
            s[jk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                xa_u32x4 kraw = *(const xa_u32x4*)(kb + (jk * 16 + fr) * KROW + ks * 64 + fq * 16);
                if (ks * 32 + fq * 8 >= DV * 16) kraw = xa_u32x4{0u, 0u, 0u, 0u};
                s[jk] = GYRE_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, kraw), qf[ks], s[jk], 0, 0, 0);
            }
        }
        float mx = -1e30f;
#pragma unroll
        for (int jk = 0; jk < NKF; ++jk) {
            if ((jk + 1) * 16 > p.Nk) {                    // (wave-uniform) only a fragment that straddles Nk is masked
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (jk * 16 + 4 * fq + e >= p.Nk) s[jk][e] = -1e30f;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(s[jk][0], s[jk][1]), fmaxf(s[jk][2], s[jk][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float l = 0.f;
#pragma unroll
        for (int jk = 0; jk < NKF; ++jk)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[jk][e] = __builtin_amdgcn_exp2f(s[jk][e] - mx); l += s[jk][e]; }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        // O^T = V_h^T P^T: k32 step t covers the key fragments 2t, 2t + 1; the lane's eight k slots are keys
        // 32 t + 4 fq + {0..3} and 32 t + 16 + 4 fq + {0..3} - the V^T fragment is read in the same order (two 8-byte reads)
        f32x4_t o[DV];
#pragma unroll
        for (int di = 0; di < DV; ++di) o[di] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (NKF + 1) / 2; ++t) {
            constexpr int nkf_c = NKF;
            const bool second = 2 * t + 1 < nkf_c;         // (the last k32 step of an odd fragment count holds one fragment: zeros in the other half)
            const int t1 = second ? 2 * t + 1 : 0;
            xa_u32x4 pw;
            pw.x = pack_bf16x2(s[2 * t][0], s[2 * t][1]); pw.y = pack_bf16x2(s[2 * t][2], s[2 * t][3]);
            pw.z = second ? pack_bf16x2(s[t1][0], s[t1][1]) : 0u;
            pw.w = second ? pack_bf16x2(s[t1][2], s[t1][3]) : 0u;
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
            for (int di = 0; di < DV; ++di) {
                const char* vrow = vb + (di * 16 + fr) * VROW + (32 * t + 4 * fq) * 2;
                const uint2 v0 = *(const uint2*)vrow;
                uint2 v1 = make_uint2(0u, 0u);
                if (second) v1 = *(const uint2*)(vrow + 32);
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, xa_u32x4{v0.x, v0.y, v1.x, v1.y});
                o[di] = GYRE_MFMA_16x16x32(vf, pf, o[di], 0, 0, 0);
            }
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int di = 0; di < DV; ++di) {
            const int d = di * 16 + 4 * fq;                // lane: query fr, dims d .. d + 3
            if (d < D) xo_store4(qr, h * D + d, make_uint2(pack_bf16x2(o[di][0] * inv, o[di][1] * inv), pack_bf16x2(o[di][2] * inv, o[di][3] * inv)));
        }
    };
    fetch_pair(0);
    __syncthreads();                                       // Q complete in XO; the ring is free
    XA_STAMP(3);
    // in flight under the whole attention phase (the projection's accumulators are dead there): the residual values and the bias of this
    // lane's accumulator positions; to_out's first weight step follows once a ring slot is free
    uint2 res[NP][MI][NI];
#pragma unroll
    for (int np = 0; np < NP; ++np)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = np * NH + wn * TN + j * 16 + 4 * fq;
#pragma unroll
            for (int i = 0; i < MI; ++i) res[np][i][j] = *(const uint2*)(p.x + (size_t)(m0 + wm * TM + i * 16 + fr) * p.ldx + n);
        }
    store_pair(0);
    constexpr int NPAIR = H / 2;
    bool wo_requested = false;
    for (int pr = 0; pr < NPAIR; ++pr) {
        __syncthreads();                                   // pair `pr` is staged (and, with two regions, every wave is done with the other one)
        if (pr + 1 < NPAIR) fetch_pair(pr + 1);
        else if (DBUF && (pr & 1) == 1) { issue_w(p.wo, 0, 0, 0); wo_requested = true; }   // last pair sits in region 1: slot 0 is free now
        const char* base = ring + (pr & 1) * REGION;
        if (HG == 1) {
            attend(2 * pr, base, base + NKEY * KROW);
            attend(2 * pr + 1, base + HEAD_BYTES, base + HEAD_BYTES + NKEY * KROW);
        } else {
            attend(2 * pr + hg, base + hg * HEAD_BYTES, base + hg * HEAD_BYTES + NKEY * KROW);
        }
        if (pr + 1 < NPAIR) {
            if (!DBUF) __syncthreads();                    // one region: every wave is done reading pair `pr` before it is overwritten
            store_pair(pr + 1);
        }
    }
    __syncthreads();                                       // O complete in XO; the ring is free for to_out's weights

    // ---- phase 3: out = O Wo^T + bo + x -------------------------------------------------------------------------------------------------------
    XA_STAMP(4);
    if (!wo_requested) issue_w(p.wo, 0, 0, 0);
    projection(p.wo, 0,
        [&](int np) {                                      // the pass's bias values, in flight under its K loop
#pragma unroll
            for (int j = 0; j < NI; ++j)
                cbb[j] = p.bo ? *(const float4*)(p.bo + np * NH + wn * TN + j * 16 + 4 * fq) : make_float4(0.f, 0.f, 0.f, 0.f);
        },
        [&](int np) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const float4 bb = cbb[j];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    pk[np][i][j] = make_uint2(pack_bf16x2(acc[i][j][0] + bb.x + bf16lo(res[np][i][j].x), acc[i][j][1] + bb.y + bf16hi(res[np][i][j].x)),
                                              pack_bf16x2(acc[i][j][2] + bb.z + bf16lo(res[np][i][j].y), acc[i][j][3] + bb.w + bf16hi(res[np][i][j].y)));
            }
        });
    XA_STAMP(5);
    flush_pk();
    __syncthreads();
    // rounded rows out: consecutive lanes store consecutive 16-byte slots of a row
    for (int v = tid; v < BM * ROWS; v += 512) {
        const int r = v / ROWS, ls = v - r * ROWS;
        const int ps = (ls & ~7) | ((ls & 7) ^ (r & 7));
        *(uint4*)(p.out + (size_t)(m0 + r) * p.ldo + ls * 8) = *(const uint4*)(xo + r * ROWB + ps * 16);
    }
    XA_STAMP(6);
    if (p.rowstat_out) {
        // four lanes per row, C / 4 channels each (any slot order: a sum), folded with two shuffles
        constexpr int VPL = ROWS / 4;
        for (int rr = tid >> 2; rr < BM; rr += 128) {
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < VPL; ++c) {
                float f[8];
                unpack8(*(const uint4*)(xo + rr * ROWB + ((tid & 3) * VPL + c) * 16), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { su += f[e]; sq = fmaf(f[e], f[e], sq); }
            }
            su += __shfl_xor(su, 1); sq += __shfl_xor(sq, 1);
            su += __shfl_xor(su, 2); sq += __shfl_xor(sq, 2);
            if ((tid & 3) == 0) ((float2*)p.rowstat_out)[m0 + rr] = make_float2(su, sq);
        }
    }
    XA_STAMP(7);
#undef XA_STAMP
}

// tuning: per-workgroup cycle stamps of the phase boundaries (8 x uint64 per workgroup; tools/r06_xattn_phases.py)
static thread_local unsigned long long* g_xattn_stamps = nullptr;
extern "C" int gyre_debug_xattn_stamps(void* dev_buf) { g_xattn_stamps = (unsigned long long*)dev_buf; return 0; }

// Shapes the fused kernel is TAKEN for: SD1.x's 64x64 level (C = 320, 8 heads, 128-row blocks that do not straddle a sample), one text
// chunk (Nk <= 80), and a grid of at least one workgroup per CU (M >= 256 * 128: batch >= 8 at 64x64).  Measured and left to the
// three-launch chain: (i) small grids - at batch 2 the 64 workgroups of the 64x64 level leave three quarters of the chip idle (UNet
// call 5.55 -> 5.73 ms with the fused kernel everywhere); (ii) the C = 640 instantiation of this template (32x32 level, BM = 64, two
// column passes): its projections are 20 K steps of 640 MFMA cycles each behind a two-stage ring - LDS has no room for a third
// stage beside XO - so every step exposes most of a request round trip: ~75 us per launch against 71 us for the chain.
bool xattn_supports(int C, int heads, int Nq, int Nk, int M) {
    if (heads != 8 || Nk < 1 || Nk > 80 || M <= 0) return false;
    if (C == 320) return Nq % 128 == 0 && M % 128 == 0 && M / 128 >= 256;
    return false;
}

template <int C, int D, int BM>
static int xattn_go(hipStream_t st, const XattnParams& p) {
    const int lds = 160 * 1024;
    auto kern = k_xattn<C, D, BM>;
    static std::atomic<unsigned long long> attr_done{0};
    if (gyre_lds_attr_needed(attr_done)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(p.M / BM), dim3(512), lds, st, p, g_xattn_stamps);
    GYRE_LAUNCH_CHECK();
    return 0;
}

int launch_xattn(hipStream_t st, const XattnParams& p, int C) {
    // (the grid-size part of xattn_supports is the CALLER's planning rule - under batch-invariant planning a sub-batch runs here on a
    //  small grid; the kernel itself needs whole 128-row blocks inside a sample)
    if (!xattn_supports(C, p.heads, p.rows_per_sample, p.Nk, 256 * 128) || p.M <= 0 || p.M % 128)
        GYRE_FAIL(-6, "xattn: shape outside the fused cross-attention kernel's domain");
    if ((p.ldx % 8) || (p.ldo % 8) || (p.ldvt % 8) || ((((size_t)p.x | (size_t)p.out | (size_t)p.k | (size_t)p.vt | (size_t)p.wq | (size_t)p.wo) & 15) != 0))
        GYRE_FAIL(-1, "xattn: operands must be 16-byte aligned with strides that are multiples of 8 elements");
    // algorithmic work: two C x C projections + the two attention products; bytes: x in (twice: operand + residual), out, weights, K / V
    const double fl = 2.0 * 2.0 * p.M * (double)C * C + 4.0 * p.M * (double)p.Nk * C;
    const double by = 3.0 * p.M * C * 2.0 + 2.0 * C * (double)C * 2.0;
    GyreProfScope prof_(KC_XATTN, st, fl, by);
    return xattn_go<320, 40, 128>(st, p);
}
