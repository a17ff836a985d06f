// "W-resident" bf16 MFMA GEMM for the square projections of the Transformer2D blocks (K = 320, N a multiple of 320:
// to_out, to_q of the cross-attention, proj_in / proj_out at the 64x64 and 32x32 UNet levels):
//
//   C[M][N] = epilogue( A[M][K] * W[N][K]^T )          same contract as k_gemm8 (linear mode, bf16 row-major out)
//
// Why another kernel.  These launches move 84 - 126 MB for 13 GFLOP: at M = 65536, N = K = 320 the 256x320 tiling is ONE round
// of 256 workgroups whose load / multiply / store phases run in lockstep across the chip (HBM busy, then idle, then busy):
// 33 - 37 us against 10.5 - 15.7 us of HBM time.  Here the weights never move: a workgroup of FOUR waves (one per SIMD, 512
// registers each) keeps a 320-column panel of W in REGISTERS for its whole life - wave w holds 80 columns x K as the MFMA's
// first operand, CB * K / 32 fragments of 4 registers (200 registers at K = 320, 400 at K = 640) - and streams its rows of A
// through a three-stage LDS ring of 40 KB tiles (64 / 32 rows).  Loads of tile t+2, MFMAs of tile t and the stores of tile
// t-1 overlap inside every workgroup, so the launch is a continuous stream instead of three phases.
//
//   * operands swapped (W = MFMA A, activations = MFMA B, v_mfma_f32_16x16x32_bf16): a lane owns ONE output row per 16-row
//     block and - the packed weight rows are permuted for it (k_wr_pack) - 4 * CB CONSECUTIVE output columns, 40 bytes at
//     CB = 5: bias / residual / rounding / row statistics need no cross-lane traffic and leave as CB 8-byte stores;
//   * every wave reads every A fragment (one ds_read_b128 per CB MFMAs: 80 MFMA cycles per LDS read at CB = 5);
//   * A tiles arrive by LDS-DMA in 1 KB pieces of 8 rows x 128 B (whole 128-byte lines of global memory); inside a piece the
//     16-byte chunks are placed so that the fragment reads (16 rows x 4 chunks per instruction) are conflict-free: chunk c of
//     row r of half h (rows 8h .. 8h+7 of a 16-row block) sits at ((c ^ h) * 8 + r) * 16;
//   * residual rows are requested by hand one tile ahead (8-byte loads into registers); the ring and they share one counted
//     s_waitcnt per tile (loads return in order; stores in flight can only make the wait stricter).  No spills allowed.
//
// K is summed in ascending order on one accumulator in steps of 32, like the 8-wave tile kernel: results are bit-identical to it.
//
// STATUS: correct, measured, and NOT the planner's choice (tuning bit 13 turns it on).  Per cold launch at M = 65536 it takes
// 30.8 us against the 256x320 tiling's 34.4 (plain), 41.0 against 44.8 (+ residual), 32.6 against 28.9 (folded LayerNorm, whose
// per-tile row statistics cost 32 extra loads); inside the UNet the 16 launches it takes over change nothing (17.11 against
// 17.09 ms).  Ablations (tools/wr_bench.py, ABL=): loads alone - W, the ring, residual - take 24 us of the 41: four tiles per
// workgroup are too short a stream to hide the 200 KB weight read in front of them, and what one launch can reach is the ~4.8
// TB/s a copy reaches on this part (26 us for the 126 MB of the residual form), not the 15.7 us of the 8 TB/s figure.
// Replaces the cuBLAS GEMMs behind torch.nn.Linear / the 1x1 convs in the third-party UNet the reference calls at
// gyre/pipeline/unet/core.py:274 (BasicTransformerBlock to_q / to_out, Transformer2DModel proj_in / proj_out).
#include "gemm_shared.h"
#include <atomic>
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

namespace {
template <int V> using ic = std::integral_constant<int, V>;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(ic<I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void wr_glds(unsigned lds_addr, const void* vptr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(vptr) : "memory");
}
}  // namespace

// W[N][K] (row-major bf16) -> fragment order of k_gemm_wr: panel P (64 * CB columns), wave w, column block b, K step kk, lane l
// holds the 8 values W[n][32 * kk + 8 * (l >> 4) ...] of n = P * 64 * CB + w * 16 * CB + ((l & 15) >> 2) * 4 * CB + 4 * b + (l & 3):
// row (l & 15) of the MFMA's A operand is THAT column, so that accumulator register i of lane group g = l >> 4 is column
// g * 4 * CB + 4 * b + i of the wave's 16 * CB - consecutive over (b, i)
__global__ __launch_bounds__(256) void k_wr_pack(const bf16_t* W, int N, int K, int CB, uint4* out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int KS = K / 32;
    const size_t total = (size_t)N * K / 8;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    size_t f = idx >> 6;
    const int kk = (int)(f % KS); f /= KS;
    const int b = (int)(f % CB); f /= CB;
    const int w = (int)(f & 3), P = (int)(f >> 2);
    const int r = l & 15;
    const int n = P * 64 * CB + w * 16 * CB + (r >> 2) * 4 * CB + 4 * b + (r & 3);
    out[idx] = *(const uint4*)(W + (size_t)n * K + 32 * kk + 8 * (l >> 4));
}

// RES: + residual[m][n]; RS: per-row sum / sum of squares of the rounded outputs (GemmParams::rowstat_out, one partial per
// wave and panel: [4 * panels][M][2]); LNF: folded LayerNorm (GemmParams::ln_colsum, ln_stats / ln_parts)
// ABL (timing ablations, garbage results): 1 = no epilogue arithmetic / stores, 2 = no MFMAs, 4 = no ring requests
template <int K, int CB, bool RES, bool RS, bool LNF, int ABL = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_wr(GemmParams p, const uint4* wpk, int rows_per_wg, int npanels) {
    constexpr int KS = K / 32, KQ = K / 64;
    constexpr int R = 20480 / K;                // rows per ring stage (40 KB): 64 / 32
    constexpr int RB = R / 16;                  // 16-row blocks per stage: 4 / 2
    constexpr int WPR = 4 / RB;                 // waves that share the requests of one row block: 1 / 2
    constexpr int KQW = KQ / WPR;               // 128-byte column chunks per wave and row block: 5
    constexpr int STAGE = R * K * 2;
    constexpr int NS = 3;
    constexpr int NC = 4 * CB;                  // consecutive output columns per lane
    constexpr int NPL = 8;                      // LNF: row-statistics partials read per row (GemmParams::ln_nparts <= NPL)
    constexpr int NQ = RES ? RB * 3 : 1;        // 16-byte residual loads per lane and tile (160 B per row and wave, row-contiguous)
    constexpr int NL = LNF ? RB * NPL : 1;      // 8-byte row-statistics loads per lane and tile
    constexpr int NRES = RES ? RB * 3 : LNF ? RB * NPL : 0;     // side loads per lane and tile
    constexpr int PROW = 176;                   // bytes per row of a wave's epilogue patch (160 + 16: conflict-free 8-byte lane accesses)
    constexpr int PATCH = 16 * PROW;
    static_assert(CB == 5, "the epilogue patch is laid out for 80 columns per wave");
    constexpr int NDMA = 2 * KQW;               // LDS-DMA requests per wave and tile
    static_assert(STAGE == 40960 && KQ % WPR == 0 && 20480 % K == 0 && R % 16 == 0, "ring geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int panel = blockIdx.x % npanels, rg = blockIdx.x / npanels;
    const int m_begin = rg * rows_per_wg;
    const int m_end = min(p.M, m_begin + rows_per_wg);
    const int ntile = (m_end - m_begin + R - 1) / R;          // >= 1 (launcher)
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;

    // ---- this wave's weight slice -> registers (the only time W is read) ---------------------------------------------------------
    bf16x8_t Wf[CB][KS];
    {
        const uint4* wsrc = wpk + ((size_t)(panel * 4 + w) * CB * KS) * 64 + lane;
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) Wf[b][kk] = __builtin_bit_cast(bf16x8_t, wsrc[(size_t)(b * KS + kk) * 64]);
    }
    const int ncol = panel * 64 * CB + w * 16 * CB + g * NC;      // first of this lane's NC output columns
    float cb[NC], cc[LNF ? NC : 1];
#pragma unroll
    for (int i = 0; i < NC; i += 4) {
        const float4 v = p.bias ? *(const float4*)(p.bias + ncol + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        cb[i] = v.x; cb[i + 1] = v.y; cb[i + 2] = v.z; cb[i + 3] = v.w;
        if constexpr (LNF) {
            const float4 c = *(const float4*)(p.ln_colsum + ncol + i);
            cc[i] = c.x; cc[i + 1] = c.y; cc[i + 2] = c.z; cc[i + 3] = c.w;
        }
    }

    // ---- ring requests: wave w fetches, of every tile, the rows of 16-row block w / WPR and KQW of its 128-byte column chunks,
    // as 2 * KQW pieces of 8 rows x 128 B.  Lane l of a piece (half h): row 8h + (l & 7), 16-byte chunk (l >> 3) ^ h.
    const int rb_w = w / WPR, kq0 = (w % WPR) * KQW;
    const int dr = lane & 7, ds = lane >> 3;
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        if constexpr (ABL & 4) return;
        const int m0 = m_begin + (t < ntile ? t : 0) * R + rb_w * 16;       // past-the-end requests re-read tile 0 (never consumed)
        const unsigned dst = lds0 + slot * STAGE + ((rb_w * KQ + kq0) * 2) * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = min(m0 + 8 * h + dr, p.M - 1);                  // rows past M re-read row M - 1 (never stored)
            const char* src = (const char*)(p.A + (size_t)row * p.lda) + kq0 * 128 + ((ds ^ h) * 16);
#pragma unroll
            for (int j = 0; j < KQW; ++j) wr_glds(dst + (2 * j + h) * 1024, src + j * 128);
        }
    };
    // fragment (rb, kk) of a stage: lane (r16, g) reads row r16 of block rb, K values 32 kk + 8 g ...: piece ((rb * KQ + kk / 2) * 2 + h),
    // chunk c = 4 * (kk & 1) + g at ((c ^ h) * 8 + (r16 & 7)) * 16, h = r16 >> 3
    const int fh = r16 >> 3;
    const unsigned rd_off = fh * 1024 + (((g ^ fh) * 8 + (r16 & 7)) * 16);

    // ---- residual rows, requested by hand one tile ahead (the compiler's own vmcnt bookkeeping must not see loads between the
    // ring's requests: it would drain them) ------------------------------------------------------------------------------------------
    struct Side { u32x4_t q[NQ]; u32x2_t l[NL]; };
    Side side[2];
    // row-contiguous view of a wave's 16 x 80 output block: 160 slots of 16 bytes (row = slot / 10, chunk = slot % 10), three per
    // lane (the third for lanes < 32 only; the others repeat slot 159)
    int prow[3], pch[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int idx = min(j * 64 + lane, 159); prow[j] = idx / 10; pch[j] = idx - prow[j] * 10; }
    const int ncolw = panel * 64 * CB + w * 16 * CB;               // first of the wave's 80 columns
    char* const patch = smem + NS * STAGE + w * PATCH;
    const int lo = r16 * PROW + g * (NC * 2);                      // this lane's 40 bytes of row r16 (lane view)
    auto res_issue = [&](int t, Side& sd) __attribute__((always_inline)) {
        const int m0 = m_begin + (t < ntile ? t : 0) * R;
        if constexpr (RES) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int m = min(m0 + rb * 16 + prow[j], p.M - 1);
                    const bf16_t* rp = p.residual + (size_t)m * p.ldr + ncolw + pch[j] * 8;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&a"(sd.q[rb * 3 + j]) : "v"(rp) : "memory");
                }
        } else if constexpr (LNF) {
            // (sum, sum of squares) partials the producing GEMM left, or the finished (rstd, rstd * mean): NPL loads per row either way
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = min(m0 + rb * 16 + r16, p.M - 1);
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const float* sp = p.ln_nparts > 0 ? p.ln_parts + ((size_t)(q < p.ln_nparts ? q : 0) * p.M + m) * 2 : p.ln_stats + (size_t)m * 2;
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=&a"(sd.l[rb * NPL + q]) : "v"(sp) : "memory");
                }
            }
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------------------
    res_issue(0, side[0]);
    issue(0, 0);
    issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    float2* const rs_out = RS ? (float2*)p.rowstat_out + (size_t)(panel * 4 + w) * p.M : nullptr;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};

    auto tile = [&](int t, int slot, Side& sd) __attribute__((always_inline)) {
        const int m0 = m_begin + t * R;
        const char* stage = smem + slot * STAGE + rd_off;
        float2 lst[RB];                              // LNF: (rstd, rstd * mean) of this lane's row in each block
        if constexpr (LNF) {
            const float invk = 1.0f / (float)p.K;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (p.ln_nparts > 0) {
                    float su = 0.f, sq = 0.f;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const bool on = q < p.ln_nparts;
                        su += on ? __uint_as_float(sd.l[rb * NPL + q][0]) : 0.f;
                        sq += on ? __uint_as_float(sd.l[rb * NPL + q][1]) : 0.f;
                    }
                    const float mean = su * invk;
                    const float rstd = 1.0f / sqrtf(fmaxf(sq * invk - mean * mean, 0.f) + p.ln_eps);
                    lst[rb] = make_float2(rstd, rstd * mean);
                } else {
                    lst[rb] = make_float2(__uint_as_float(sd.l[rb * NPL][0]), __uint_as_float(sd.l[rb * NPL][1]));
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x4_t acc[CB];
#pragma unroll
            for (int b = 0; b < CB; ++b) acc[b] = zero4;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const bf16x8_t bf = *(const bf16x8_t*)(stage + (rb * KQ + kk / 2) * 2048 + (kk & 1) * 512);
                if constexpr (!(ABL & 2)) {
#pragma unroll
                    for (int b = 0; b < CB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[b][kk], bf, acc[b], 0, 0, 0);
                } else {
                    acc[kk % CB][0] += (float)bf[0];
                }
            }
            if constexpr (ABL & 1) {
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < CB; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
                if (s == 12345.678f) ((float*)p.out)[0] = s;
                continue;
            }
            const int mb = m0 + rb * 16;                 // (wave-uniform: M and rows_per_wg are multiples of 16)
            if (mb >= m_end) continue;
            const int m = mb + r16;
            // residual: the row-contiguous 16-byte pieces requested a tile ago -> patch -> this lane's 40 bytes
            u32x2_t rv[CB];
            if constexpr (RES) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < 2 || lane < 32) *(u32x4_t*)(patch + prow[j] * PROW + pch[j] * 16) = sd.q[rb * 3 + j];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int b = 0; b < CB; ++b) rv[b] = *(const u32x2_t*)(patch + lo + 8 * b);
                __builtin_amdgcn_wave_barrier();
            }
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (LNF) v[i] = fmaf(acc[b][i], lst[rb].x, fmaf(-lst[rb].y, cc[4 * b + i], cb[4 * b + i]));
                    else v[i] = acc[b][i] + cb[4 * b + i];
                }
                if constexpr (RES) { v[0] += bf16lo(rv[b][0]); v[1] += bf16hi(rv[b][0]); v[2] += bf16lo(rv[b][1]); v[3] += bf16hi(rv[b][1]); }
                const uint2 o = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                *(uint2*)(patch + lo + 8 * b) = o;
                if constexpr (RS) {
                    const float f0 = bf16lo(o.x), f1 = bf16hi(o.x), f2 = bf16lo(o.y), f3 = bf16hi(o.y);
                    su += (f0 + f1) + (f2 + f3);
                    sq = fmaf(f0, f0, sq); sq = fmaf(f1, f1, sq); sq = fmaf(f2, f2, sq); sq = fmaf(f3, f3, sq);
                }
            }
            // rounded rows back out of the patch, 160 contiguous bytes per row and wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < 2 || lane < 32)
                    *(uint4*)((bf16_t*)p.out + (size_t)(mb + prow[j]) * p.ldc + ncolw + pch[j] * 8) = *(const uint4*)(patch + prow[j] * PROW + pch[j] * 16);
            __builtin_amdgcn_wave_barrier();
            if constexpr (RS) {
                su += __shfl_xor(su, 16); sq += __shfl_xor(sq, 16);
                su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
                if (g == 0) rs_out[m] = make_float2(su, sq);
            }
        }
    };

    // ---- main loop: top of tile t = "tile t landed for every wave, tile t-1's stage is free" -------------------------------------------
    // Requests per wave in issue order: ... RES(t) DMA(t+1) | RES(t+1) DMA(t+2) | ...  Before tile t runs, DMA(t) [requested two
    // tiles ago] and RES(t) must have landed: the loads younger than RES(t) are DMA(t+1), RES(t+1), DMA(t+2).
    constexpr int YOUNG = NDMA + NRES + NDMA;
    auto step = [&](int t, Side& cur, Side& nxt) __attribute__((always_inline)) {
        if (t > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES + NDMA) : "memory");       // DMA(t): younger loads = RES(t) + DMA(t+1)
            __builtin_amdgcn_s_barrier();
        }
        res_issue(t + 1, nxt);
        issue(t + 2, (t + 2) % NS);
        if constexpr (NRES > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) asm volatile("" : "+a"(cur.q[i]));
            } else {
#pragma unroll
                for (int i = 0; i < NL; ++i) asm volatile("" : "+a"(cur.l[i]));
            }
        }
        tile(t, t % NS, cur);
    };
    for (int t = 0; t < ntile; t += 2) {
        step(t, side[0], side[1]);
        if (t + 1 < ntile) step(t + 1, side[1], side[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // past-the-end requests: nothing may land in LDS after the wave exits
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
bool gemm_wr_supports(const GemmParams& p) {
    if (p.mode != GEMM_LINEAR || p.out_mode != OUT_BF16 || p.batch > 1 || p.geglu || p.vt_out) return false;
    if (p.K != 320) return false;          // (K = 640: 400 weight registers per wave - three of the five forms spill; not instantiated)
    if (p.A2 && p.A2 != p.A) return false;
    if (p.rowbias || p.colstat_out || p.w_sample_stride) return false;
    if (p.N % 320 || p.M % 16 || p.lda % 8 || p.ldc % 8 || (p.residual && p.ldr % 8)) return false;
    if ((((size_t)p.A | (size_t)p.out | (size_t)p.residual) & 15) != 0) return false;
    if (p.ln_colsum && (p.rowstat_out || p.residual || !p.bias || p.ln_nparts > 8 || (p.ln_nparts <= 0 && !p.ln_stats))) return false;
    return true;
}
size_t gemm_wr_packed_bytes(int N, int K) { return (size_t)N * K * 2; }
int launch_wr_pack(hipStream_t st, const bf16_t* W, int N, int K, void* out) {
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(k_wr_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, 5, (uint4*)out);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// rows per workgroup: whole ring stages, chosen so that the grid is about one workgroup per CU
static int wr_rows_per_wg(const GemmParams& p) {
    const int R = 20480 / p.K, npanels = p.N / 320;
    int want = (256 + npanels - 1) / npanels;                 // row groups
    int rows = (p.M + want - 1) / want;
    rows = (rows + R - 1) / R * R;
    if (rows < 2 * R) rows = 2 * R;
    return rows;
}
int gemm_wr_parts(const GemmParams& p) { return 4 * (p.N / 320); }

template <int K>
static int launch_wr_t(hipStream_t st, const GemmParams& p, const void* wpk) {
    const int npanels = p.N / 320;
    const int rows = wr_rows_per_wg(p);
    const int grid = (p.M + rows - 1) / rows * npanels;
    const size_t lds = (size_t)3 * 40960 + 4 * 16 * 176;
    GyreProfScope prof_(KC_GEMM_WR, st, 2.0 * p.M * (double)p.N * p.K,
                        (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * 2.0 * (p.residual ? 2.0 : 1.0));
#define GYRE_WR_GO(RES_, RS_, LNF_, ABL_)                                                                                \
    do {                                                                                                                 \
        auto kern = k_gemm_wr<K, 5, RES_, RS_, LNF_, ABL_>;                                                              \
        static std::atomic<unsigned long long> attr_done{0};                                                             \
        if (gyre_lds_attr_needed(attr_done))                                                                             \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p, (const uint4*)wpk, rows, npanels);                   \
    } while (0)
    const bool lnf = p.ln_colsum != nullptr, rs = p.rowstat_out != nullptr;
#ifdef GYRE_WR_ABLATIONS
    if (const int abl = (p.debug >> 23) & 7; abl && p.residual && !rs) {
        if (abl == 1) GYRE_WR_GO(true, false, false, 1);
        else if (abl == 2) GYRE_WR_GO(true, false, false, 2);
        else if (abl == 3) GYRE_WR_GO(true, false, false, 3);
        else if (abl == 4) GYRE_WR_GO(true, false, false, 4);
        else if (abl == 6) GYRE_WR_GO(true, false, false, 6);
        else GYRE_WR_GO(true, false, false, 7);
        GYRE_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (lnf) GYRE_WR_GO(false, false, true, 0);
    else if (p.residual) { if (rs) GYRE_WR_GO(true, true, false, 0); else GYRE_WR_GO(true, false, false, 0); }
    else { if (rs) GYRE_WR_GO(false, true, false, 0); else GYRE_WR_GO(false, false, false, 0); }
#undef GYRE_WR_GO
    GYRE_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_wr(hipStream_t st, const GemmParams& p, const void* wpk) {
    if (!gemm_wr_supports(p)) GYRE_FAIL(-6, "gemm: problem outside the W-resident kernel's domain");
    if (!wpk) GYRE_FAIL(-6, "gemm: the W-resident kernel needs the packed weight copy");
    return launch_wr_t<320>(st, p, wpk);
}
