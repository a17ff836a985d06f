// "A-resident" bf16 MFMA GEMM for short reductions (K = 320 / 640: every Linear / 1x1 projection and the GEGLU FF1 of the
// 64x64 and 32x32 UNet levels):
//
//   C[M][N] = epilogue( A[M][K] * W[N][K]^T )          same contract as k_gemm8 / k_gemm4s (linear mode, bf16 row-major out)
//
// Why another kernel.  With K this short a BM x BN output tile lives for only 5 - 10 K steps, and every tile re-delivers its
// A rows AND its W rows through the CU's load path (L2 -> LDS, ~37 GB/s per CU whatever issues the requests - the round 1 - 4 model, withdrawn in round 5: profiles/HISTORY.md 4c):
// 328 KB per 256x256 tile of the 64x64 FF1, ten tiles per CU, 9.7 us each at the load path's rate against 4.3 us of MFMA
// time.  Here a workgroup (8 waves) OWNS 256 rows for its whole life: wave w keeps its 32 x K slab of A in REGISTERS as the
// MFMA's second operand (KT = K / 16 fragments of 4 registers: 80 registers at K = 320, 160 at K = 640), loaded once, and
// sweeps the N axis in tiles of TNW = 32 * NB weight rows that every wave reads from one LDS ring:
//
//   * the only operand that moves in the loop is W: 2 * K bytes per weight row and workgroup, instead of (BM + BN) * K * 2 per
//     BM x BN tile - 45 % of the load-path bytes of the 256x256 tiling at K = 320, N = 2560 - and W (1.6 MB) sits in every L2;
//   * the weights are PRE-PACKED per handle (k_ar_pack, cached like the LayerNorm-folded copies) in MFMA fragment order, so a
//     ring stage is a linear 40 KB copy (LDS-DMA, 5 x 1 KB per wave) and every ds_read_b128 of the loop is lane-linear
//     (conflict-free by construction, immediate offsets);
//   * two accumulator sets alternate: the epilogue of tile t-1 (LayerNorm fold, GEGLU, rounding - VALU) sits in the same
//     basic block as the MFMAs of tile t, and its 16-byte row stores are issued one tile later, behind the next barrier;
//   * a lane owns ONE output row (operands swapped: weights are the MFMA's A), so the folded LayerNorm needs two registers
//     of row statistics for the whole kernel, and v_permlane32_swap turns the fragment's 4-channel runs into 8 consecutive
//     channels per lane (16-byte stores, 32 B contiguous per row and instruction).
//
// Synchronisation: ring of three stages; tile t+2 is requested right behind the barrier at the top of tile t (the slot of
// tile t-1, which every wave has finished reading by then) and waited for two tiles later with a COUNTED s_waitcnt
// vmcnt(PPW): loads return in order, so "at most the PPW requests of the younger tile outstanding" implies the older tile
// has landed even though stores (which share the counter and may retire out of order with loads) are in flight - they can
// only make the wait stricter.  The kernel must not spill (scratch stores would count too): build.py checks it.
//
// Same K summation order as every other tile config (k ascending in steps of 16 on one accumulator).
// Replaces the cuBLAS GEMMs behind torch.nn.Linear in the third-party UNet the reference calls at
// gyre/pipeline/unet/core.py:274 (BasicTransformerBlock: to_q / to_k / to_v / to_out, GEGLU ff.net.0.proj, proj_in / proj_out).
#include "gemm_shared.h"
#include <atomic>
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int V> using ic = std::integral_constant<int, V>;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(ic<I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ void ar_glds(unsigned lds_addr, const void* vptr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(vptr) : "memory");
}

// W[N][K] (row-major bf16) -> tiles of 32 * NB rows in fragment order: tile t, fragment f = ks * NB + b, lane l holds the 8
// values W[32 * (NB * t + b) + (l & 31)][16 * ks + 8 * (l >> 5) ...]: what lane l feeds v_mfma_f32_32x32x16_bf16 as A
__global__ __launch_bounds__(256) void k_ar_pack(const bf16_t* W, int N, int K, int NB, uint4* out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int KT = K / 16;
    const size_t per_tile = (size_t)NB * KT * 64;
    const size_t total = (size_t)(N / (32 * NB)) * per_tile;
    if (idx >= total) return;
    const int t = (int)(idx / per_tile), r = (int)(idx - (size_t)t * per_tile);
    const int f = r >> 6, l = r & 63;
    const int ks = f / NB, b = f - ks * NB;
    const int n = 32 * (NB * t + b) + (l & 31), k = 16 * ks + 8 * (l >> 5);
    out[idx] = *(const uint4*)(W + (size_t)n * K + k);
}

// ABL: timing ablations (results are garbage; builds with GYRE_AR_ABLATIONS only): 1 = no epilogue arithmetic, 2 = no MFMAs,
// 4 = no output stores, 8 = no fragment reads (one stale fragment), 16 = no ring requests / waits
// VT: the fused Q | K | V projection of a self-attention (GemmParams::vt_out): weight rows >= vt_col0 are the V projection, whose
// 32 x 32 output blocks leave TRANSPOSED - V^T[b][channel][token], what the attention kernel streams - through a 2 KB
// per-wave LDS patch (16 x ds_write_b16 per lane, read back as two 16-byte rows of 8 tokens)
template <int KT, int NB, bool LNF, bool GEGLU, bool RES, bool RS, int ABL = 0, bool VT = false>
__global__ __launch_bounds__(512, 2) void k_gemm_ar(GemmParams p, const char* wpk, int nt_total, int n_split) {
    constexpr int TB = NB * KT * 1024;          // bytes of one N tile = one ring stage
    constexpr int NS = 3;
    constexpr int PPW = TB / 8192;              // 1 KB LDS-DMA pieces per wave and tile
    constexpr int TNW = NB * 32;                // weight rows per tile
    constexpr int OUTC = GEGLU ? TNW / 2 : TNW; // output columns per tile
    static_assert(TB % 8192 == 0 && PPW >= 1 && PPW <= 8, "tile bytes must split into whole 1 KB pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int bm = blockIdx.x / n_split, sp = blockIdx.x - bm * n_split;
    const int t0 = (int)((long)sp * nt_total / n_split), t1 = (int)((long)(sp + 1) * nt_total / n_split);
    const int nt = t1 - t0;                     // >= 1 (launcher)
    const int m = bm * 256 + w * 32 + l31;
    const bool mok = m < p.M;
    const int mc = mok ? m : p.M - 1;           // rows past M re-read row M-1 (never stored)
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    float* cbias = (float*)(smem + NS * TB);    // [nt * TNW] bias of this workgroup's weight rows (zeros when absent)
    float* ccols = cbias + nt * TNW;            // [nt * TNW] LNF: column sums of the gamma-folded weights
    // VT: this wave's transposition patch [32 channels][32 tokens] bf16, and the first V block of this workgroup's range
    bf16_t* const vscr = (bf16_t*)(smem + NS * TB + (size_t)nt * TNW * 4 * (LNF ? 2 : 1)) + w * 1024;
    const int vblk0 = VT ? p.vt_col0 / 32 - t0 * NB : (1 << 30);       // local block index >= vblk0: a V block

    // ---- ring: request tile tt (local index) into slot `slot`; past-the-end requests re-read tile 0 (never consumed) so that
    // the in-flight count stays uniform
    const char* wsrc = wpk + (size_t)t0 * TB + (size_t)w * 1024 + lane * 16;
    auto issue = [&](int tt, int slot) {
        const char* s = wsrc + (size_t)(tt < nt ? tt : 0) * TB;
        const unsigned d = lds0 + slot * TB + w * 1024;
        if constexpr (ABL & 16) return;
#pragma unroll
        for (int j = 0; j < PPW; ++j) ar_glds(d + j * 8192, s + j * 8192);
    };
    issue(0, 0);
    issue(1, 1);

    // ---- the wave's A slab: lane (row m, half hi) holds k = 16 ks + 8 hi .. + 7 of every 16-wide step
    bf16x8_t af[KT];
    {
        const bf16_t* arow = p.A + (size_t)mc * p.lda + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) af[ks] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(arow + 16 * ks));
    }
    // ---- column constants of the whole N range into LDS
    {
        const int n0 = t0 * TNW, cnt = nt * TNW;
        for (int i = tid; i < cnt; i += 512) {
            cbias[i] = p.bias ? p.bias[n0 + i] : 0.f;
            if (LNF) ccols[i] = p.ln_colsum[n0 + i];
        }
    }
    // ---- folded LayerNorm: this lane's row statistics
    float lrstd = 1.f, lrmu = 0.f;
    if constexpr (LNF) {
        if (p.ln_nparts > 0) {
            float su = 0.f, sq = 0.f;
            for (int t = 0; t < p.ln_nparts; ++t) {
                const float2 v = ((const float2*)p.ln_parts)[(size_t)t * p.M + mc];
                su += v.x; sq += v.y;
            }
            const float invk = 1.0f / (float)p.K;
            const float mean = su * invk;
            lrstd = 1.0f / sqrtf(fmaxf(sq * invk - mean * mean, 0.f) + p.ln_eps);
            lrmu = lrstd * mean;
        } else {
            const float2 rs = ((const float2*)p.ln_stats)[mc];
            lrstd = rs.x; lrmu = rs.y;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tiles 0 and 1 landed (and every prologue load above)
    __syncthreads();                                     // ... for every wave; constants visible

    // Accumulator blocks (32 weight rows x 32 output rows each).  NB == 1 (K = 640): two blocks alternate - while the MFMAs of
    // block g run into one, the epilogue arithmetic of block g-1 reads the other.  NB == 2 (K = 320): two SETS of two blocks
    // alternate per ring stage, and inside a stage consecutive MFMAs go to different blocks: a v_mfma whose C operand is the
    // result of the MFMA issued just before it runs at full rate only when NOTHING is issued between the two - with the
    // epilogue's vector instructions in the gaps every dependent MFMA held the matrix pipe ~40 cycles longer, for both waves
    // of the SIMD (tools/ar_ablate.py: 130 - 160 cycles per MFMA stage instead of 64)
    f32x16_t accA, accB;
    f32x16_t accX[2], accY[2];
    constexpr int SPB = GEGLU ? 1 : 2;          // 16-byte stores per lane and block
    u32x4_t holdA[SPB], holdB[SPB];             // rounded outputs of the last even / odd block, stored behind the next barrier
    u32x4_t holdX[2][SPB], holdY[2][SPB];       // NB == 2: ... of the two blocks of the last even / odd tile
    const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int OUTB = GEGLU ? 16 : 32;       // output columns per block
    bf16_t* const orow = (bf16_t*)p.out + (size_t)mc * p.ldc + 8 * hi + (size_t)t0 * OUTC;
    const bf16_t* const rrow = RES ? p.residual + (size_t)mc * p.ldr + 8 * hi + (size_t)t0 * OUTC : nullptr;

    // residual rows of block g (local index), requested by hand: the compiler's own vmcnt bookkeeping must not see loads
    // between the ring's requests (it would drain them); consumed behind the counted wait inside the epilogue
    u32x4_t rresA[SPB], rresB[SPB];
    u32x4_t rresX[2][SPB], rresY[2][SPB];
    auto res_issue = [&](int g, u32x4_t (&rr)[SPB]) {
        if constexpr (RES) {
            const bf16_t* r0 = rrow + g * OUTB;
#pragma unroll
            for (int s = 0; s < SPB; ++s)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(rr[s]) : "v"(r0 + 16 * s) : "memory");
        }
    };
    // RS: per-row sum / sum of squares of the ROUNDED outputs this lane stores (GemmParams::rowstat_out): a lane keeps its row
    // for the whole kernel, so the two sums live in registers; the halves (hi = 0 / 1) are combined once at the end
    float rs_s = 0.f, rs_q = 0.f;

    // ---- the epilogue of one block, cut into NSTG stages ------------------------------------------------------------------------
    // Left to itself hipcc emits the block's 20 - 40 MFMAs (each behind its own ds_read + full lgkmcnt wait) and THEN the
    // ~250 vector instructions of the previous block's epilogue: the wave's two instruction streams never overlap.  The fused
    // loop below therefore places one stage behind each MFMA and pins it there (sched_barrier): ~11 vector instructions per
    // 32x32x16 MFMA - what a wave can issue in the 64 cycles its MFMA slot lasts when two waves share the SIMD's matrix pipe.
    // State between stages (one block's epilogue is in flight at a time):
    float4 kb0, kb1, kc0, kc1;           // bias / folded-LayerNorm column sums of the two 4-channel runs being processed
    float eo[8];                         // GEGLU: the block's 8 outputs; else: the two runs on their way to one 16-byte store
    float ev = 0.f, eg = 0.f, ez = 0.f, ep = 0.f;
    constexpr int NSTG = GEGLU ? 19 : (VT ? 9 : 8);
    auto epi_stage = [&](auto s_c, int g, const f32x16_t& acc, u32x4_t (&hold)[SPB], u32x4_t (&rr)[SPB], auto drain_tag) {
        constexpr int S = decltype(s_c)::value;
        constexpr int WAITN = decltype(drain_tag)::value ? 0 : PPW;   // DRAIN: nothing younger than the residual request is in flight
        auto load_consts = [&](int ja, int jb) {
            const float* cb = cbias + g * 32 + 4 * hi;
            kb0 = *(const float4*)(cb + 8 * ja); kb1 = *(const float4*)(cb + 8 * jb);
            if constexpr (LNF) {
                const float* cc = ccols + g * 32 + 4 * hi;
                kc0 = *(const float4*)(cc + 8 * ja); kc1 = *(const float4*)(cc + 8 * jb);
            }
        };
        auto aff = [&](float a, float bb, float cc) {
            if constexpr (LNF) return fmaf(a, lrstd, fmaf(-lrmu, cc, bb));
            else return a + bb;
        };
        auto f4 = [](const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; };
        // x = eo[0..3] holds channels c .. c+3 (+4 hi), y = eo[4..7] the run 8 channels further: after the swap a lane holds 8
        // consecutive channels
        auto swap_runs = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(eo[i]), __float_as_uint(eo[4 + i]), false, false);
                eo[i] = __uint_as_float(r[0]); eo[4 + i] = __uint_as_float(r[1]);
            }
        };
        auto pack_to = [&](int st) {
            if constexpr (RES) {
                const u32x4_t rv = rr[st];
                eo[0] += bf16lo(rv[0]); eo[1] += bf16hi(rv[0]); eo[2] += bf16lo(rv[1]); eo[3] += bf16hi(rv[1]);
                eo[4] += bf16lo(rv[2]); eo[5] += bf16hi(rv[2]); eo[6] += bf16lo(rv[3]); eo[7] += bf16hi(rv[3]);
            }
            hold[st] = u32x4_t{pack_bf16x2(eo[0], eo[1]), pack_bf16x2(eo[2], eo[3]), pack_bf16x2(eo[4], eo[5]), pack_bf16x2(eo[6], eo[7])};
            if constexpr (RS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf16lo(hold[st][e]), hv = bf16hi(hold[st][e]);
                    rs_s += lo; rs_q = fmaf(lo, lo, rs_q);
                    rs_s += hv; rs_q = fmaf(hv, hv, rs_q);
                }
            }
        };
        // LEAN (K = 640: the A slab takes 160 registers): column constants are fetched one element / one run ahead as scalars
        // into four registers instead of 16-byte vectors into sixteen
        constexpr bool LEAN = KT > 20;
        if constexpr (GEGLU) {
            // value runs j = 0, 1 (accumulator elements 0 .. 7), their gates 16 weight rows further (elements 8 .. 15);
            // element e = 4 j + i: stage 2 e + 1 = affine + first half of erf-GELU's polynomial, stage 2 e + 2 = the rest
            // (the arithmetic of gelu_erf_f, common.h, in the same order: bit-identical to the other kernels' GEGLU)
            auto lean_consts = [&](int e) {      // kb0.x / kb1.x / kc0.x / kc1.x = constants of element e's value and gate
                const float* cb = cbias + g * 32 + 4 * hi + 8 * (e / 4) + (e % 4);
                kb0.x = cb[0]; kb1.x = cb[16];
                if constexpr (LNF) { const float* cc = ccols + g * 32 + 4 * hi + 8 * (e / 4) + (e % 4); kc0.x = cc[0]; kc1.x = cc[16]; }
            };
            if constexpr (S == 0) { if constexpr (LEAN) lean_consts(0); else load_consts(0, 2); }
            else if constexpr (S <= 16) {
                constexpr int e = (S - 1) / 2, j = e / 4, i = e % 4;
                if constexpr ((S - 1) % 2 == 0) {
                    ev = aff(acc[4 * j + i], f4(kb0, LEAN ? 0 : i), f4(kc0, LEAN ? 0 : i));
                    eg = aff(acc[8 + 4 * j + i], f4(kb1, LEAN ? 0 : i), f4(kc1, LEAN ? 0 : i));
                    ez = fabsf(eg) * 0.70710678118654752f;
                    ep = fmaf(ez, 0.0000430638f, 0.0002765672f);
                    ep = fmaf(ez, ep, 0.0001520143f);
                    ep = fmaf(ez, ep, 0.0092705272f);
                    if constexpr (LEAN && e < 7) lean_consts(e + 1);
                } else {
                    ep = fmaf(ez, ep, 0.0422820123f);
                    ep = fmaf(ez, ep, 0.0705230784f);
                    ep = fmaf(ez, ep, 1.0f);
                    ep *= ep; ep *= ep; ep *= ep; ep *= ep;
                    const float q = __builtin_amdgcn_rcpf(ep);
                    eo[e] = ev * fmaf(-0.70710678118654752f * ez, q, fmaxf(eg, 0.f));
                    if constexpr (!LEAN && S == 8) load_consts(1, 3);
                }
            } else if constexpr (S == 17) swap_runs();
            else if constexpr (S == 18) pack_to(0);
        } else {
            constexpr int half = S / 4, q = S % 4;      // half 0: runs 0, 1 -> store 0; half 1: runs 2, 3 -> store 1
            auto run_consts = [&](int j) {
                kb0 = *(const float4*)(cbias + g * 32 + 4 * hi + 8 * j);
                if constexpr (LNF) kc0 = *(const float4*)(ccols + g * 32 + 4 * hi + 8 * j);
            };
            if constexpr (S == 8) {
                // (VT only) V block: every value of the block sits in the patch; two 16-byte rows of 8 tokens per lane come back
                if (g >= vblk0) {
                    hold[0] = *(const u32x4_t*)(vscr + (lane >> 2) * 32 + (lane & 3) * 8);
                    hold[1] = *(const u32x4_t*)(vscr + (16 + (lane >> 2)) * 32 + (lane & 3) * 8);
                }
            } else if constexpr (q == 0) {
                if constexpr (S == 0) run_consts(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) eo[i] = aff(acc[8 * half + i], f4(kb0, i), f4(kc0, i));
                run_consts(2 * half + 1);
            } else if constexpr (q == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) eo[4 + i] = aff(acc[8 * half + 4 + i], f4(kb0, i), f4(kc0, i));
                if constexpr (half == 0) run_consts(2);
            } else if constexpr (q == 2) {
                if constexpr (RES && half == 0) {
                    if constexpr (SPB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rr[0]), "+v"(rr[1]) : "n"(WAITN) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rr[0]) : "n"(WAITN) : "memory");
                }
                if (VT && g >= vblk0) {
                    // channel n = 8 (2 half + r) + 4 hi + i of run r, token l31 (a wave's LDS operations complete in order)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        vscr[(16 * half + 4 * hi + i) * 32 + l31] = f32_to_bf16(eo[i]);
                        vscr[(16 * half + 8 + 4 * hi + i) * 32 + l31] = f32_to_bf16(eo[4 + i]);
                    }
                } else swap_runs();
            } else {
                if (!(VT && g >= vblk0)) pack_to(half);
            }
        }
    };
    // the epilogue of block g alone (last block of a workgroup)
    auto epi_only = [&](int g, const f32x16_t& acc, u32x4_t (&hold)[SPB], u32x4_t (&rr)[SPB], auto drain_tag) {
        static_for<0, NSTG>([&](auto s_c) { epi_stage(s_c, g, acc, hold, rr, drain_tag); });
    };
    using no_drain = std::false_type;
    using drain = std::true_type;
    // One ring stage = NB * KT = 40 MFMAs, issued as ONE stream: the weight fragment of stage i + PF is requested from LDS
    // while stage i's MFMA issues (the two waves of a SIMD run the same code between the same barriers, so a wave's exposed
    // LDS latency is not covered by its partner: with a prefetch distance of 2 the fragment reads and the MFMAs simply added
    // up, 44 + 43 us of the 64x64 FF1 - tools/ar_ablate.py), and behind each MFMA sits one stage of the previous block's
    // epilogue.
    //   NB == 2: stages 0 .. KT-1 = block g (accumulators `a0`) beside the epilogue of block g - 1 (`a1`, when HAS_PREV),
    //            stages KT .. 2 KT-1 = block g + 1 (`a1`) beside the epilogue of block g (`a0`)
    //   NB == 1: the stage's single block (`a0`) beside the epilogue of block g - 1 (`a1`), one stage every other MFMA
    constexpr int PF = KT == 20 ? 6 : 3, RING = KT == 20 ? 8 : 4, NSTREAM = NB * KT;   // (K = 640: the A slab leaves no registers for more)
    auto fused_tile = [&](int slot, int g, f32x16_t& a0, f32x16_t& a1, u32x4_t (&h0)[SPB], u32x4_t (&h1)[SPB], u32x4_t (&r0)[SPB],
                          u32x4_t (&r1)[SPB], auto has_prev_tag) {
        constexpr bool HAS_PREV = decltype(has_prev_tag)::value;
        const char* sb = smem + slot * TB + lane * 16;
        bf16x8_t wf[RING];
        auto frag_off = [](int i) { return NB == 2 ? ((i % KT) * 2 + i / KT) * 1024 : i * 1024; };
        static_for<0, PF>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            wf[i % RING] = *(const bf16x8_t*)(sb + frag_off(i));
        });
        static_for<0, NSTREAM>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr int b = NB == 2 ? i / KT : 0, ks = NB == 2 ? i % KT : i;
            if constexpr (i + PF < NSTREAM && !(ABL & 8)) wf[(i + PF) % RING] = *(const bf16x8_t*)(sb + frag_off(i + PF));
            f32x16_t& acc = b == 0 ? a0 : a1;
            constexpr int fi = (ABL & 8) ? 0 : i % RING;
            if constexpr (!(ABL & 2)) acc = GYRE_MFMA_32x32x16(wf[fi], af[ks], ks == 0 ? zero16 : acc, 0, 0, 0);
            else { if constexpr (ks == 0) acc = zero16; asm volatile("" ::"v"(wf[fi]), "v"(af[ks])); }
            if constexpr (NB == 2) {
                if constexpr (b == 0) {
                    if constexpr (HAS_PREV && ks < NSTG && !(ABL & 1)) epi_stage(ic<ks>{}, g - 1, a1, h1, r1, no_drain{});
                } else {
                    if constexpr (ks < NSTG && !(ABL & 1)) epi_stage(ic<ks>{}, g, a0, h0, r0, no_drain{});
                }
                if constexpr (ks == 0 && (ABL & 1)) {
                    if constexpr (b == 0) h1[0] = u32x4_t{__float_as_uint(a1[0]), __float_as_uint(a1[5]), __float_as_uint(a1[10]), __float_as_uint(a1[15])};
                    else h0[0] = u32x4_t{__float_as_uint(a0[0]), __float_as_uint(a0[5]), __float_as_uint(a0[10]), __float_as_uint(a0[15])};
                }
            } else {
                constexpr int S = (i & 1) ? -1 : i / 2;
                if constexpr (HAS_PREV && S >= 0 && S < NSTG && !(ABL & 1)) epi_stage(ic<S>{}, g - 1, a1, h1, r1, no_drain{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using has_prev = std::true_type;
    using no_prev = std::false_type;
    // NB == 2: one ring stage = 2 blocks x KT MFMAs as ONE stream in fragment order (k step outer, block inner: consecutive MFMAs
    // write different accumulators), the fragment of stage i + PF2 requested while stage i issues, and behind every MFMA one
    // stage of the PREVIOUS tile's epilogue (block 0 behind MFMAs 0 .., block 1 behind MFMAs KT ..)
    // (tuning variants, valid results: ABL & 32 = fragment request pinned in FRONT of the stage's MFMA, & 64 = 10-deep prefetch,
    //  & 128 = the second wave of every SIMD at s_setprio 1, & 256 = the epilogue stage in front of the MFMA)
    constexpr int PF2 = (ABL & 64) ? 10 : ((RES && RS) || VT) ? 4 : 6, RING2 = (ABL & 64) ? 12 : VT ? 6 : 8;
    if constexpr (ABL & 128) { if (w >= 4) __builtin_amdgcn_s_setprio(1); }
    auto fused_tile2 = [&](int slot, int gp, f32x16_t (&cur)[2], f32x16_t (&prev)[2], u32x4_t (&hp)[2][SPB], u32x4_t (&rp)[2][SPB],
                           auto has_prev_tag) {
        constexpr bool HAS_PREV = decltype(has_prev_tag)::value;
        const char* sb = smem + slot * TB + lane * 16;
        bf16x8_t wf[RING2];
        static_for<0, PF2>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            wf[i % RING2] = *(const bf16x8_t*)(sb + i * 1024);
        });
        static_for<0, 2 * KT>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr int ks = i >> 1, b = i & 1;
            if constexpr (i + PF2 < 2 * KT && !(ABL & 8)) wf[(i + PF2) % RING2] = *(const bf16x8_t*)(sb + (i + PF2) * 1024);
            if constexpr (ABL & 32) __builtin_amdgcn_sched_barrier(0);
            constexpr int fi = (ABL & 8) ? 0 : i % RING2;
#define GYRE_AR_MFMA()                                                                                                          \
            do {                                                                                                                \
                if constexpr (!(ABL & 2)) cur[b] = GYRE_MFMA_32x32x16(wf[fi], af[ks], ks == 0 ? zero16 : cur[b], 0, 0, 0); \
                else { if constexpr (ks == 0) cur[b] = zero16; asm volatile("" ::"v"(wf[fi]), "v"(af[ks])); }                   \
            } while (0)
            if constexpr (!(ABL & 256)) GYRE_AR_MFMA();
            if constexpr (HAS_PREV && !(ABL & 1)) {
                if constexpr (i < NSTG) epi_stage(ic<i>{}, gp, prev[0], hp[0], rp[0], no_drain{});
                else if constexpr (i >= KT && i - KT < NSTG) epi_stage(ic<i - KT>{}, gp + 1, prev[1], hp[1], rp[1], no_drain{});
            }
            if constexpr (ABL & 256) { __builtin_amdgcn_sched_barrier(0); GYRE_AR_MFMA(); }
#undef GYRE_AR_MFMA
            if constexpr (HAS_PREV && (ABL & 1) && (i == 0 || i == KT)) {
                constexpr int q = i == 0 ? 0 : 1;
                hp[q][0] = u32x4_t{__float_as_uint(prev[q][0]), __float_as_uint(prev[q][5]), __float_as_uint(prev[q][10]), __float_as_uint(prev[q][15])};
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto store = [&](int g, const u32x4_t (&hold)[SPB]) {
        if constexpr (VT) {
            if (g >= vblk0) {       // V^T[(b * Cv + c) * ldt + token]: lane = (channel lane >> 2 (+16), tokens 8 (lane & 3) ..)
                const int row0 = bm * 256 + w * 32;                     // first row (= b * tokens + token) of this wave's block
                if (row0 >= p.M) return;                                // (M is a multiple of 32: tokens_per_batch % 32 == 0)
                const int bb = row0 / p.tokens_per_batch, tok0 = row0 - bb * p.tokens_per_batch;
                const int cv = (t0 * NB + g) * 32 - p.vt_col0, Cv = p.N - p.vt_col0;
                bf16_t* o = p.vt_out + ((size_t)bb * Cv + cv + (lane >> 2)) * p.ldt + tok0 + (lane & 3) * 8;
                *(u32x4_t*)o = hold[0];
                *(u32x4_t*)(o + (size_t)16 * p.ldt) = hold[1];
                return;
            }
        }
        if (!mok) return;
        if constexpr (ABL & 4) { if (hold[0][0] != 0x12345678u) return; }
        bf16_t* o = orow + g * OUTB;
#pragma unroll
        for (int s = 0; s < SPB; ++s) *(u32x4_t*)(o + 16 * s) = hold[s];
    };
    // top of tile tt: its stage has landed for every wave; the slot of tile tt-1 is free -> tile tt+2 goes there
    auto top = [&](int tt, int slot_next2) {
        if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto nxt = [](int s) { return s == 2 ? 0 : s + 1; };
    auto prv = [](int s) { return s == 0 ? 2 : s - 1; };
    const int nblk = nt * NB;                   // blocks of this workgroup; block g lives in tile g / NB

    // Block pairs (2q -> accA, 2q + 1 -> accB).  Pair q: [top] mma(2q) | epi(2q - 1)  ;  [top when NB == 1] mma(2q + 1) | epi(2q)
    // The stores of a block's outputs and the residual request of the NEXT epilogue go right behind a barrier, followed by
    // the tile request, so that a whole tile time lies between them and the counted wait they take part in.
    int slot = 1;                                // ring slot of the tile the next block belongs to
    if constexpr (NB == 2) {
        // tile tt = blocks 2 tt, 2 tt + 1; even tiles use set X, odd tiles set Y.  body(tt): [top] store the outputs of tile
        // tt - 2 (they sit in this tile's own set), request the residual rows of tile tt - 1 and ring stage tt + 2, then the
        // MFMAs of tile tt beside the epilogue of tile tt - 1
        auto store2 = [&](int tt, const u32x4_t (&h)[2][SPB]) { store(2 * tt, h[0]); store(2 * tt + 1, h[1]); };
        auto body = [&](int tt, int slot_, f32x16_t (&cur)[2], f32x16_t (&prev)[2], u32x4_t (&hc)[2][SPB], u32x4_t (&hp)[2][SPB],
                        u32x4_t (&rp)[2][SPB]) {
            top(tt, prv(slot_));
            if (tt >= 2) store2(tt - 2, hc);
            res_issue(2 * tt - 2, rp[0]);
            res_issue(2 * tt - 1, rp[1]);
            issue(tt + 2, prv(slot_));
            fused_tile2(slot_, 2 * tt - 2, cur, prev, hp, rp, has_prev{});
        };
        issue(2, 2);                             // (tiles 0 and 1 are resident: prologue)
        fused_tile2(0, 0, accX, accY, holdY, rresY, no_prev{});
        int tt = 1;
        for (; tt + 1 < nt; tt += 2) {
            body(tt, slot, accY, accX, holdY, holdX, rresX);
            slot = nxt(slot);
            body(tt + 1, slot, accX, accY, holdX, holdY, rresY);
            slot = nxt(slot);
        }
        bool lastY = false;
        if (tt < nt) {
            body(tt, slot, accY, accX, holdY, holdX, rresX);
            lastY = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // past-the-end requests: nothing may land in LDS after the wave exits
        // pending: the outputs of tile nt - 2 (in hold of the set that is NOT the last tile's), tile nt - 1 in its accumulators
        if (lastY) {
            store2(nt - 2, holdX);
            res_issue(2 * nt - 2, rresY[0]); res_issue(2 * nt - 1, rresY[1]);
            epi_only(2 * nt - 2, accY[0], holdY[0], rresY[0], drain{});
            epi_only(2 * nt - 1, accY[1], holdY[1], rresY[1], drain{});
            store2(nt - 1, holdY);
        } else {
            if (nt >= 2) store2(nt - 2, holdY);
            res_issue(2 * nt - 2, rresX[0]); res_issue(2 * nt - 1, rresX[1]);
            epi_only(2 * nt - 2, accX[0], holdX[0], rresX[0], drain{});
            epi_only(2 * nt - 1, accX[1], holdX[1], rresX[1], drain{});
            store2(nt - 1, holdX);
        }
    } else {
        // tile = block: even blocks in accA / holdA / rresA, odd ones in accB / holdB / rresB
        issue(2, 2);
        fused_tile(0, 0, accA, accB, holdA, holdB, rresA, rresB, no_prev{});
        int g = 1;
        for (; g + 1 < nblk; g += 2) {
            top(g, prv(slot));
            if (g >= 2) store(g - 2, holdB);
            res_issue(g - 1, rresA);
            issue(g + 2, prv(slot));
            fused_tile(slot, g, accB, accA, holdB, holdA, rresB, rresA, has_prev{});
            slot = nxt(slot);
            top(g + 1, prv(slot));
            store(g - 1, holdA);
            res_issue(g, rresB);
            issue(g + 3, prv(slot));
            fused_tile(slot, g + 1, accA, accB, holdA, holdB, rresA, rresB, has_prev{});
            slot = nxt(slot);
        }
        bool lastB = false;
        if (g < nblk) {
            top(g, prv(slot));
            if (g >= 2) store(g - 2, holdB);
            res_issue(g - 1, rresA);
            issue(g + 2, prv(slot));
            fused_tile(slot, g, accB, accA, holdB, holdA, rresB, rresA, has_prev{});
            lastB = true;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // outputs still pending: lastB: holdA = block nblk-2 (stored now), accB = block nblk-1
        //                        else : holdB = block nblk-2 (when nblk >= 2), accA = block nblk-1
        if (lastB) {
            store(nblk - 2, holdA);
            res_issue(nblk - 1, rresB);
            epi_only(nblk - 1, accB, holdB, rresB, drain{});
            store(nblk - 1, holdB);
        } else {
            if (nblk >= 2) store(nblk - 2, holdB);
            res_issue(nblk - 1, rresA);
            epi_only(nblk - 1, accA, holdA, rresA, drain{});
            store(nblk - 1, holdA);
        }
    }
    if constexpr (RS) {
        rs_s += __shfl_xor(rs_s, 32);
        rs_q += __shfl_xor(rs_q, 32);
        if (mok && hi == 0) ((float2*)p.rowstat_out)[(size_t)sp * p.M + m] = make_float2(rs_s, rs_q);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
bool gemm_ar_supports(const GemmParams& p) {
    if (p.mode != GEMM_LINEAR || p.out_mode != OUT_BF16 || p.batch > 1) return false;
    // K = 640: the A slab takes 160 registers; only the GEGLU FF1 (the layer that gains most) has a form that fits without spills
    if (p.K != 320 && !(p.K == 640 && p.geglu)) return false;
    if (p.A2 && p.A2 != p.A) return false;
    if (p.rowbias || p.colstat_out) return false;
    if (p.vt_out && (p.K != 320 || p.geglu || p.residual || p.rowstat_out || p.vt_col0 % 64 || p.tokens_per_batch % 32 || p.ldt % 8 ||
                     p.M % 32 || ((size_t)p.vt_out & 15)))
        return false;
    const int tnw = p.K == 320 ? 64 : 32;
    if (p.N % tnw || p.lda % 8 || p.ldc % 8 || (p.residual && p.ldr % 8)) return false;
    if ((((size_t)p.A | (size_t)p.out | (size_t)p.residual) & 15) != 0) return false;
    if (p.geglu && (p.residual || p.rowstat_out)) return false;
    if (p.ln_colsum && (p.rowstat_out || p.residual)) return false;      // (no layer of the UNet asks for these combinations)
    return true;
}
size_t gemm_ar_packed_bytes(int N, int K) { return (size_t)N * K * 2; }
int launch_ar_pack(hipStream_t st, const bf16_t* W, int N, int K, void* out) {
    const int NB = K == 320 ? 2 : 1;
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(k_ar_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, NB, (uint4*)out);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// workgroups that share a row block split its N tiles; chosen so that the grid covers the chip
int gemm_ar_nsplit(const GemmParams& p) {
    const int tiles_m = (p.M + 255) / 256;
    const int tnw = p.K == 320 ? 64 : 32;
    const int nt = p.N / tnw;
    int ns = 1;
    while (tiles_m * ns * 2 <= 256 && ns * 2 <= nt && nt / (ns * 2) >= 4) ns *= 2;
    return ns;
}

template <int KT, int NB>
static int launch_ar_t(hipStream_t st, const GemmParams& p, const void* wpk) {
    const int tiles_m = (p.M + 255) / 256;
    const int nt = p.N / (32 * NB);
    const int ns = gemm_ar_nsplit(p);
    const int grid = tiles_m * ns;
    const int nt_max = (nt + ns - 1) / ns;
    const size_t lds = (size_t)3 * NB * KT * 1024 + (size_t)nt_max * 32 * NB * 4 * (p.ln_colsum ? 2 : 1) + (p.vt_out ? 8 * 2048 : 0);
    if (lds > 160 * 1024) GYRE_FAIL(-6, "gemm: the A-resident kernel's column constants exceed LDS for this N");
    const double n_out = p.geglu ? p.N / 2.0 : (double)p.N;
    GyreProfScope prof_(KC_GEMM_AR, st, 2.0 * p.M * (double)p.N * p.K,
                        (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * n_out * 2.0 * (p.residual ? 2.0 : 1.0));
#define GYRE_AR_GO(LNF_, GG_, RES_, RS_)                                                                                 \
    do {                                                                                                                 \
        auto kern = k_gemm_ar<KT, NB, LNF_, GG_, RES_, RS_>;                                                             \
        static std::atomic<unsigned long long> attr_done{0};                                                             \
        if (gyre_lds_attr_needed(attr_done))                                                                             \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, (const char*)wpk, nt, ns);                           \
    } while (0)
    const bool lnf = p.ln_colsum != nullptr, rs = p.rowstat_out != nullptr;
    if (p.vt_out) {
        if constexpr (KT == 20) {
            if (lnf) { auto kern = k_gemm_ar<KT, NB, true, false, false, false, 0, true>;
                       static std::atomic<unsigned long long> ad{0};
                       if (gyre_lds_attr_needed(ad)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                       hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, (const char*)wpk, nt, ns); }
            else { auto kern = k_gemm_ar<KT, NB, false, false, false, false, 0, true>;
                   static std::atomic<unsigned long long> ad{0};
                   if (gyre_lds_attr_needed(ad)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                   hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, (const char*)wpk, nt, ns); }
            GYRE_LAUNCH_CHECK();
            return 0;
        } else GYRE_FAIL(-6, "gemm: the fused Q|K|V form of the A-resident kernel exists for K = 320 only");
    }
#ifdef GYRE_AR_ABLATIONS
    if (p.geglu && !lnf && KT == 20 && (p.debug >> 23) & 511) {
        const int abl = (p.debug >> 23) & 511;
#define GYRE_AR_ABL(A_)                                                                                                  \
        if (abl == A_) {                                                                                                 \
            auto kern = k_gemm_ar<KT, NB, false, true, false, false, A_>;                                                \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, (const char*)wpk, nt, ns);                       \
        }
        GYRE_AR_ABL(1) GYRE_AR_ABL(2) GYRE_AR_ABL(3) GYRE_AR_ABL(4) GYRE_AR_ABL(5) GYRE_AR_ABL(8) GYRE_AR_ABL(16) GYRE_AR_ABL(7) GYRE_AR_ABL(10)
        GYRE_AR_ABL(32) GYRE_AR_ABL(64) GYRE_AR_ABL(96) GYRE_AR_ABL(128) GYRE_AR_ABL(256) GYRE_AR_ABL(160) GYRE_AR_ABL(13) GYRE_AR_ABL(12) GYRE_AR_ABL(37) GYRE_AR_ABL(69)
#undef GYRE_AR_ABL
        GYRE_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (p.geglu) { if (lnf) GYRE_AR_GO(true, true, false, false); else GYRE_AR_GO(false, true, false, false); }
    else if constexpr (KT == 20) {
        if (p.residual) {
            if (rs) GYRE_AR_GO(false, false, true, true);
            else GYRE_AR_GO(false, false, true, false);
        } else {
            if (lnf) GYRE_AR_GO(true, false, false, false);
            else if (rs) GYRE_AR_GO(false, false, false, true);
            else GYRE_AR_GO(false, false, false, false);
        }
    } else GYRE_FAIL(-6, "gemm: at K = 640 the A-resident kernel exists for the GEGLU form only");
#undef GYRE_AR_GO
    GYRE_LAUNCH_CHECK();
    return 0;
}

int launch_gemm_ar(hipStream_t st, const GemmParams& p, const void* wpk) {
    if (!gemm_ar_supports(p)) GYRE_FAIL(-6, "gemm: problem outside the A-resident kernel's domain");
    if (!wpk) GYRE_FAIL(-6, "gemm: the A-resident kernel needs the packed weight copy");
    if (p.K == 320) return launch_ar_t<20, 2>(st, p, wpk);
    return launch_ar_t<40, 1>(st, p, wpk);
}
