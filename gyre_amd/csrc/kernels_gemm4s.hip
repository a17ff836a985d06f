// One-wave-per-SIMD bf16 MFMA GEMM / implicit-GEMM 3x3 convolution ("4s": 4 waves, single occupancy).
//
//   C[M][N] = epilogue( A[M][K] * W[N][K]^T )          same contract as k_gemm8 (kernels_gemm.hip)
//
// Why another kernel: the 8-wave kernels run two waves per SIMD in lock step behind one barrier per K step; all
// waves issue their LDS-DMA requests, then all compute, then all wait (MFMA busy ~41 %, profiles/r01e).  Here a
// workgroup is 4 waves, ONE per SIMD, each owning the whole 512-entry register file (MI355X_MICROARCH.md "Register
// files"), so the wave tile doubles to 128 x 160 (-36 % LDS reads per FLOP) and everything that is not an MFMA - the
// ds_read_b128 fragment reads of the NEXT k16 sub-step, the LDS-DMA requests of the NEXT K step and their address
// arithmetic - is issued from the gaps of the wave's own v_mfma_f32_32x32x16_bf16 stream (32 cycles each = ~8 issue
// slots, cdna_hip_programming.md "4-wave, one-wave-per-SIMD" rules).
//
// K step = 64 (128-byte LDS rows, full cache lines per DMA row), 4 sub-steps of k = 16, two LDS stages.
// The per-K-step barrier sits BEFORE the last sub-step, not after it:
//
//   iteration t :  P1  mfma(sub0) | read sub1        | second half of the DMA requests for stage t+1
//                  P2  mfma(sub1) | read sub2
//                  P3  mfma(sub2) | read sub3        ; s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier   (B_t)
//                  P0  mfma(sub3) | read sub0 of t+1 | first half of the DMA requests for stage t+2
//
// At B_t every wave has (a) received its own DMA pieces of stage t+1 and (b) finished every LDS read of stage t (sub3's
// fragments are already in registers), so behind it stage t+1 may be read and the buffer of stage t may be refilled -
// both while the 20 MFMAs of sub3 still run.  The matrix pipe therefore never waits for a ds_read round trip at the K
// step seam, and a DMA piece has at least one full K step (~2500 cycles) to land.
//
// LDS image: lane-linear per DMA instruction (8 rows x 128 B); the 16-byte slot of row r is XORed with (r >> 1) & 7 on
// the per-lane SOURCE address and on the ds_read_b128 (guide rule 21).  For the 32x32x16 fragment (lane l: row l & 31,
// k-slot 2*ks + (l >> 5)) the four 16-lane ds_read_b128 groups then touch 16 distinct 16-byte slots of the 256-byte
// bank row: conflict-free (checked exhaustively in tests/test_host_cpu.py::test_gemm4s_lds_swizzle_is_conflict_free).
//
// Operands are issued swapped (weights = MFMA A, activations = MFMA B): a lane ends up with ONE output row
// (m = lane & 31) and, per fragment, four runs of 4 consecutive channels -> float4 LDS writes in the staged epilogue.
// Same K summation order as every other tile config (64-channel chunk outer, tap inner), so results are
// bit-identical to k_gemm8's (tests/test_gpu_kernels.py::test_all_tile_configs_sum_in_the_same_order).
//
// Replaces cuBLAS/cuDNN (hipBLASLt/MIOpen) GEMM + conv reached through torch.nn.Linear / torch.nn.Conv2d inside the
// third-party UNet/VAE the reference calls at gyre/pipeline/unet/core.py:274 and
// gyre/pipeline/unified_pipeline.py:309,1531.
#include "gemm_shared.h"
#include <atomic>
#include <type_traits>

template <int V> using ic = std::integral_constant<int, V>;

// LDS-DMA requests as inline asm: the compiler's own selection of the builtin falls back to 64-bit VGPR addresses (one
// v_lshl_add_u64 per request); the saddr form takes the uniform base in an SGPR pair and a 32-bit per-lane offset.
// M0 (LDS destination) is written in the same statement that uses it (cdna_hip_programming.md 5.7); completion is
// tracked by hand: every request is retired by the "s_waitcnt vmcnt(0)" in front of the K-step barrier.
__device__ __forceinline__ void glds_saddr(unsigned lds_addr, unsigned voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void glds_vaddr(unsigned lds_addr, const void* vptr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(vptr) : "memory");
}

// Raw-buffer form for the conv A operand: out-of-range lanes (voffset >= num_records) deliver zeros, which is the conv's
// zero padding without a second pointer per lane.
typedef __attribute__((ext_vector_type(4))) int srd_t;
__device__ __forceinline__ void blds(unsigned lds_addr, unsigned voff, srd_t srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned bytes) {
    const size_t a = (size_t)base;
    return srd_t{(int)(unsigned)a, (int)(unsigned)(a >> 32), (int)bytes, 0x00020000};
}

// ZFILL: how the conv's zero padding reaches LDS: 0 = out-of-range raw-buffer requests, 1 = 64-bit pointers to a zero page
// WM x WN waves: 4 (one per SIMD, 512 registers each) or 8 (two per SIMD, the 256x320 tile: 160 accumulators per wave).
// LNF: LayerNorm folded into the GEMM (GemmParams::ln_colsum with ln_stats or ln_parts): the epilogue applies
// rstd * acc - rstd * mean * colsum + bias per element (linear mode, unsplit; instantiated for the 8-wave 256x320 tile only -
// the weight-dominated GEGLU FF1 of the 16x16 level)
// CS: column statistics of the output tile for the GroupNorm that consumes it (GemmParams::colstat_out; unsplit launches)
template <int BM, int BN, int WM, int WN, int MODE, int ZFILL, bool LNF = false, bool CS = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 4) void k_gemm4s(GemmParams p, int tiles_m, int tiles_n, int splits, int group_m) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int AP = BM / (8 * NW), BP = BN / (8 * NW);   // DMA pieces (8 rows x 128 B) per wave per stage: A rows, W rows
    constexpr int DTOT = AP + BP, DH0 = (DTOT + 1) / 2;     // pieces [0, DH0) form half 0 of a stage, [DH0, DTOT) half 1
    constexpr int STAGE = (BM + BN) * 128;
    static_assert((NW == 4 || NW == 8) && TM % 32 == 0 && TN % 32 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0 &&
                  MI * NI * 16 <= (NW == 4 ? 256 : 160) && (NI == 4 || NI == 5), "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntiles = tiles_m * tiles_n;
    const int split = blockIdx.x / ntiles;
    const int bid = xcd_tile_id(blockIdx.x - split * ntiles, ntiles);
    int tm, tn;
    // weight-dominated problems (deep UNet levels): tm fastest, the workgroups of an XCD stream the same weight slice;
    // otherwise groups of `group_m` tile rows, column-major inside a group, so that the ~32 tiles an XCD runs at a time
    // form a compact block of the output (A rows x W rows both shared in that L2)
    if ((long)p.M < (MODE == GEMM_CONV3 ? 9L : 1L) * p.N) { tn = bid / tiles_m; tm = bid - tn * tiles_m; }
    else {
        const int per = group_m * tiles_n;
        const int g = bid / per, r = bid - g * per;
        const int rows = min(group_m, tiles_m - g * group_m);
        tn = r / rows; tm = g * group_m + (r - tn * rows);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WN, wn = w - wm * WN;
    const int prow = lane >> 3;                                        // row of this lane inside a DMA piece
    const int kvs = (lane & 7) ^ ((4 * (w & 1) + (lane >> 4)) & 7);    // global 16-byte k-slot this lane fetches
    const bf16_t* zero = p.zero_page;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS offset of the ring
    // Tuning ablations (results are garbage) exist only in builds with -DGYRE_GEMM_ABLATIONS: each test is a scalar and / branch
    // pair per DMA piece inside the K loop, and a wave's issue slots are what bounds that loop (280 instructions around 40 MFMAs)
#ifdef GYRE_GEMM_ABLATIONS
    const bool abl_zero = p.debug & 1, abl_nodma = p.debug & 2, abl_a0 = p.debug & 8, abl_w0 = p.debug & 16;
#else
    constexpr bool abl_zero = false, abl_nodma = false, abl_a0 = false, abl_w0 = false;
#endif

    // ---- per-lane DMA source state ------------------------------------------------------------------------------
    // piece i of this wave covers tile rows 8 * (w + NW * i) .. + 7
    // weights: byte offset of (row, k-slot) from the tile's first weight row; rows past N re-read row N-1 (never stored)
    unsigned vo_w[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = min(n0 + 8 * (w + NW * i) + prow, p.N - 1) - n0;
        // (GemmParams::W_blk: 1-KiB blocks of 8 rows x 64 k - the request reads one contiguous KiB)
        vo_w[i] = p.W_blk ? (unsigned)(n >> 3) * (unsigned)(p.K >> 6) * 1024u + (unsigned)(n & 7) * 128u + kvs * 16u
                          : (unsigned)n * (unsigned)p.K * 2u + kvs * 16u;
    }
    const char* w_tile = p.W_blk ? (const char*)(p.W_blk + (size_t)(n0 >> 3) * (p.K >> 6) * 512) : (const char*)(p.W + (size_t)n0 * p.K);
    const bool wblk = p.W_blk != nullptr;
    // activations
    unsigned vo_a[AP];                   // linear: byte offset of (row, k-slot) from the tile's first row
    int a_sbp[AP], a_y0[AP], a_x0[AP];   // conv: sample base pixel relative to the tile's first sample, window origin
    const int hw_o = MODE == GEMM_CONV3 ? p.Ho * p.Wo : 1;
    const int ns0 = m0 / hw_o;           // first sample this tile touches (conv): sources are addressed relative to it
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int row = m0 + 8 * (w + NW * i) + prow;
        if (MODE == GEMM_LINEAR) {
            const unsigned rl = (unsigned)(min(row, p.M - 1) - m0);   // rows past M re-read row M-1 (never stored)
            vo_a[i] = rl * (unsigned)p.lda * 2u + kvs * 16u;
        } else {
            const int n = row / hw_o, rem = row - n * hw_o;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_sbp[i] = (n - ns0) * p.Hi * p.Wi;
            a_y0[i] = row < p.M ? oy * p.stride - p.pad : 0x40000000;   // rows past M: every tap fails the bounds test
            a_x0[i] = ox * p.stride - p.pad;
        }
    }
    const char* a_tile = (const char*)(p.A + (size_t)m0 * p.lda);
    const int Hlim = p.ups ? (p.Hup ? p.Hup : 2 * p.Hi) : p.Hi, Wlim = p.ups ? (p.Wup ? p.Wup : 2 * p.Wi) : p.Wi;
    const int ups = p.ups ? 1 : 0;
    // conv sources, based at the tile's first sample; a tile spans a handful of samples, so 32-bit offsets suffice
    const size_t spix = (size_t)ns0 * p.Hi * p.Wi;
    const char* cA = (const char*)(p.A + spix * p.lda);
    const char* cA2 = (const char*)(p.A2 + spix * p.lda2);
    srd_t srdA{}, srdA2{};
    if (MODE == GEMM_CONV3 && ZFILL == 0) {
        const size_t left = (size_t)(p.M / hw_o - ns0) * p.Hi * p.Wi;      // pixels from the tile's first sample to the end
        const size_t b1 = left * p.lda * 2, b2 = left * p.lda2 * 2;
        srdA = make_srd(cA, (unsigned)(b1 < 0x7ffffff0ull ? b1 : 0x7ffffff0ull));
        srdA2 = make_srd(cA2, (unsigned)(b2 < 0x7ffffff0ull ? b2 : 0x7ffffff0ull));
    }
    // folded 1x1 shortcut (GemmParams::sc_*): a second image pair of the output's size, walked through the centre tap BEHIND the conv's K
    // steps.  The conv sources are dead from the first shortcut step on, so that step REPLACES them (bases, buffer descriptors, strides,
    // split) instead of carrying a second set through the loop - the kernel sits at 82 of ~100 scalar registers, and a "s" asm
    // operand that no longer fits is silently handed over in vector registers.  step_src is called with non-decreasing kc.
    const int nk_conv = MODE == GEMM_CONV3 ? 9 * (p.Cin / BK) : 0;
    int src_C1 = p.C1;
    unsigned src_ld = (unsigned)p.lda * 2u, src_ld2 = (unsigned)p.lda2 * 2u;
    bool on_shortcut = false;
    // Uniform description of one K step's sources.  A step past the end of this block's K range is requested all the
    // same, from the zero page / out of range (every lane the same 16 bytes): the loop body stays one straight-line block.
    struct StepSrc { const char* ab; const char* wb; srd_t srd; unsigned soff, ld2; int ky, kx; bool live, live_w; };
    auto step_src = [&](int kc, bool live0) {
        StepSrc s;
        const bool live = live0 && !abl_a0;             // ablation: activations from the zero page only
        s.live_w = live0 && !abl_w0;                    // ablation: weights from the zero page only
        s.live = live;
        int kw;
        if (MODE == GEMM_LINEAR) {
            kw = kc * BK;
            s.ab = live ? a_tile + (size_t)kw * 2 : (const char*)zero;
        } else {
            int c0;
            if (kc >= nk_conv && p.sc_K) {                  // (wave-uniform) a K step of the folded shortcut: centre tap of the second image pair
                if (!on_shortcut) {                         // once: the shortcut's sources take the conv sources' place
                    on_shortcut = true;
                    cA = (const char*)(p.sc_A + spix * p.sc_lda);
                    cA2 = (const char*)(p.sc_A2 + spix * p.sc_lda2);
                    src_C1 = p.sc_C1; src_ld = (unsigned)p.sc_lda * 2u; src_ld2 = (unsigned)p.sc_lda2 * 2u;
                    if (ZFILL == 0) {
                        const size_t left = (size_t)(p.M / hw_o - ns0) * p.Hi * p.Wi;
                        const size_t b1 = left * p.sc_lda * 2, b2 = left * p.sc_lda2 * 2;
                        srdA = make_srd(cA, (unsigned)(b1 < 0x7ffffff0ull ? b1 : 0x7ffffff0ull));
                        srdA2 = make_srd(cA2, (unsigned)(b2 < 0x7ffffff0ull ? b2 : 0x7ffffff0ull));
                    }
                }
                c0 = (kc - nk_conv) * BK;
                s.ky = p.pad; s.kx = p.pad;
                kw = 9 * p.Cin + c0;
            } else {
                const int chunk = kc / 9, tap = kc - chunk * 9;
                s.ky = tap / 3; s.kx = tap - s.ky * 3;
                c0 = chunk * BK;
                kw = tap * p.Cin + c0;
            }
            const bool first = c0 < src_C1;
            s.soff = (unsigned)(first ? c0 : c0 - src_C1) * 2u;
            s.ld2 = first ? src_ld : src_ld2;
            s.srd = first ? srdA : srdA2;
            s.ab = (first ? cA : cA2) + s.soff;
        }
        s.wb = s.live_w ? w_tile + (wblk ? (size_t)(kw >> 6) * 1024 : (size_t)kw * 2) : (const char*)zero;
        return s;
    };
    // DMA piece D (compile-time, 0 .. DTOT-1: activations first) of the step described by s, into LDS stage `buf`
    auto dma_piece = [&](const StepSrc& s, int buf, auto d_c) {
        constexpr int D = decltype(d_c)::value;
        if (abl_nodma) return;
        const unsigned dst = lds0 + buf * STAGE + w * 1024;
        if constexpr (D < AP) {
            constexpr int i = D;
            if (MODE == GEMM_LINEAR) {
                glds_saddr(dst + i * (NW * 1024), s.live ? +vo_a[i] : 0u, s.ab);
            } else {
#ifdef GYRE_GEMM_ABLATIONS
                if ((p.debug & 0x80) && (s.ky | s.kx)) return;        // ablation: activations requested on tap 0 of a chunk only (1/9)
                if ((p.debug & 0x200) && s.kx) return;                // ablation: ... on the kx = 0 taps only (3/9)
#endif
                const int iy = a_y0[i] + s.ky, ix = a_x0[i] + s.kx;
                const bool ok = ((unsigned)iy < (unsigned)Hlim) & ((unsigned)ix < (unsigned)Wlim) & s.live;
                const unsigned pix = (unsigned)(a_sbp[i] + (iy >> ups) * p.Wi + (ix >> ups));   // meaningless when !ok
                const unsigned off = __umul24(pix, s.ld2) + kvs * 16u;
                if constexpr (ZFILL == 0) {
                    blds(dst + i * (NW * 1024), ok ? off : 0x80000000u, s.srd, s.soff);
                } else {
                    const char* ptr = s.ab + off;
                    glds_vaddr(dst + i * (NW * 1024), ok ? ptr : (const char*)zero);
                }
            }
        } else {
            constexpr int i = D - AP;
            glds_saddr(dst + BM * 128 + i * (NW * 1024), s.live_w ? +vo_w[i] : 0u, s.wb);
        }
    };
    // pieces [D0, D1) in program order (D1 - D0 <= 9)
    auto dma_range = [&](const StepSrc& s, int buf, auto d0_c, auto d1_c) {
        constexpr int D0 = decltype(d0_c)::value, D1 = decltype(d1_c)::value;
        static_assert(D1 - D0 <= 9, "piece range");
        if constexpr (D0 + 0 < D1) dma_piece(s, buf, ic<D0 + 0>{});
        if constexpr (D0 + 1 < D1) dma_piece(s, buf, ic<D0 + 1>{});
        if constexpr (D0 + 2 < D1) dma_piece(s, buf, ic<D0 + 2>{});
        if constexpr (D0 + 3 < D1) dma_piece(s, buf, ic<D0 + 3>{});
        if constexpr (D0 + 4 < D1) dma_piece(s, buf, ic<D0 + 4>{});
        if constexpr (D0 + 5 < D1) dma_piece(s, buf, ic<D0 + 5>{});
        if constexpr (D0 + 6 < D1) dma_piece(s, buf, ic<D0 + 6>{});
        if constexpr (D0 + 7 < D1) dma_piece(s, buf, ic<D0 + 7>{});
        if constexpr (D0 + 8 < D1) dma_piece(s, buf, ic<D0 + 8>{});
    };

    // ---- fragments and accumulators ---------------------------------------------------------------------------------
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8_t af[2][MI], wf[2][NI];
    const unsigned lo = (unsigned)(lane & 31) * 128u + ((unsigned)((lane >> 5) ^ ((lane >> 1) & 7)) << 4);

    // One phase = the NI x MI MFMAs of one k16 sub-step on fragment set U, cut into NI chunks (one weight fragment
    // each); between the chunks: the ds_read_b128 of the NEXT sub-step into the other set (R >= 0: all of them before
    // the last chunk, whose MFMAs cover their latency) and the LDS-DMA requests of half H (>= 0) of a later K step.
    // sched_barrier(0) pins the chunk boundaries: left alone, hipcc sinks the reads to the end of the phase (in front
    // of the barrier, where their latency is exposed) and bunches the DMA requests.
    auto phase = [&](auto use_c, auto rd_c, auto ks_c, const char* sb_rd, auto half_c, const StepSrc& ss, int buf_dma) {
        constexpr int U = decltype(use_c)::value, R = decltype(rd_c)::value, KS = decltype(ks_c)::value;
        constexpr int H = decltype(half_c)::value;
        constexpr int RT = MI + NI;
        constexpr int DB = H == 1 ? DH0 : 0, DN = H == 1 ? DTOT - DH0 : DH0;    // this half's pieces: DB .. DB + DN - 1
        const unsigned o = lo ^ (unsigned)(KS << 5);
        const char* pa = sb_rd + wm * (TM * 128) + o;
        const char* pw = sb_rd + BM * 128 + wn * (TN * 128) + o;
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[i][c] = GYRE_MFMA_32x32x16(wf[U][c], af[U][i], acc[i][c], 0, 0, 0);
            if constexpr (R >= 0 && c < NI - 1) {
                constexpr int r0 = c * RT / (NI - 1), r1 = (c + 1) * RT / (NI - 1);
#pragma unroll
                for (int r = r0; r < r1; ++r) {
                    if (r < MI) af[R][r] = *(const bf16x8_t*)(pa + r * 4096);
                    else wf[R][r - MI] = *(const bf16x8_t*)(pw + (r - MI) * 4096);
                }
            }
            if constexpr (H >= 0) dma_range(ss, buf_dma, ic<DB + c * DN / NI>{}, ic<DB + (c + 1) * DN / NI>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        chunk(ic<0>{}); chunk(ic<1>{}); chunk(ic<2>{}); chunk(ic<3>{});
        if constexpr (NI > 4) chunk(ic<4>{});
    };

    const int nk_all = (p.K + BK - 1) / BK;
    const int nk_per = (nk_all + splits - 1) / splits;
    const int kc0 = split * nk_per;
    const int nk = min(nk_all, kc0 + nk_per);       // this block's K steps: kc0 .. nk-1  (at least one, see launcher)

    // ---- prologue: stage kc0 resident, its sub-step 0 in set 0, first half of stage kc0+1 requested -------------------------
    {
        const StepSrc s0 = step_src(kc0, true);
        dma_range(s0, 0, ic<0>{}, ic<DH0>{});
        dma_range(s0, 0, ic<DH0>{}, ic<DTOT>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* pa = smem + wm * (TM * 128) + lo;
        const char* pw = smem + BM * 128 + wn * (TN * 128) + lo;
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][i] = *(const bf16x8_t*)(pa + i * 4096);
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[0][j] = *(const bf16x8_t*)(pw + j * 4096);
        const StepSrc s1 = step_src(kc0 + 1, kc0 + 1 < nk && !abl_zero);
        dma_range(s1, 1, ic<0>{}, ic<DH0>{});
    }
    const StepSrc none{};
    for (int kc = kc0; kc < nk; ++kc) {
        const int cur = (kc - kc0) & 1;
        const char* sb = smem + cur * STAGE;
        const StepSrc s1 = step_src(kc + 1, kc + 1 < nk && !abl_zero);
        phase(ic<0>{}, ic<1>{}, ic<1>{}, sb, ic<1>{}, s1, cur ^ 1);       // P1: sub0 | read sub1 | 2nd half of stage kc+1
        phase(ic<1>{}, ic<0>{}, ic<2>{}, sb, ic<-1>{}, none, 0);          // P2: sub1 | read sub2
        phase(ic<0>{}, ic<1>{}, ic<3>{}, sb, ic<-1>{}, none, 0);          // P3: sub2 | read sub3
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                       // B: stage kc+1 landed, stage kc's LDS reads done
        const StepSrc s2 = step_src(kc + 2, kc + 2 < nk && !abl_zero);
        phase(ic<1>{}, ic<0>{}, ic<0>{}, smem + (cur ^ 1) * STAGE, ic<0>{}, s2, cur);   // P0: sub3 | read sub0 of kc+1 | 1st half of kc+2
    }
    // the zero-page requests of the last iteration still target the ring the epilogue is about to reuse
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int hi = lane >> 5, l31 = lane & 31;
    const int m_base = m0 + wm * TM, n_base = n0 + wn * TN;
    if (!LNF && !CS && splits > 1) {
        // fp32 partial slab of this K slice; bias / residual / rounding happen once in k_splitk_reduce
        float* slab = p.splitk_ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m_base + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n_base + j * 32 + 8 * g + 4 * hi;
                    if (n < p.N)
                        *(float4*)(slab + (size_t)m * p.N + n) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                }
        }
        return;
    }
    if (!LNF && !CS && (p.debug & 4)) {  // tuning ablation: no epilogue (keep the accumulators alive with one conditional store)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 12345.678f) ((float*)p.out)[0] = sum;
        return;
    }

    // ---- staged epilogue: 32-row slabs of the wave tile through LDS (fp32), then whole rows out --------------------------------
    // (the operand ring is dead: the barrier above is behind every LDS read and every DMA write of every wave).
    if constexpr (CS) {
        // Column statistics for the consuming GroupNorm (GemmParams::colstat_out; plain bias / time-embedding / residual
        // epilogue only).  Passes of <= 2 column blocks = 8 (or 4) 16-byte vectors per row: a power of two, so a lane keeps
        // the SAME 8 channels for every row it reads back and their sum / sum of squares stay in registers over the wave
        // tile's rows (of the rounded values: what the consumer will read).  Per pass the lanes that share a channel vector
        // are folded through the wave's own slab region (LDS operations of one wave complete in order), then - behind the
        // block barrier - the WM wave tiles per column, then the channels of each unit; fixed order throughout.
        constexpr int NPC = (NI + 1) / 2, ROWC = 2 * 32 + 4, FST = 20;   // FST: floats per lane in the fold area (conflict-free b128)
        static_assert(64 * FST <= 32 * ROWC, "fold area exceeds the wave's slab");
        float* slabc = (float*)smem + w * (32 * ROWC);
        float2* cs_all = (float2*)((float*)smem + NW * 32 * ROWC);
        float2* chan = cs_all + NW * TN;
#pragma unroll
        for (int ps = 0; ps < NPC; ++ps) {
            const int j0 = 2 * ps, j1 = j0 + 2 < NI ? j0 + 2 : NI;
            const int vpr = (j1 - j0) * 4, R = 64 / vpr;            // vectors per row, rows per read-back step
            const int c8 = lane & (vpr - 1), r0 = lane / vpr;
            const int nn = n_base + j0 * 32 + c8 * 8;
            float cs_s[8], cs_q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m_base + i * 32 + l31;
                const float* rbias = (p.rowbias && m < p.M) ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
#pragma unroll
                for (int j = j0; j < j1; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n_base + j * 32 + 8 * g + 4 * hi;
                        float4 o = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                        if (n < p.N) {
                            if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
                            if (rbias) { const float4 tv = *(const float4*)(rbias + n); o.x += tv.x; o.y += tv.y; o.z += tv.z; o.w += tv.w; }
                        }
                        *(float4*)(slabc + l31 * ROWC + (j - j0) * 32 + 8 * g + 4 * hi) = o;
                    }
                for (int row = r0; row < 32; row += R) {
                    const int mm = m_base + i * 32 + row;
                    if (mm < p.M && nn < p.N) {
                        const float4 a = *(const float4*)(slabc + row * ROWC + c8 * 8);
                        const float4 b = *(const float4*)(slabc + row * ROWC + c8 * 8 + 4);
                        float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        if (p.residual) {
                            float r[8];
                            unpack8(*(const uint4*)(p.residual + (size_t)mm * p.ldr + nn), r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += r[e];
                        }
                        const uint4 pk = pack8(f);
                        *(uint4*)((bf16_t*)p.out + (size_t)mm * p.ldc + nn) = pk;
                        unpack8(pk, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { cs_s[e] += f[e]; cs_q[e] = fmaf(f[e], f[e], cs_q[e]); }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            float* fold = slabc + lane * FST;
            *(float4*)(fold) = make_float4(cs_s[0], cs_s[1], cs_s[2], cs_s[3]);
            *(float4*)(fold + 4) = make_float4(cs_s[4], cs_s[5], cs_s[6], cs_s[7]);
            *(float4*)(fold + 8) = make_float4(cs_q[0], cs_q[1], cs_q[2], cs_q[3]);
            *(float4*)(fold + 12) = make_float4(cs_q[4], cs_q[5], cs_q[6], cs_q[7]);
            __builtin_amdgcn_wave_barrier();
            if (lane < (j1 - j0) * 32) {
                const int cv = lane >> 3, e = lane & 7;
                float su = 0.f, sq = 0.f;
                for (int r = 0; r < R; ++r) { const float* src = slabc + (cv + vpr * r) * FST; su += src[e]; sq += src[8 + e]; }
                cs_all[w * TN + j0 * 32 + lane] = make_float2(su, sq);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        for (int t = tid; t < BN; t += NW * 64) {
            const int wn_ = t / TN, col = t - wn_ * TN;
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int ww = 0; ww < WM; ++ww) { const float2 v = cs_all[(ww * WN + wn_) * TN + col]; su += v.x; sq += v.y; }
            chan[t] = make_float2(su, sq);
        }
        __syncthreads();
        const int unit = p.colstat_unit;
        for (int u = tid; u * unit < BN; u += NW * 64) {
            const int n = n0 + u * unit;
            if (n < p.N) {
                float su = 0.f, sq = 0.f;
                for (int c = 0; c < unit; ++c) { const float2 v = chan[u * unit + c]; su += v.x; sq += v.y; }
                ((float2*)p.colstat_out)[(size_t)tm * (p.N / unit) + n / unit] = make_float2(su, sq);
            }
        }
        return;
    }
    // With 8 waves a full-width slab per wave would not fit LDS: the column blocks go in two passes.
    constexpr int NPASS = NW == 8 ? 2 : 1;
    constexpr int JP = (NI + NPASS - 1) / NPASS;          // column blocks (32 wide) per pass
    constexpr int ROWF = JP * 32 + 4;                     // floats per slab row (+16 B: conflict-free b128 writes)
    static_assert((size_t)NW * 32 * ROWF * 4 <= 160 * 1024, "epilogue slabs exceed LDS");
    float* slab = (float*)smem + w * (32 * ROWF);
    const bool gg = p.geglu != 0;
    const int n_out = gg ? p.N / 2 : p.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 32 + l31;
        const float* rbias = (p.rowbias && m < p.M) ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        // folded LayerNorm: this lane's accumulators of slab i all belong to row m
        float lrstd = 1.f, lrmu = 0.f;
        if constexpr (LNF) {
            if (m < p.M) {
                if (p.ln_nparts > 0) {            // partial sums left by the GEMM that produced the rows
                    float su = 0.f, sq = 0.f;
                    for (int t = 0; t < p.ln_nparts; ++t) {
                        const float2 v = ((const float2*)p.ln_parts)[(size_t)t * p.M + m];
                        su += v.x; sq += v.y;
                    }
                    const float invk = 1.0f / (float)p.K;
                    const float mean = su * invk;
                    lrstd = 1.0f / sqrtf(fmaxf(sq * invk - mean * mean, 0.f) + p.ln_eps);
                    lrmu = lrstd * mean;
                } else {
                    const float2 rs = ((const float2*)p.ln_stats)[m];
                    lrstd = rs.x; lrmu = rs.y;
                }
            }
        }
        auto affine = [&](float a0, float a1, float a2, float a3, const float4& b, const float4& c) {
            if (LNF) return make_float4(fmaf(a0, lrstd, fmaf(-lrmu, c.x, b.x)), fmaf(a1, lrstd, fmaf(-lrmu, c.y, b.y)),
                                        fmaf(a2, lrstd, fmaf(-lrmu, c.z, b.z)), fmaf(a3, lrstd, fmaf(-lrmu, c.w, b.w)));
            return make_float4(a0 + b.x, a1 + b.y, a2 + b.z, a3 + b.w);
        };
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            constexpr int dummy = 0; (void)dummy;
            const int j0 = ps * JP, j1 = (ps + 1) * JP < NI ? (ps + 1) * JP : NI;
            if (gg) {
#pragma unroll
                for (int j = j0; j < j1; ++j)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int nin = n_base + j * 32 + 8 * g + 4 * hi;     // value rows; their gates are 16 rows further
                        float4 bv = make_float4(0, 0, 0, 0), bg = make_float4(0, 0, 0, 0), cv = bv, cg = bv;
                        if (p.bias && nin < p.N) { bv = *(const float4*)(p.bias + nin); bg = *(const float4*)(p.bias + nin + 16); }
                        if (LNF && nin < p.N) { cv = *(const float4*)(p.ln_colsum + nin); cg = *(const float4*)(p.ln_colsum + nin + 16); }
                        const float4 val = affine(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3], bv, cv);
                        const float4 gate = affine(acc[i][j][4 * g + 8], acc[i][j][4 * g + 9], acc[i][j][4 * g + 10], acc[i][j][4 * g + 11], bg, cg);
                        float4 o;
                        o.x = val.x * gelu_erf_f(gate.x);
                        o.y = val.y * gelu_erf_f(gate.y);
                        o.z = val.z * gelu_erf_f(gate.z);
                        o.w = val.w * gelu_erf_f(gate.w);
                        *(float4*)(slab + l31 * ROWF + (j - j0) * 16 + 8 * g + 4 * hi) = o;
                    }
            } else {
#pragma unroll
                for (int j = j0; j < j1; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = n_base + j * 32 + 8 * g + 4 * hi;
                        float4 o = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                        if (n < p.N) {
                            if (LNF) o = affine(o.x, o.y, o.z, o.w, *(const float4*)(p.bias + n), *(const float4*)(p.ln_colsum + n));
                            else if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
                            if (rbias) { const float4 tv = *(const float4*)(rbias + n); o.x += tv.x; o.y += tv.y; o.z += tv.z; o.w += tv.w; }
                        }
                        *(float4*)(slab + l31 * ROWF + (j - j0) * 32 + 8 * g + 4 * hi) = o;
                    }
            }
            // read the slab back row-wise: 8 consecutive channels per lane -> one 16-byte store (residual added in fp32 first)
            const int vpr = (j1 - j0) * (gg ? 2 : 4);                               // 16-byte output vectors per row
            const int n_pass = (gg ? n_base / 2 + j0 * 16 : n_base + j0 * 32);      // first output column of this pass
            for (int v = lane; v < 32 * vpr; v += 64) {
                const int row = v / vpr, c8 = v - row * vpr;
                const int mm = m_base + i * 32 + row, nn = n_pass + c8 * 8;
                if (mm < p.M && nn < n_out) {
                    const float4 a = *(const float4*)(slab + row * ROWF + c8 * 8);
                    const float4 b = *(const float4*)(slab + row * ROWF + c8 * 8 + 4);
                    float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    if (p.residual) {
                        float r[8];
                        unpack8(*(const uint4*)(p.residual + (size_t)mm * p.ldr + nn), r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += r[e];
                    }
                    *(uint4*)((bf16_t*)p.out + (size_t)mm * p.ldc + nn) = pack8(f);
                }
            }
        }
    }
}

// Shapes the 4s kernels take: bf16 row-major output (or split-K slabs), 16-byte addressable rows, K steps that are
// whole 64-channel chunks of one source (so a step never needs zero fill along K), at least one K step per split.
bool gemm4s_supports(const GemmParams& p, int cfg) {
    if (cfg < 20 || cfg > 24) return false;
    const int bn = (cfg == 20 || cfg == 22 || cfg == 24) ? 320 : 256;
    if (p.out_mode != OUT_BF16 || p.vt_out) return false;
    const int n_out = p.geglu ? p.N / 2 : p.N;
    if (n_out % 8 || p.ldc % 8 || (p.residual && p.ldr % 8) || (((size_t)p.out | (size_t)p.residual) & 15)) return false;
    if (p.N % 8) return false;
    if (p.geglu && p.N % 32) return false;
    const int c1 = p.A2 ? p.C1 : (p.mode == GEMM_LINEAR ? p.K : p.Cin);
    if (p.mode == GEMM_LINEAR) { if (p.K % BK || (p.A2 && p.A2 != p.A)) return false; }
    else if (p.Cin % BK || c1 % BK || p.K != 9 * p.Cin + p.sc_K || (p.sc_K && (p.sc_K % BK || cfg != 24))) return false;
    if (p.mode == GEMM_CONV3) {
        // a tile's rows span at most 256 / (Ho*Wo) + 2 samples; sources are addressed relative to the first of them with
        // 24-bit pixel indices and 31-bit byte offsets
        const size_t span = ((size_t)256 / ((size_t)p.Ho * p.Wo) + 2) * p.Hi * p.Wi;
        const size_t ld = p.lda > p.lda2 ? p.lda : p.lda2;
        if (span >= (1u << 24) || ld * 2 >= (1u << 24) || span * ld * 2 >= 0x7ff00000ull) return false;
    }
    if ((size_t)bn * p.K * 2 >= (1ull << 32) || (size_t)256 * (size_t)(p.lda > p.lda2 ? p.lda : p.lda2) * 2 >= (1ull << 32)) return false;
    return true;
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg4s(hipStream_t st, const GemmParams& p, int kcls_base, int splits) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nk_all = (p.K + BK - 1) / BK;
    if (splits > nk_all) splits = nk_all;
    while (splits > 1 && (splits - 1) * ((nk_all + splits - 1) / splits) >= nk_all) --splits;   // no empty K slice
    const int grid = tiles_m * tiles_n * splits;
    // LDS: two operand stages; the epilogue's four 32-row fp32 slabs reuse them
    constexpr int NW = WM * WN, NI = BN / WN / 32, JP = NW == 8 ? (NI + 1) / 2 : NI;
    constexpr size_t ring = (size_t)2 * (BM + BN) * 128,
                     epi = (size_t)NW * 32 * (JP * 32 + 4) * 4 + (size_t)(NW * (BN / WN) + BN) * 8;   // slabs + column-statistics scratch
    constexpr size_t lds = ring > epi ? ring : epi;
    // tile rows per group: the ~32 tiles an XCD runs at a time should cover about as many A rows as W rows
    int group_m = 1;
    if (tiles_n >= 2) { group_m = 32 / tiles_n; if (group_m < 1) group_m = 1; if (tiles_n >= 8) group_m = 4; }
    if (group_m > tiles_m) group_m = tiles_m;
    const int kcls = kcls_base + (p.mode == GEMM_CONV3 ? 0 : 1);
    const double n_out = p.geglu ? p.N / 2.0 : (double)p.N;
    const double a_bytes = p.mode == GEMM_CONV3 ? (double)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Cin * 2.0
                                                : (double)p.M * p.K * 2.0;
    GyreProfScope prof_(kcls, st, 2.0 * p.M * (double)p.N * p.K,
                        a_bytes + (double)p.N * p.K * 2.0 + (double)p.M * n_out * 2.0 * (p.residual ? 2.0 : 1.0));
#define GYRE_GEMM4S_GO(MODE_, ZF_)                                                                                       \
    do {                                                                                                            \
        auto kern = k_gemm4s<BM, BN, WM, WN, MODE_, ZF_>;                                                                        \
        static std::atomic<unsigned long long> attr_done{0};                                                        \
        if (gyre_lds_attr_needed(attr_done))                                                                        \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p, tiles_m, tiles_n, splits, group_m);             \
    } while (0)
    if (p.mode == GEMM_LINEAR && p.ln_colsum) {
        if constexpr (BM == 256 && BN == 320 && NW == 8) {
            auto kern = k_gemm4s<BM, BN, WM, WN, GEMM_LINEAR, 0, true>;
            static std::atomic<unsigned long long> attr_done{0};
            if (gyre_lds_attr_needed(attr_done))
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p, tiles_m, tiles_n, splits, group_m);
        } else {
            GYRE_FAIL(-6, "gemm: the folded LayerNorm exists for the 8-wave 256x320 pipelined tile only");
        }
    } else if (p.colstat_out && splits == 1) {
        if constexpr (BM == 256 && BN == 320 && NW == 8) {
            if (p.mode != GEMM_CONV3) GYRE_FAIL(-6, "gemm: the pipelined tile emits column statistics for convolutions only");
            auto kern = k_gemm4s<BM, BN, WM, WN, GEMM_CONV3, 0, false, true>;
            static std::atomic<unsigned long long> attr_done{0};
            if (gyre_lds_attr_needed(attr_done))
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p, tiles_m, tiles_n, splits, group_m);
        } else {
            GYRE_FAIL(-6, "gemm: column statistics exist for the 8-wave 256x320 pipelined tile only");
        }
    } else if (p.mode == GEMM_LINEAR) GYRE_GEMM4S_GO(GEMM_LINEAR, 0);
#ifndef GYRE_GEMM_ABLATIONS
    else if (p.debug & 0x200) GYRE_GEMM4S_GO(GEMM_CONV3, 1);
#endif
    else GYRE_GEMM4S_GO(GEMM_CONV3, 0);
#undef GYRE_GEMM4S_GO
    GYRE_LAUNCH_CHECK();
    prof_.stop();
    if (splits > 1) return launch_splitk_reduce(st, p, splits);
    return 0;
}

int launch_gemm4s(hipStream_t st, const GemmParams& p, int cfg, int splits) {
    if (!gemm4s_supports(p, cfg)) GYRE_FAIL(-6, "gemm: the one-wave-per-SIMD tile configs need K in whole 64-channel steps and bf16 row-major output");
    switch (cfg) {
        case 20: return launch_cfg4s<192, 320, 2, 2>(st, p, KC_G4S_192x320, splits);
        case 21: return launch_cfg4s<256, 256, 2, 2>(st, p, KC_G4S_256x256, splits);
        case 22: return launch_cfg4s<128, 320, 2, 2>(st, p, KC_G4S_128x320, splits);
        case 23: return launch_cfg4s<128, 256, 2, 2>(st, p, KC_G4S_128x256, splits);
        case 24: return launch_cfg4s<256, 320, 4, 2>(st, p, KC_G4S_256x320, splits);
    }
    GYRE_FAIL(-1, "gemm: unknown tile config");
}
