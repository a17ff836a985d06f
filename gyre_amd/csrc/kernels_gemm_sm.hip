// Small-problem bf16 MFMA GEMM (tile config 32): linear problems too small for the 8-wave tiles to cover the chip - the C x C
// projections of the 32x32 / 16x16 / 8x8 UNet levels at batch 1 - 8 (M = 128 ... 4096), the text-context K / V projections
// (M = 77 B), i.e. everything the planner used to give to the register-staged 4-wave kernel (k_gemm<64, 64, ...>):
//
//   C[M][N] = bf16( A[M][K] * W[N][K]^T + bias (+ residual) )
//
// plus the forms the model's deep levels need at small batch: the fused Q | K | V of a self-attention (GemmParams::vt_out: tiles at
// or behind vt_col0 leave transposed, V^T[b][channel][token]), two-source rows (A | A2: the 1x1 shortcut over a skip concatenation),
// and the long-K few-row problems the planner used to cut into K slices.  (Round 4 also had the folded LayerNorm and per-row
// output statistics here behind a tuning bit: measured slower than the separate LayerNorm pass, deleted in round 5.)
//
// Those launches are LATENCY-bound, not bandwidth- or MFMA-bound: 1.7 GFLOP and 6 - 16 MB took 15 - 24 us because the old
// kernel fetches a K step (global -> registers -> LDS), waits for it, multiplies, and only then fetches the next one - about a
// microsecond of exposed memory latency per 64-wide K step, 10 - 20 steps per workgroup.  Here the same 64x64 tile (4 waves,
// 32x32 per wave, v_mfma_f32_32x32x16_bf16) is fed by a FOUR-stage LDS ring filled by LDS-DMA: three K steps (48 KB) are in
// flight per workgroup while one is multiplied, two workgroups per CU (64 KB each).  A stage is 16 pieces of 8 rows x 128 B
// (whole 128-byte lines); chunk c of row r sits at r * 128 + ((c ^ ((r >> 1) & 7)) * 16), the swizzle of the pipelined
// 8-wave kernel (conflict-free ds_read_b128 for the 32x32 fragment).  Counted s_waitcnt vmcnt + one barrier per K step;
// past-the-end requests re-read step 0 into a free slot so that the in-flight count stays uniform.  No spills allowed
// (build.py checks it).  The epilogue stages the fp32 tile through the (then idle) ring and writes whole 128-byte rows.
//
// (An eight-stage ring for grids of at most one workgroup per CU was measured: 10.1 -> 10.6 us at M = 512, N = K = 1280 - what is left
// is launch and prologue, not the K chain.)
// (A 128x128 tile for the problems whose 64x64 grid is more than one round - M = 2048, N = K = 1280: 640 tiles - was measured too:
// 21.2 -> 22.1 us.  That problem moves 210 MB through the CUs' load paths with 64x64 tiles, 820 KB per CU = 22 us at the 37 GB/s a
// CU's path delivers (profiles/HISTORY.md 4c); 128x128 halves the bytes but leaves 96 CUs idle.  The 128x160 tile of the 8-wave kernel: 20 us.)
// Same K order on one accumulator as every other tile config.
// Replaces the cuBLAS GEMMs behind torch.nn.Linear in the third-party UNet the reference calls at
// gyre/pipeline/unet/core.py:274 (BasicTransformerBlock to_q / to_k / to_v / to_out, proj_in / proj_out at the deep levels).
#include "gemm_shared.h"
#include <atomic>

typedef __attribute__((address_space(3))) char lds_char_t;

namespace {
__device__ __forceinline__ void sm_glds(unsigned lds_addr, const void* vptr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(vptr) : "memory");
}
}  // namespace

template <bool RES>
__global__ __launch_bounds__(256, 2) void k_gemm_sm(GemmParams p, int tiles_m, int tiles_n) {
    constexpr int NS = 4, STAGE = 16384, WOFF = 8192, PPW = 4;       // ring stages; bytes per stage; W half; requests per wave and stage
    constexpr int ROWF = 68;                                           // floats per row of the epilogue tile (64 + 4: conflict-free)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;    // neighbouring workgroups share their A rows
    const int m0 = tm * 64, n0 = tn * 64;
    const int nk = p.K / 64;
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;

    // ---- ring requests: wave w fetches pieces 2w, 2w + 1 (rows 16w .. 16w + 15) of the A half and of the W half of every stage;
    // lane l' of a piece: row 8 * piece + (l' >> 3), slot l' & 7 holds chunk slot ^ ((row >> 1) & 7)
    // Two-source rows (the 1x1 shortcut of an up-path resnet reads the concatenation [x | skip] without materialising it): K steps
    // below C1 / 64 come from A, the others from A2 (GemmParams::A2, lda2, C1 - whole 64-channel steps of one source)
    const char* src[PPW];
    const char* src2[2];
    const int k1 = p.C1 / 64;                                        // first K step of the second source (= nk for one source)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * w + 8 * i + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int mr = min(m0 + row, p.M - 1);                        // rows past M re-read row M - 1 (never stored)
        src[i] = (const char*)(p.A + (size_t)mr * p.lda) + chunk * 16;
        src2[i] = (const char*)(p.A2 + (size_t)mr * p.lda2) + chunk * 16 - (size_t)k1 * 128;
        const int nr = min(n0 + row, p.N - 1);
        src[2 + i] = p.W_blk ? (const char*)(p.W_blk + (size_t)(nr >> 3) * (p.K >> 6) * 512 + (nr & 7) * 64) + chunk * 16
                             : (const char*)(p.W + (size_t)nr * p.K) + chunk * 16;
    }
    const unsigned wstep = p.W_blk ? 1024u : 128u;                    // bytes between consecutive K steps of a weight row group
    auto issue = [&](int kt, int slot) __attribute__((always_inline)) {
        const int kk = kt < nk ? kt : 0;                            // past-the-end: re-read step 0 (never consumed)
        const unsigned dst = lds0 + slot * STAGE + (2 * w) * 1024;
        const bool second = kk >= k1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sm_glds(dst + i * 1024, (second ? src2[i] : src[i]) + (size_t)kk * 128);
            sm_glds(dst + WOFF + i * 1024, src[2 + i] + (size_t)kk * wstep);
        }
    };
    // Round 5: the epilogue's bias and residual rows are requested HERE, in front of the ring (by hand: a compiler-visible load
    // would put the compiler's own vmcnt bookkeeping into the counted loop).  They are older than every ring request, so the first
    // counted wait covers them, and the epilogue no longer ends the launch with one more cold round trip.
    // INVARIANT (the compiler does not know these "=v" registers are pending until the vmcnt(0) behind the K loop): no instruction on
    // any path between a request and that wait may read or write its destination registers - no copy, no spill, no re-materialisation.
    // Checked on the ISA of every build: tests/test_host_cpu.py::test_hand_issued_loads_stay_untouched_until_their_wait.
    typedef unsigned sm_u32x4 __attribute__((ext_vector_type(4)));
    sm_u32x4 e_res[2], e_b0[2], e_b1[2];
    const bool e_plain = !(p.vt_out && n0 >= p.vt_col0);
    if (e_plain) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i, row = v >> 3, c8 = v & 7;
            const int m = min(m0 + row, p.M - 1), n = n0 + 8 * c8;
            if constexpr (RES)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_res[i]) : "v"(p.residual + (size_t)m * p.ldr + n) : "memory");
            if (p.bias) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_b0[i]) : "v"(p.bias + n) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_b1[i]) : "v"(p.bias + n + 4) : "memory");
            }
        }
    }
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);

    // fragment addresses inside a stage: row R, k16 sub-step j: chunk 2j + hi at R * 128 + ((chunk ^ ((R >> 1) & 7)) * 16)
    const int ra = wm * 32 + l31, rw = wn * 32 + l31;
    const int swa = (ra >> 1) & 7, sww = (rw >> 1) & 7;
    f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt landed for this wave (younger requests: stages kt + 1, kt + 2), then for every wave; slot (kt - 1) % NS is free.
        // lgkmcnt(0) IN FRONT of the barrier (round 6, the ring discipline of k_gemm8 / kernels_attn.hip): the refill issued right
        // behind this barrier targets slot (kt + 3) % NS == (kt - 1) % NS, the slot step kt - 1 just read, and hipcc may sink the last
        // MFMAs of that step together with the lgkmcnt wait in front of them BELOW the barrier (the builtin orders memory operations,
        // not register-only MFMAs or compiler-derived waits) - a wave could then pass the barrier with ds_reads of the slot still
        // queued while another wave's LDS-DMA for the same slot is on its way.  Every fragment read of this wave has returned here.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(kt + 3, (kt + 3) % NS);
        const char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8_t af = *(const bf16x8_t*)(st + ra * 128 + (((2 * j + hi) ^ swa) * 16));
            const bf16x8_t wf = *(const bf16x8_t*)(st + WOFF + rw * 128 + (((2 * j + hi) ^ sww) * 16));
            acc = GYRE_MFMA_32x32x16(af, wf, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // past-the-end requests: nothing may land in LDS after this point
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: fp32 tile through LDS (the ring is idle), whole rows out ------------------------------------------------------------
    float* tile = (float*)smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        tile[m * ROWF + wn * 32 + l31] = acc[r];
    }
    __syncthreads();
    if (p.vt_out && n0 >= p.vt_col0) {
        // V columns of a fused Q | K | V projection leave TRANSPOSED: vt_out[(b * Cv + channel) * ldt + token] (what the attention
        // kernel streams); a thread packs 8 consecutive tokens of one channel, 8 lanes cover 128 contiguous bytes of a V^T row
        const int cv_total = p.N - p.vt_col0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i, col = v >> 3, t8 = v & 7;
            const int m = m0 + 8 * t8, n = n0 + col;
            if (m >= p.M) continue;
            const float bb = p.bias ? p.bias[n] : 0.f;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = tile[(8 * t8 + e) * ROWF + col] + bb;
            const int b = m / p.tokens_per_batch, t = m - b * p.tokens_per_batch;
            *(uint4*)(p.vt_out + ((size_t)b * cv_total + (n - p.vt_col0)) * p.ldt + t) = pack8(f);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + 256 * i, row = v >> 3, c8 = v & 7;
        const int m = m0 + row, n = n0 + 8 * c8;
        if (m >= p.M) continue;
        const float4 a = *(const float4*)(tile + row * ROWF + 8 * c8), b = *(const float4*)(tile + row * ROWF + 8 * c8 + 4);
        float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (p.bias) {       // requested in front of the ring (vmcnt(0) above covers them)
            f[0] += __uint_as_float(e_b0[i].x); f[1] += __uint_as_float(e_b0[i].y); f[2] += __uint_as_float(e_b0[i].z); f[3] += __uint_as_float(e_b0[i].w);
            f[4] += __uint_as_float(e_b1[i].x); f[5] += __uint_as_float(e_b1[i].y); f[6] += __uint_as_float(e_b1[i].z); f[7] += __uint_as_float(e_b1[i].w);
        }
        if constexpr (RES) {
            float r8[8];
            unpack8(make_uint4(e_res[i].x, e_res[i].y, e_res[i].z, e_res[i].w), r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r8[e];
        }
        *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldc + n) = pack8(f);
    }
}

bool gemm_sm_supports(const GemmParams& p) {
    if (p.mode != GEMM_LINEAR || p.out_mode != OUT_BF16 || p.batch > 1 || p.geglu) return false;
    if (p.vt_out && (p.vt_col0 <= 0 || p.vt_col0 % 64 || p.tokens_per_batch <= 0 || p.tokens_per_batch % 8 || p.ldt % 8 || p.M % 8 ||
                     p.residual || ((size_t)p.vt_out & 15)))
        return false;
    if (p.A2 && p.A2 != p.A && (p.C1 <= 0 || p.C1 >= p.K || p.C1 % 64 || p.lda2 % 8 || ((size_t)p.A2 & 15))) return false;
    if (p.rowbias || p.colstat_out || p.w_sample_stride || p.rowstat_out || p.ln_colsum) return false;
    if (p.K % 64 || p.K < 64 || p.N % 64 || p.M < 1 || p.lda % 8 || p.ldc % 8 || (p.residual && p.ldr % 8)) return false;
    if ((((size_t)p.A | (size_t)p.W | (size_t)p.out | (size_t)p.residual) & 15) != 0 || (((size_t)p.bias) & 15) != 0) return false;
    return true;
}

int launch_gemm_sm(hipStream_t st, const GemmParams& p) {
    if (!gemm_sm_supports(p)) GYRE_FAIL(-6, "gemm: problem outside the small-problem kernel's domain (linear, K and N multiples of 64, bf16 row-major output)");
    const int tiles_m = (p.M + 63) / 64, tiles_n = p.N / 64;
    const size_t lds = (size_t)4 * 16384 > (size_t)64 * 68 * 4 ? (size_t)4 * 16384 : (size_t)64 * 68 * 4;
    GyreProfScope prof_(KC_GEMM_SM, st, 2.0 * p.M * (double)p.N * p.K,
                        (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * 2.0 * (p.residual ? 2.0 : 1.0));
    if (p.residual) {
        auto kern = k_gemm_sm<true>;
        static std::atomic<unsigned long long> attr_done{0};
        if (gyre_lds_attr_needed(attr_done)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, st, p, tiles_m, tiles_n);
    } else {
        auto kern = k_gemm_sm<false>;
        static std::atomic<unsigned long long> attr_done{0};
        if (gyre_lds_attr_needed(attr_done)) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, st, p, tiles_m, tiles_n);
    }
    GYRE_LAUNCH_CHECK();
    return 0;
}
