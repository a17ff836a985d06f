// bf16 MFMA GEMM with an implicit-GEMM 3x3 convolution front end and fused epilogues.
//
//   C[M][N] = epilogue( A[M][K] * W[N][K]^T )
//
// A is either a row-major activation matrix (Linear / 1x1 conv over NHWC tokens) or an NHWC
// image gathered on the fly through a 3x3 window (stride 1|2, zero padding, optional fused
// nearest-2x upsample, optional second source for the UNet skip-concat), so a 3x3 conv is nine
// shifted K-slices accumulating into the same MFMA tile - the im2col matrix is never
// materialised.  W is the weight in its natural [out][K] order (K contiguous), i.e. both MFMA
// operands are "K-contiguous per lane" and are staged identically.
//
// CDNA4 mapping (guides: cdna_hip_programming.md section 5, MI355X_MICROARCH.md LDS):
//   * v_mfma_f32_16x16x32_bf16, 4 waves (256 threads) per workgroup, wave tile (BM/WM)x(BN/WN)
//   * K step 64: global -> registers (16 B/lane, coalesced along K) -> LDS, double buffered, one
//     barrier per K step; the loads of step k+1 are issued before the MFMAs of step k and written
//     to the other LDS buffer after them (async-STAGE split, T14)
//   * LDS rows are 128 B (64 bf16); 16-byte slots are XOR-swizzled with (row & 7) so the
//     ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct slots (T2)
//   * operands are issued swapped (weights as MFMA-A, activations as MFMA-B) so that each lane
//     ends up with 4 consecutive output channels of one output row -> 8-byte epilogue accesses;
//     the transposed-output variant (V^T for attention) uses the natural order instead
//   * XCD-aware tile order: consecutive tiles (same A rows, different N) go to the same XCD L2
//
// Replaces cuBLAS/cuDNN (hipBLASLt/MIOpen) GEMM+conv reached through torch.nn.Linear /
// torch.nn.Conv2d inside the third-party UNet/VAE the reference calls at
// gyre/pipeline/unet/core.py:274 and gyre/pipeline/unified_pipeline.py:309,1531.
#include "gemm_shared.h"
#include <cstdio>
#include <cstdlib>
#include <cstdlib>
#include <atomic>
#include <utility>


// Non-transposed epilogue shared by the 4-wave and 8-wave kernels.  Operands were issued swapped, so a
// lane holds, per 16x16 fragment, output row m = m_base + 16*i + fr and 4 consecutive columns
// n = n_base + 16*j + 4*fq + {0..3}: bias / time-embedding bias / residual / store are 8- or 16-byte accesses.
// 16-byte non-temporal load (data read exactly once: residual rows)
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load_u4(const void* p) {
    const gemm_u32x4 v = __builtin_nontemporal_load((const gemm_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base,
                                              int fr, int fq) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + fr;
        if (m >= p.M) continue;
        const float* rbias = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        if (p.geglu) {
#pragma unroll
            for (int j = 0; j + 1 < NI; j += 2) {
                const int nin = n_base + j * 16 + 4 * fq;  // column in the interleaved 2N space
                if (nin >= p.N) continue;
                float4 bv = p.bias ? *(const float4*)(p.bias + nin) : make_float4(0, 0, 0, 0);
                float4 bg = p.bias ? *(const float4*)(p.bias + nin + 16) : make_float4(0, 0, 0, 0);
                float o[4];
                o[0] = (acc[i][j][0] + bv.x) * gelu_erf_f(acc[i][j + 1][0] + bg.x);
                o[1] = (acc[i][j][1] + bv.y) * gelu_erf_f(acc[i][j + 1][1] + bg.y);
                o[2] = (acc[i][j][2] + bv.z) * gelu_erf_f(acc[i][j + 1][2] + bg.z);
                o[3] = (acc[i][j][3] + bv.w) * gelu_erf_f(acc[i][j + 1][3] + bg.w);
                const int nout = (n_base + j * 16) / 2 + 4 * fq;
                uint2 pk = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldc + nout) = pk;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n_base + j * 16 + 4 * fq;
                if (n >= p.N) continue;
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (p.out_mode == OUT_NCHW) {
                    // tiny-N output conv: out[(b*N + n)*HW + pix], runtime dtype
                    const int hw = p.rows_per_sample;
                    const int b = m / hw, pix = m - b * hw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N) {
                            float v = o[e] + (p.bias ? p.bias[n + e] : 0.f);
                            store_from_f32(p.out, p.out_dtype, ((size_t)b * p.N + n + e) * hw + pix, v);
                        }
                    }
                    continue;
                }
                // N is a multiple of 4 on this path (checked by the launcher)
                if (p.bias) {
                    float4 bv = *(const float4*)(p.bias + n);
                    o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
                }
                if (rbias) {
                    float4 tv = *(const float4*)(rbias + n);
                    o[0] += tv.x; o[1] += tv.y; o[2] += tv.z; o[3] += tv.w;
                }
                if (p.residual) {
                    uint2 rv = *(const uint2*)(p.residual + (size_t)m * p.ldr + n);
                    o[0] += bf16lo(rv.x); o[1] += bf16hi(rv.x); o[2] += bf16lo(rv.y); o[3] += bf16hi(rv.y);
                }
                uint2 pk = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldc + n) = pk;
            }
        }
    }
}

// LDS-staged epilogue of the 8-wave kernel (bf16 row-major output).  The register layout after the swapped
// MFMAs gives every lane 4 consecutive channels of 16 different rows, i.e. 8-byte accesses in 32-byte runs;
// measured, that epilogue cost ~20 us per 42 MB output.  Here each wave parks one 16-row slab of its tile in
// LDS as fp32 (bias / time-embedding bias / GEGLU already applied), reads it back as whole rows and issues
// 16-byte coalesced residual loads and stores.  One rounding, after the residual add - same as the direct path.
// Folded LayerNorm (GemmParams::ln_colsum): out = rstd * acc - (rstd * mean) * colsum[n] + bias[n]
template <bool LNF>
__device__ __forceinline__ float4 ep_affine(const f32x4_t& a, const float4& b, const float4& c, float rstd, float rmu) {
    if (LNF) return make_float4(fmaf(a[0], rstd, fmaf(-rmu, c.x, b.x)), fmaf(a[1], rstd, fmaf(-rmu, c.y, b.y)),
                                fmaf(a[2], rstd, fmaf(-rmu, c.z, b.z)), fmaf(a[3], rstd, fmaf(-rmu, c.w, b.w)));
    return make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
}

// RS: per-row (sum, sum of squares) of this wave tile's rounded outputs -> rs_row[row of the wave tile] (LDS); rs_part = per-wave
// LDS scratch of 16 x (TN / 8) float2 (GemmParams::rowstat_out)
// CS: per-COLUMN (sum, sum of squares) of this wave tile's rounded outputs -> cs_wave[column of the wave tile] (LDS)
// (GemmParams::colstat_out): every lane puts the values it stored (what the consumer will read; zeros past M / N) back into
// the slab position it took them from, then each lane walks the 16 rows of "its" columns - LDS operations of one wave
// complete in order, so no barrier is involved - and keeps the running sums over the wave tile's slabs in registers.
template <int MI, int NI, int TN, bool GG, bool LNF = false, bool RS = false, bool CS = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base,
                                                     int fr, int fq, int lane, float* my, const float* lrstd = nullptr,
                                                     const float* lrmu = nullptr, const float* lcs = nullptr,
                                                     const float* lbb = nullptr, float2* rs_part = nullptr,
                                                     float2* rs_row = nullptr, float2* cs_wave = nullptr) {
    constexpr int TNO_FULL = TN;                 // staged columns per wave without GEGLU
    constexpr bool gg = GG;
    constexpr int tno = gg ? TNO_FULL / 2 : TNO_FULL;    // output columns this wave produces
    constexpr int rowf = TNO_FULL + 4;                     // floats per staged row (+16 B pad: conflict-free b128 writes)
    const int n_out_base = gg ? n_base / 2 : n_base;
    const int n_out = gg ? p.N / 2 : p.N;
    constexpr int vec_per_row = tno / 8;
    constexpr int NV = (16 * vec_per_row + 63) / 64;       // row-wise vectors per lane per slab
    constexpr int CSN = CS ? (tno + 63) / 64 : 1;          // columns per lane of the column-statistics walk
    float cs_s[CSN], cs_q[CSN];
#pragma unroll
    for (int c = 0; c < CSN; ++c) { cs_s[c] = 0.f; cs_q[c] = 0.f; }
    // residual vectors of slab i: requested one slab AHEAD (round 5) - slab 0's in front of the loop, slab i+1's right after slab i's
    // accumulators have gone to LDS (those registers are free from there on, so the extra 4 * NV registers cost nothing at the
    // peak: requesting them earlier - tried in round 2 - spilled in the 256x320 kernel).  With one request per slab the epilogue of
    // the 256x320 tile was four dependent memory round trips per wave, 40 KB in flight per CU: 23 of the 48 us of a cold
    // M = 65536, K = N = 320 launch with residual (tools/ring_bench.py ablation bit 2).
    auto res_load = [&](int i, uint4 (&dst)[NV]) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int v = lane + 64 * q;
            const int row = v / vec_per_row, c8 = v - row * vec_per_row;
            const int mm = m_base + i * 16 + row, nn = n_out_base + c8 * 8;
            dst[q] = make_uint4(0, 0, 0, 0);
            if (v < 16 * vec_per_row && mm < p.M && nn < n_out) dst[q] = nt_load_u4(p.residual + (size_t)mm * p.ldr + nn);
        }
    };
    uint4 rres[NV], rnext[NV];
    if (p.residual) res_load(0, rres);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m_base + i * 16 + fr;
        const float* rbias = (p.rowbias && m < p.M) ? p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        if (gg) {
#pragma unroll
            for (int j = 0; j + 1 < NI; j += 2) {
                const int nin = n_base + j * 16 + 4 * fq;
                float4 bv = make_float4(0, 0, 0, 0), bg = make_float4(0, 0, 0, 0), cv = bv, cg = bv;
                if (LNF) {      // column constants of this wave's tile staged in LDS by the kernel (zeros past N)
                    bv = *(const float4*)(lbb + j * 16 + 4 * fq); bg = *(const float4*)(lbb + j * 16 + 16 + 4 * fq);
                    cv = *(const float4*)(lcs + j * 16 + 4 * fq); cg = *(const float4*)(lcs + j * 16 + 16 + 4 * fq);
                } else if (p.bias && nin < p.N) { bv = *(const float4*)(p.bias + nin); bg = *(const float4*)(p.bias + nin + 16); }
                const float4 val = ep_affine<LNF>(acc[i][j], bv, cv, LNF ? lrstd[i] : 0.f, LNF ? lrmu[i] : 0.f);
                const float4 gate = ep_affine<LNF>(acc[i][j + 1], bg, cg, LNF ? lrstd[i] : 0.f, LNF ? lrmu[i] : 0.f);
                float4 o;
                if (p.debug & 8) { o.x = val.x * gate.x; o.y = val.y * gate.y; o.z = val.z * gate.z; o.w = val.w * gate.w; }   // ablation: no GELU
                else {
                o.x = val.x * gelu_erf_f(gate.x);
                o.y = val.y * gelu_erf_f(gate.y);
                o.z = val.z * gelu_erf_f(gate.z);
                o.w = val.w * gelu_erf_f(gate.w);
                }
                *(float4*)(my + fr * rowf + (j / 2) * 16 + 4 * fq) = o;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n_base + j * 16 + 4 * fq;
                float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                if (n < p.N) {
                    if (LNF) {
                        o = ep_affine<true>(acc[i][j], *(const float4*)(lbb + j * 16 + 4 * fq),
                                            *(const float4*)(lcs + j * 16 + 4 * fq), lrstd[i], lrmu[i]);
                    } else if (p.bias) { float4 bv = *(const float4*)(p.bias + n); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
                    if (rbias) { float4 tv = *(const float4*)(rbias + n); o.x += tv.x; o.y += tv.y; o.z += tv.z; o.w += tv.w; }
                }
                *(float4*)(my + fr * rowf + j * 16 + 4 * fq) = o;
            }
        }
        if (p.residual && i + 1 < MI) res_load(i + 1, rnext);
        // read the slab back row-wise: 8 consecutive channels (32 B fp32) per lane -> one 16-byte store
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int v = lane + 64 * q;
            if (v >= 16 * vec_per_row) break;
            const int row = v / vec_per_row, c8 = v - row * vec_per_row;
            const int mm = m_base + i * 16 + row, nn = n_out_base + c8 * 8;
            const float4 lo = *(const float4*)(my + row * rowf + c8 * 8);
            const float4 hi = *(const float4*)(my + row * rowf + c8 * 8 + 4);
            float rs_s = 0.f, rs_q = 0.f;
            float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (mm < p.M && nn < n_out) {
                f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
                if (p.residual) {
                    float r[8];
                    unpack8(rres[q], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += r[e];
                }
                const uint4 pk = pack8(f);
                if (!(p.debug & 16)) *(uint4*)((bf16_t*)p.out + (size_t)mm * p.ldc + nn) = pk;       // (bit4 ablation: no stores)
                if (RS || CS) unpack8(pk, f);     // statistics of what the consumer will read: the rounded values
                if (RS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { rs_s += f[e]; rs_q = fmaf(f[e], f[e], rs_q); }
                }
            }
            if (RS) rs_part[v] = make_float2(rs_s, rs_q);       // v = row * vec_per_row + c8
            if (CS) {
                *(float4*)(my + row * rowf + c8 * 8) = make_float4(f[0], f[1], f[2], f[3]);
                *(float4*)(my + row * rowf + c8 * 8 + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
        }
        if (CS) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < CSN; ++c) {
                const int col = lane + 64 * c;
                if (col < tno) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float t = my[r * rowf + col]; cs_s[c] += t; cs_q[c] = fmaf(t, t, cs_q[c]); }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (RS) {
            // LDS operations of one wave complete in order: the 16 row sums below see this slab's partials
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) {
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < vec_per_row; ++c) { const float2 t = rs_part[lane * vec_per_row + c]; su += t.x; sq += t.y; }
                rs_row[i * 16 + lane] = make_float2(su, sq);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (p.residual && i + 1 < MI) {
#pragma unroll
            for (int q = 0; q < NV; ++q) rres[q] = rnext[q];
        }
    }
    if (CS) {
#pragma unroll
        for (int c = 0; c < CSN; ++c) {
            const int col = lane + 64 * c;
            if (col < tno) cs_wave[col] = make_float2(cs_s[c], cs_q[c]);
        }
    }
}

// GEGLU epilogue of the folded-LayerNorm FF1 (no residual, no statistics).  Column pair outer, 16-row slab inner: the column
// constants of a (value, gate) fragment pair are read from LDS once instead of once per slab, and the MI GELU evaluations of a
// pair are independent chains that cover each other's ds_read / transcendental latency (in the slab-outer form each chain
// waited for its own constants: tools/ff1_ablation.py, the GELU arithmetic alone cost 52 of the 210 us of the 64x64 FF1).
// Every slab of the wave tile has its own rows in LDS (MI * 16 rows of TN / 2 + 4 floats per wave: gemm_geglu_lnf_lds_bytes),
// read back row-wise after the last pair.
template <int MI, int NI, int TN>
__device__ __forceinline__ void gemm_epilogue_geglu_lnf(const GemmParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base,
                                                        int fr, int fq, int lane, float* my, const float* lrstd,
                                                        const float* lrmu, const float* lcs, const float* lbb) {
    constexpr int tno = TN / 2, rowf = tno + 4, vec_per_row = tno / 8;
    constexpr int NV = (16 * vec_per_row + 63) / 64;
    const int n_out_base = n_base / 2, n_out = p.N / 2;
#pragma unroll
    for (int j = 0; j + 1 < NI; j += 2) {
        const float4 bv = *(const float4*)(lbb + j * 16 + 4 * fq), bg = *(const float4*)(lbb + j * 16 + 16 + 4 * fq);
        const float4 cv = *(const float4*)(lcs + j * 16 + 4 * fq), cg = *(const float4*)(lcs + j * 16 + 16 + 4 * fq);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float4 val = ep_affine<true>(acc[i][j], bv, cv, lrstd[i], lrmu[i]);
            const float4 gate = ep_affine<true>(acc[i][j + 1], bg, cg, lrstd[i], lrmu[i]);
            float4 o;
            if (p.debug & 8) { o.x = val.x * gate.x; o.y = val.y * gate.y; o.z = val.z * gate.z; o.w = val.w * gate.w; }   // ablation: no GELU
            else {
                o.x = val.x * gelu_erf_f(gate.x);
                o.y = val.y * gelu_erf_f(gate.y);
                o.z = val.z * gelu_erf_f(gate.z);
                o.w = val.w * gelu_erf_f(gate.w);
            }
            *(float4*)(my + (i * 16 + fr) * rowf + (j / 2) * 16 + 4 * fq) = o;
        }
    }
    // LDS operations of one wave complete in order: the read-back below sees the writes above
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int v = lane + 64 * q;
            if (v >= 16 * vec_per_row) break;
            const int row = v / vec_per_row, c8 = v - row * vec_per_row;
            const int mm = m_base + i * 16 + row, nn = n_out_base + c8 * 8;
            const float4 lo = *(const float4*)(my + (i * 16 + row) * rowf + c8 * 8);
            const float4 hi = *(const float4*)(my + (i * 16 + row) * rowf + c8 * 8 + 4);
            if (mm < p.M && nn < n_out && !(p.debug & 16)) {
                const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                *(uint4*)((bf16_t*)p.out + (size_t)mm * p.ldc + nn) = pack8(f);
            }
        }
    }
}
// LDS the form above needs for a BM x BN tile of 8 waves (slabs + the column constants behind them)
static inline size_t gemm_geglu_lnf_lds_bytes(int BM, int BN, int WM, int WN) {
    const int TM = BM / WM, TN = BN / WN;
    return (size_t)8 * TM * (TN / 2 + 4) * 4 + (size_t)2 * BN * 4;
}

// Transposed variant for the V columns of a fused Q|K|V projection: the 16-row slab is read back column-wise, 8
// consecutive tokens of one channel per lane -> one 16-byte store into V^T[b][channel][token].
template <int MI, int NI, int TN, bool LNF = false>
__device__ __forceinline__ void gemm_epilogue_staged_t(const GemmParams& p, f32x4_t (&acc)[MI][NI], int m_base, int n_base,
                                                       int fr, int fq, int lane, float* my, const float* lrstd = nullptr,
                                                       const float* lrmu = nullptr, const float* lcs = nullptr,
                                                       const float* lbb = nullptr) {
    constexpr int rowf = TN + 4;
    const int cv_total = p.N - p.vt_col0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n_base + j * 16 + 4 * fq;
            float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (LNF) {
                o = ep_affine<true>(acc[i][j], *(const float4*)(lbb + j * 16 + 4 * fq), *(const float4*)(lcs + j * 16 + 4 * fq),
                                    lrstd[i], lrmu[i]);
            } else if (p.bias && n < p.N) { float4 bv = *(const float4*)(p.bias + n); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
            *(float4*)(my + fr * rowf + j * 16 + 4 * fq) = o;
        }
#pragma unroll
        for (int v = lane; v < TN * 2; v += 64) {
            const int col = v >> 1, half = v & 1;
            const int m = m_base + i * 16 + half * 8, n = n_base + col;
            if (m < p.M && n < p.N) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = my[(half * 8 + e) * rowf + col];
                const int b = m / p.tokens_per_batch, t = m - b * p.tokens_per_batch;
                *(uint4*)(p.vt_out + ((size_t)b * cv_total + (n - p.vt_col0)) * p.ldt + t) = pack8(f);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// 8-wave, LDS-DMA staged variant for the big problems (tiles BM x BN with BN = 320 or 256).
// (A persistent variant that issued the next tile's first K step before the epilogue was built and measured: no
// gain - vmcnt retires in order on CDNA, so a wave's pending epilogue stores gate its next load wait, and with one
// workgroup per CU nothing else can run under the store drain.)
// The 4-wave 128x128 kernel above tops out near 750 TFLOP/s because its operand traffic
// (64 FLOP per byte fetched into LDS) saturates the L2->LDS path (~11-12 TB/s measured); every SD
// UNet width is a multiple of 320, so a 256x320 tile (142 FLOP/B) reads each activation row once per
// tap for the 320-wide layers.  Staging uses global_load_lds (16 B per lane straight into LDS, no
// VGPR round trip and no ds_write pass); the LDS image is lane-linear per wave instruction (8 rows x
// 128 B), so the XOR swizzle is applied to the per-lane SOURCE address and mirrored on the ds_read
// (guide rule 21).  Zero padding (conv halo, M/N/K tails) is fetched from a 256-byte zero page.
// ------------------------------------------------------------------------------------------------
// LNF: folded LayerNorm (GemmParams::ln_colsum / ln_stats) applied by the staged epilogue
// RS: also emit per-row partial statistics of the output tile (GemmParams::rowstat_out)
// CS: also emit per-column-unit partial statistics of the output tile (GemmParams::colstat_out)
template <int BM, int BN, int WM, int WN, int MODE, bool UNIFORM_TAP, bool LNF = false, bool RS = false, bool CS = false>
__global__ __launch_bounds__(512) void k_gemm8(GemmParams p, int tiles_m, int tiles_n, int splits, int nst) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int AR = BM / 64, BR = (BN + 63) / 64;  // 64-row staging granules per K step (A, B); the last B granule may be partial
    constexpr int STAGE_BYTES = (BM + BR * 64) * 128;
    static_assert(WM * WN == 8 && BM % 64 == 0 && BN % 16 == 0, "8 waves, 64-row staging granules");
    // the pair-outer GEGLU epilogue of the folded-LayerNorm form needs a slab per 16-row fragment (gemm_geglu_lnf_lds_bytes)
    constexpr bool GG_SLABS = LNF && NI % 2 == 0 && (size_t)8 * TM * (TN / 2 + 4) * 4 + (size_t)2 * BN * 4 <= 160 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];

    // split-K: the K range is cut into `splits` contiguous slices; slice s of a tile is block s*ntiles + tile
    const int ntiles = tiles_m * tiles_n;
    const int split = blockIdx.x / ntiles;
    const int bid = xcd_tile_id(blockIdx.x - split * ntiles, ntiles);
    // Within an XCD chunk, run fastest along the dimension whose operand is SMALL so that the big operand is shared by
    // neighbouring workgroups in that L2.  Usually the activations are the big one (tn fastest: same A rows).  In the
    // deep UNet levels the weights dominate (8x8 level: 1024 rows of 1280 channels vs 29 MB of weights): there tm
    // runs fastest, so the workgroups of one XCD stream the same weight slice together (measured +14-17 % on the
    // 8x8 convs, +1 % at 16x16; debug bit 8 forces the default order).
    const bool tm_fast = (long)p.M < (MODE == GEMM_CONV3 ? 9L : 1L) * p.N && !(p.debug & 0x100);
    const int tn = tm_fast ? bid / tiles_m : bid % tiles_n, tm = tm_fast ? bid % tiles_m : bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r0 = tid >> 3;                 // 0..63: row inside a 64-row staging granule
    const int kvs = (tid & 7) ^ (r0 & 7);    // global k-vector this lane fetches into LDS slot (tid & 7)

    int a_base[AR], a_y0[AR], a_x0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int row = m0 + r0 + 64 * i;
        bool ok = row < p.M;
        if (MODE == GEMM_LINEAR) {
            a_base[i] = ok ? row : -1;
            a_y0[i] = 0; a_x0[i] = 0;
        } else {
            int hw = p.Ho * p.Wo;
            int n = row / hw, rem = row - n * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = ok ? n * p.Hi * p.Wi : -1;
            a_y0[i] = oy * p.stride - p.pad;
            a_x0[i] = ox * p.stride - p.pad;
        }
    }
    const int Hlim = p.ups ? (p.Hup ? p.Hup : 2 * p.Hi) : p.Hi, Wlim = p.ups ? (p.Wup ? p.Wup : 2 * p.Wi) : p.Wi;
    const bf16_t* zero = p.zero_page;
    // per-sample weights (GemmParams::w_sample_stride: a GroupNorm folded into this layer): the tile's rows belong to ONE sample
    const bf16_t* wbase = p.W + (p.w_sample_stride ? (size_t)(m0 / p.rows_per_sample) * p.w_sample_stride : (size_t)0);

    auto issue_stage = [&](int kc, int s) {
        char* sbase = smem_raw + s * STAGE_BYTES + wave * 1024;
        const int k = kc * BK + kvs * 8;
        const bool kok = k < p.K;
        if (MODE == GEMM_LINEAR) {
            const bool first = k < p.C1;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bf16_t* src = zero;
                if (kok && a_base[i] >= 0)
                    src = first ? p.A + (size_t)a_base[i] * p.lda + k : p.A2 + (size_t)a_base[i] * p.lda2 + (k - p.C1);
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sbase + i * 8192), 16, 0, 0);
            }
        } else {
            int tap, c;
            if (UNIFORM_TAP) {
                // K order = channel chunk outer, tap inner: the nine taps of one 64-channel chunk run back to
                // back, so the (shifted) input rows they share are re-read from L1/L2 instead of the fabric
                const int chunk = kc / 9;
                tap = kc - chunk * 9;
                c = chunk * BK + kvs * 8;
            } else {
                tap = k / p.Cin;
                c = k - tap * p.Cin;
            }
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool first = c < p.C1;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bf16_t* src = zero;
                int iy = a_y0[i] + ky, ix = a_x0[i] + kx;
                if (kok && a_base[i] >= 0 && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    size_t pix = (size_t)a_base[i] + (size_t)iy * p.Wi + ix;
                    src = first ? p.A + pix * p.lda + c : p.A2 + pix * p.lda2 + (c - p.C1);
                }
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sbase + i * 8192), 16, 0, 0);
            }
        }
        int kw = k;  // position of this lane's 8 weights inside a [taps][Cin] weight row
        if (MODE == GEMM_CONV3 && UNIFORM_TAP) {
            const int chunk = kc / 9, tap = kc - chunk * 9;
            kw = tap * p.Cin + chunk * BK + kvs * 8;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + r0 + 64 * i;
            const bf16_t* src = zero;
            if (kok && n < p.N && (BN % 64 == 0 || r0 + 64 * i < BN))
                src = p.W_blk ? p.W_blk + ((size_t)(n >> 3) * (p.K >> 6) + (kw >> 6)) * 512 + (n & 7) * 64 + (kw & 63) : wbase + (size_t)n * p.K + kw;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(sbase + BM * 128 + i * 8192), 16, 0, 0);
        }
    };

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (p.K + BK - 1) / BK;
    const int nk_per = (nk_all + splits - 1) / splits;
    const int kc0 = split * nk_per;
    const int nk = min(nk_all, kc0 + nk_per);
    const int fr = lane & 15, fq = lane >> 4;
    // one 64-deep K step out of ring slot `slot`
    // (Round 5: issuing all fragment reads of a k32 sub-step - or of the whole step - ahead of its MFMAs, pinned with sched_barrier, was
    //  measured on the 128x160 tile: -1 us of 23 at one workgroup per CU, but 159 registers instead of 97 - the tile's two workgroups
    //  per CU are gone (M = 16384: 70 -> 85 us) and the 128x320 tile spills.  With the loads switched off the 20-step loop still
    //  takes 11 us against 5.3 us of MFMA issue: 14 ds_read_b128 per wave and step = 112 KB per CU and step is 437 cycles of the
    //  LDS array beside 640 of MFMA - the 32x80 wave tile reads twice the bytes per FLOP the 64x160 one does.  tools/floor_probe.py)
    auto k_step = [&](int slot) {
        const uint4* a = (const uint4*)(smem_raw + slot * STAGE_BYTES);
        const uint4* b = a + BM * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // (explicit 2-deep fragment prefetch pinned with sched_group_barrier was measured 3-6 % slower: with two
            // waves per SIMD the partner wave already covers the ds_read latency)
            bf16x8_t af[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int r = wm * TM + i * 16 + fr;
                af[i] = __builtin_bit_cast(bf16x8_t, a[r * 8 + ((ks * 4 + fq) ^ (r & 7))]);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int r = wn * TN + j * 16 + fr;
                bf16x8_t bf = __builtin_bit_cast(bf16x8_t, b[r * 8 + ((ks * 4 + fq) ^ (r & 7))]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i][j] = GYRE_MFMA_16x16x32(bf, af[i], acc[i][j], 0, 0, 0);
            }
        }
    };
    bool ring_done = false;
    // (conv ring: the 128x160 tile and the VAE's 256x128 one - three 48 KB stages fit a CU; the 256x320 instance spills with it, the
    //  256x256 one has no room for a third stage)
    if constexpr (MODE == GEMM_LINEAR || (UNIFORM_TAP && ((BM == 128 && BN == 160) || (BM == 256 && BN == 128)))) {
        // Round 5: the K loop of the linear problems as an nst-deep LDS ring (nst = 2 ... 4, chosen by the launcher from the LDS a
        // workgroup may take), nst - 1 stages IN FLIGHT while one is multiplied.  The two-stage loop below requests stage k+1,
        // multiplies stage k and then waits for everything: one exposed memory latency per 64-deep K step - with the weights of a
        // 16x16-level projection cold in HBM that was ~1.3 us per step against 0.27 us of MFMA work (M = 4096, N = K = 1280: 20
        // steps, 32 us for 5.4 us of matrix work).  tools/ubench/l2_bw.hip (profiles/r05_l2_bw.txt) shows what the load path can
        // do when fed: 125 - 137 GB/s per CU from L2 by LDS-DMA, 30 from the Infinity Cache, 24 - 27 from HBM.
        //   * requests are inline asm (hipcc would make the next ds_read wait for a builtin LDS-DMA), one running 64-bit source
        //     pointer per request slot (+128 B per step; rows / weight rows past the end: the zero page, step 0);
        //   * counted s_waitcnt vmcnt((nst - 2) * requests per wave) + ONE barrier per step; s_waitcnt lgkmcnt(0) in front of the
        //     barrier: the slot refilled right behind it is the one the previous step read (every fragment read has returned
        //     before any wave can request the refill - the ring discipline of kernels_attn.hip);
        //   * past-the-end stages are requested from the zero page so that the in-flight count stays uniform.
        // Same fragment reads, same K order, same accumulators: bit-identical to the two-stage loop (tuning bit 24 = that loop).
        // Late round 5: the same ring for the 3x3 convolutions whose K steps are (64-channel chunk, tap) pairs (UNIFORM_TAP: every UNet /
        // VAE conv but conv_in): the split-K convs of the 16x16 / 8x8 levels run 256 workgroups of ~23 K steps, one per CU - the
        // shape the two-stage loop serves worst.  The gather has no running pointer (tap shifts, padding from the zero page), so a
        // step's sources are recomputed from its (chunk, tap) exactly as issue_stage does; tuning bit 27 = two-stage loop for convs.
        if (nst >= 2 && p.K % BK == 0 && p.C1 % BK == 0 && !(p.debug & (MODE == GEMM_LINEAR ? 0x1000003 : 0x8000003))) {
            constexpr int PPW = AR + BR;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
            auto dma16 = [&](unsigned dst, const char* src) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
            };
            const int k1 = p.C1 / BK;                                 // first K step of the second source (= all steps for one source)
            const char* ca[AR]; const char* cw[BR];
            unsigned ia[AR], iw[BR];
            auto a_ptr = [&](int i, bool second, int kstep) -> const char* {
                const int row = a_base[i];
                return second ? (const char*)(p.A2 + (size_t)row * p.lda2 + kvs * 8) + (size_t)(kstep - k1) * 128
                              : (const char*)(p.A + (size_t)row * p.lda + kvs * 8) + (size_t)kstep * 128;
            };
            if constexpr (MODE == GEMM_LINEAR) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = a_base[i] >= 0;
                ca[i] = ok ? a_ptr(i, kc0 >= k1, kc0) : (const char*)zero;
                ia[i] = ok ? 128u : 0u;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int n = n0 + r0 + 64 * i;
                const bool ok = n < p.N && (BN % 64 == 0 || r0 + 64 * i < BN);
                if (p.W_blk) {       // blocked copy: the 8 rows x 128 B of a request are one KiB, the next K step the next KiB
                    cw[i] = ok ? (const char*)(p.W_blk + ((size_t)(n >> 3) * (p.K >> 6) + kc0) * 512 + (n & 7) * 64 + kvs * 8) : (const char*)zero;
                    iw[i] = ok ? 1024u : 0u;
                } else {
                    cw[i] = ok ? (const char*)(wbase + (size_t)n * p.K + kvs * 8) + (size_t)kc0 * 128 : (const char*)zero;
                    iw[i] = ok ? 128u : 0u;
                }
            }
            }
            // (Round 6: an L2 PREFETCH beside the ring was built and measured - one dword load per wave and K step whose lanes touch the
            //  eight 128-byte lines of each of the wave's DMA pieces six steps ahead, counted into the vmcnt waits; bit-identical, and
            //  SLOWER on every 128x160 shape: M = 4096, N = K = 1280 30.3 -> 35.7 us cold / 29.5 -> 33.1 warm, M = 16384, K = N = 640
            //  30.9 -> 38.1, K = 2560 75.8 -> 103.9, UNet call 17.20 -> 17.59 ms (profiles/r06_l2_prefetch_ab.txt).  The loop is not
            //  waiting for first-touch HBM latency: a K step moves 36.9 KB through the CU's vector-memory path, 576 cycles at its
            //  64 B per clock beside 640 cycles of MFMA - the path is as busy as the matrix pipe, and 40 more line requests per wave
            //  and step queue in front of the operand requests.  What bounds this tile is operand bytes per FLOP through that path
            //  (profiles/HISTORY.md 4c), which neither a prefetch nor a split of K inside the workgroup changes.
            //  Second experiment: the refill's request pieces issued FROM INSIDE the K step, one behind every second MFMA group (the
            //  way the pipelined kernel places its requests), instead of as a block behind the barrier - on the theory that 5 pieces x
            //  60 - 185 cycles of issue cost per wave were serialised in front of the MFMAs.  ISA as intended, bit-identical, and no
            //  gain: 30.1 -> 29.9 / 29.6 -> 30.8 us on the two 13.4-GFLOP shapes, the 128x160 convs 59.5 -> 65.4 and 82.3 -> 93.2 us,
            //  UNet call unchanged (profiles/r06_ring_issue_interleave_ab.txt).  Removed.)
            int kabs = kc0;                                           // K step the next request fetches
            auto ring_issue = [&](int slot) {
                const unsigned dst = lds0 + slot * STAGE_BYTES + wave * 1024;
                if (kabs < nk) {
                    if constexpr (MODE == GEMM_LINEAR) {
                        if (kabs == k1 && kabs > kc0) {               // (wave-uniform, at most once) the rows continue in the second source
#pragma unroll
                            for (int i = 0; i < AR; ++i) if (a_base[i] >= 0) ca[i] = a_ptr(i, true, kabs);
                        }
#pragma unroll
                        for (int i = 0; i < AR; ++i) { dma16(dst + i * 8192, ca[i]); ca[i] += ia[i]; }
#pragma unroll
                        for (int i = 0; i < BR; ++i) { dma16(dst + BM * 128 + i * 8192, cw[i]); cw[i] += iw[i]; }
                    } else {
                        // issue_stage's conv gather for K step kabs = (chunk, tap), chunk outer / tap inner
                        const int chunk = kabs / 9, tap = kabs - chunk * 9;
                        const int c = chunk * BK + kvs * 8;
                        const int ky = tap / 3, kx = tap - ky * 3;
                        const bool first = c < p.C1;
#pragma unroll
                        for (int i = 0; i < AR; ++i) {
                            const bf16_t* src = zero;
                            int iy = a_y0[i] + ky, ix = a_x0[i] + kx;
                            if (a_base[i] >= 0 && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim) {
                                if (p.ups) { iy >>= 1; ix >>= 1; }
                                const size_t pix = (size_t)a_base[i] + (size_t)iy * p.Wi + ix;
                                src = first ? p.A + pix * p.lda + c : p.A2 + pix * p.lda2 + (c - p.C1);
                            }
                            dma16(dst + i * 8192, (const char*)src);
                        }
                        const int kw = tap * p.Cin + c;              // this lane's 8 weights inside a [taps][Cin] weight row
#pragma unroll
                        for (int i = 0; i < BR; ++i) {
                            const int n = n0 + r0 + 64 * i;
                            const bf16_t* src = zero;
                            if (n < p.N && (BN % 64 == 0 || r0 + 64 * i < BN))
                                src = p.W_blk ? p.W_blk + ((size_t)(n >> 3) * (p.K >> 6) + (kw >> 6)) * 512 + (n & 7) * 64 + (kw & 63)
                                              : wbase + (size_t)n * p.K + kw;
                            dma16(dst + BM * 128 + i * 8192, (const char*)src);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < PPW; ++i) dma16(dst + i * 8192, (const char*)zero);
                }
                ++kabs;
            };
            int fill = 0;                                             // slot the next request fills
            for (int s = 0; s + 1 < nst; ++s) { ring_issue(fill); ++fill; }
            int slot = 0;
            for (int kc = kc0; kc < nk; ++kc) {
                // stage kc has landed for this wave (the nst - 2 younger ones may be in flight), then - barrier - for every wave
                if (nst == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else if (nst == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
                __builtin_amdgcn_s_barrier();
                ring_issue(fill);                                     // the slot stage kc - 1 was read from
                fill = fill + 1 == nst ? 0 : fill + 1;
                k_step(slot);
                slot = slot + 1 == nst ? 0 : slot + 1;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-page requests past the end: nothing may land after this
            __syncthreads();
            ring_done = true;
        }
    }
    if (!ring_done) {
        issue_stage(kc0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kc = kc0; kc < nk; ++kc) {
            const int cur = (kc - kc0) & 1;
            if (kc + 1 < nk && !(p.debug & 1)) issue_stage(kc + 1, cur ^ 1);
            if (!(p.debug & 2)) k_step(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (!LNF && !RS && !CS && splits > 1) {     // (the folded-LayerNorm / statistics forms are launched unsplit, staged epilogue only)
        // fp32 partial slab of this K slice; bias / residual / rounding happen once in k_splitk_reduce
        float* slab = p.splitk_ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * TM + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * TN + j * 16 + 4 * fq;
                if (n < p.N) *(float4*)(slab + (size_t)m * p.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    if (!LNF && !RS && !CS && (p.debug & 4)) {  // tuning ablation: no epilogue (keep the accumulators alive with one conditional store)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sum == 12345.678f) ((float*)p.out)[0] = sum;
        return;
    }
    // coalesced LDS-staged epilogue when every 8-channel group is 16-byte addressable, else the direct one
    const int n_out = p.geglu ? p.N / 2 : p.N;
    if (LNF || RS || CS || (p.out_mode == OUT_BF16 && (n_out % 8) == 0 && (p.ldc % 8) == 0 && (!p.residual || (p.ldr % 8) == 0) &&
                (((size_t)p.out | (size_t)p.residual) & 15) == 0)) {
        float* my = (float*)smem_raw + wave * (16 * (TN + 4));   // the operand ring is dead after the last barrier
        float lrstd[MI], lrmu[MI];
        const float *lcs = nullptr, *lbb = nullptr;
        if constexpr (LNF) {
            // column constants of the tile (colsum, folded bias) go to LDS behind the epilogue slabs: the epilogue reads them
            // with ds_read_b128 instead of holding dozens of global loads in flight next to 160 live accumulators
            // (the GEGLU form keeps a slab per 16-row fragment: its constants sit behind MI slabs of TN / 2 + 4 floats per wave)
            const bool gg_slabs = GG_SLABS && p.geglu && !p.residual && !(p.debug & 0x80000);
            float* colc = (float*)smem_raw + (gg_slabs ? 8 * TM * (TN / 2 + 4) : 8 * 16 * (TN + 4));
            for (int t = tid; t < BN; t += 512) {
                const int n = n0 + t;
                colc[t] = n < p.N ? p.ln_colsum[n] : 0.f;
                colc[BN + t] = n < p.N ? p.bias[n] : 0.f;
            }
            __syncthreads();
            lcs = colc + wn * TN; lbb = colc + BN + wn * TN;
            // per-row statistics of the rows this lane's accumulators belong to: from the separate pass
            // (launch_layernorm_stats) or from the partial sums the producing GEMM left (rowstat_out of that launch)
            const float invk = 1.0f / (float)p.K;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * TM + i * 16 + fr;
                float2 rs = make_float2(1.f, 0.f);
                if (m < p.M) {
                    if (p.ln_nparts > 0) {
                        float su = 0.f, sq = 0.f;
                        for (int t = 0; t < p.ln_nparts; ++t) {
                            const float2 v = ((const float2*)p.ln_parts)[(size_t)t * p.M + m];
                            su += v.x; sq += v.y;
                        }
                        const float mean = su * invk;
                        const float rstd = 1.0f / sqrtf(fmaxf(sq * invk - mean * mean, 0.f) + p.ln_eps);
                        rs = make_float2(rstd, rstd * mean);
                    } else {
                        rs = ((const float2*)p.ln_stats)[m];
                    }
                }
                lrstd[i] = rs.x; lrmu[i] = rs.y;
            }
        }
        if constexpr (RS) {
            // wave tile rows -> LDS [WN][BM] behind the slabs and the per-wave scratch; after the block barrier one thread per
            // row adds the WN wave tiles in order and writes the tile's partial
            float2* rs_part = (float2*)((float*)smem_raw + 8 * 16 * (TN + 4)) + wave * (16 * (TN / 8));
            float2* rs_all = (float2*)((float*)smem_raw + 8 * 16 * (TN + 4)) + 8 * (16 * (TN / 8));
            gemm_epilogue_staged<MI, NI, TN, false, false, true>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane, my, nullptr, nullptr,
                                                                   nullptr, nullptr, rs_part, rs_all + wn * BM + wm * TM);
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) { const float2 t = rs_all[w * BM + tid]; su += t.x; sq += t.y; }
                ((float2*)p.rowstat_out)[(size_t)tn * p.M + m0 + tid] = make_float2(su, sq);
            }
            return;
        }
        if constexpr (CS) {
            // wave tile column sums -> LDS [8 waves][TN] behind the slabs; after the block barrier one thread per tile column adds
            // the WM wave tiles in order, then one thread per unit adds its channels and writes the tile's partial
            float2* cs_all = (float2*)((float*)smem_raw + 8 * 16 * (TN + 4));
            float2* chan = cs_all + 8 * TN;
            gemm_epilogue_staged<MI, NI, TN, false, false, false, true>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane, my, nullptr, nullptr,
                                                                          nullptr, nullptr, nullptr, nullptr, cs_all + wave * TN);
            __syncthreads();
            for (int t = tid; t < BN; t += 512) {
                const int wn_ = t / TN, col = t - wn_ * TN;
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { const float2 v = cs_all[(w * WN + wn_) * TN + col]; su += v.x; sq += v.y; }
                chan[t] = make_float2(su, sq);
            }
            __syncthreads();
            const int unit = p.colstat_unit;
            for (int u = tid; u * unit < BN; u += 512) {
                const int n = n0 + u * unit;
                if (n < p.N) {
                    float su = 0.f, sq = 0.f;
                    for (int c = 0; c < unit; ++c) { const float2 v = chan[u * unit + c]; su += v.x; sq += v.y; }
                    ((float2*)p.colstat_out)[(size_t)tm * (p.N / unit) + n / unit] = make_float2(su, sq);
                }
            }
            return;
        }
        if (p.vt_out && n0 + wn * TN >= p.vt_col0) {             // V columns of a fused Q|K|V projection (wave-uniform)
            gemm_epilogue_staged_t<MI, NI, TN, LNF>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane, my, lrstd, lrmu, lcs, lbb);
            return;
        }
        if constexpr (NI % 2 == 0) {
            if constexpr (LNF && GG_SLABS) {
                if (p.geglu && !p.residual && !(p.debug & 0x80000)) {      // (bit 19: the slab-outer form, for A/B runs)
                    gemm_epilogue_geglu_lnf<MI, NI, TN>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane,
                                                        (float*)smem_raw + wave * (TM * (TN / 2 + 4)), lrstd, lrmu, lcs, lbb);
                    return;
                }
            }
            if (p.geglu) { gemm_epilogue_staged<MI, NI, TN, true, LNF>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane, my, lrstd, lrmu, lcs, lbb); return; }
        }
        gemm_epilogue_staged<MI, NI, TN, false, LNF>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq, lane, my, lrstd, lrmu, lcs, lbb);
        return;
    }
    if constexpr (!LNF && !RS && !CS) gemm_epilogue<MI, NI>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq);
}

// out = bf16( sum_s slab[s] + bias + rowbias + residual ), fixed summation order (deterministic)
// (Round 5: the same reduction INSIDE the GEMM launch - per-tile arrival counters, agent-scope release by every K slice, the last
//  arriver adds the slabs row-wise in slice order and finishes the tile: cdna_hip_programming.md guideline 16, counter form - was
//  built for k_gemm8 and the pipelined 256x320 tile, bit-identical to this kernel (12 repetitions under concurrent traffic per
//  config), and measured: UNet call 17.50 -> 18.37 ms at batch 16 (338 -> 306 launches), 5.89 -> 6.60 ms at batch 2.  A tile's last
//  arriver reads splits x 328 KB alone (64 workgroups finish what this kernel spreads over 256 CUs) behind a write-back of
//  freshly dirtied L2 lines; the guide prices the form as worth it up to "a few tens of KB" per tile.  Removed again.)
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmParams p, int splits) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 consecutive columns
    const int nq = p.N / 4;
    if (q >= (size_t)p.M * nq) return;
    const int m = (int)(q / nq), n = (int)(q % nq) * 4;
    // Round 5: every request of the thread is in flight before the first add (the plain loop compiled to load / s_waitcnt vmcnt(0) /
    // add per slab, then the same for bias, row bias and residual: up to seven dependent round trips per thread, which only a
    // full chip of resident waves hides - at batch 1 - 2 these launches were pure latency).  Slabs four at a time, same order of adds.
    float4 bv, rbv;                                  // no initialisers: a zero on the not-taken side is a register write the
    uint2 rv;                                        // compiler orders behind the requests already in flight (s_waitcnt vmcnt(0))
    if (p.residual) rv = *(const uint2*)(p.residual + (size_t)m * p.ldr + n);
    if (p.bias) bv = *(const float4*)(p.bias + n);
    if (p.rowbias) rbv = *(const float4*)(p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias + n);
    const float* src = p.splitk_ws + (size_t)m * p.N + n;
    const size_t slab = (size_t)p.M * p.N;
    float4 a = make_float4(0, 0, 0, 0);
    for (int s0 = 0; s0 < splits; s0 += 4) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s0 + j < splits) v[j] = *(const float4*)(src + (size_t)(s0 + j) * slab);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s0 + j < splits) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    if (p.bias) { a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w; }
    if (p.rowbias) { a.x += rbv.x; a.y += rbv.y; a.z += rbv.z; a.w += rbv.w; }
    if (p.residual) { a.x += bf16lo(rv.x); a.y += bf16hi(rv.x); a.z += bf16lo(rv.y); a.w += bf16hi(rv.y); }
    *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldc + n) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
}

// The same reduction for a tensor whose consumer is a GroupNorm (GemmParams::colstat_out): a workgroup owns CS_RED_ROWS
// consecutive rows (never straddling a sample) x CB consecutive columns (whole units), a thread 8 consecutive columns of every
// PY-th row, so the per-channel sums of the ROUNDED outputs accumulate in registers; they are folded rows -> channels -> units
// through LDS in a fixed order and leave as one partial per (row block, unit).  Grid = row blocks x column blocks: as many
// workgroups as the plain reduction has at these sizes (a one-dimensional grid of row blocks left the chip idle at small M).
// CS_RED_ROWS rows per block, or 64 where a sample has more than 64 such blocks (the consumer's prologue adds a sample's row
// blocks serially: model_impl.h attach_colstats takes at most 64 per sample - the 64x64 level at small batch).
#define CS_RED_ROWS 16
static inline int cs_red_rows(int rows_per_sample) {
    return (rows_per_sample > 64 * CS_RED_ROWS && rows_per_sample % 64 == 0) ? 64 : CS_RED_ROWS;
}
__global__ __launch_bounds__(256) void k_splitk_reduce_cs(GemmParams p, int splits, int CB, int TX, int PY, int R) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* red = (float2*)smem_raw;             // [PY][CB]
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int m0 = blockIdx.x * R, nb = blockIdx.y * CB;
    if (ty < PY && tx * 8 < CB) {
        const int n = nb + tx * 8;
        float s[8], ss[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
        float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
            bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
        }
        for (int r = ty; r < R; r += PY) {
            const int m = m0 + r;
            if (m >= p.M) break;
            // all requests of the row first (residual, row bias, slabs four at a time), then the adds in the original order
            uint4 rraw;                              // (no initialisers, see k_splitk_reduce)
            float4 b0, b1;
            if (p.residual) rraw = *(const uint4*)(p.residual + (size_t)m * p.ldr + n);
            if (p.rowbias) {
                const float* rb = p.rowbias + (size_t)(m / p.rows_per_sample) * p.ld_rowbias + n;
                b0 = *(const float4*)rb; b1 = *(const float4*)(rb + 4);
            }
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* src = p.splitk_ws + (size_t)m * p.N + n;
            const size_t slab = (size_t)p.M * p.N;
            for (int s0 = 0; s0 < splits; s0 += 4) {
                float4 v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s0 + j < splits) {
                        const float* sj = src + (size_t)(s0 + j) * slab;
                        v0[j] = *(const float4*)sj; v1[j] = *(const float4*)(sj + 4);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (s0 + j < splits) {
                        a[0] += v0[j].x; a[1] += v0[j].y; a[2] += v0[j].z; a[3] += v0[j].w;
                        a[4] += v1[j].x; a[5] += v1[j].y; a[6] += v1[j].z; a[7] += v1[j].w;
                    }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += bs[e];
            if (p.rowbias) { a[0] += b0.x; a[1] += b0.y; a[2] += b0.z; a[3] += b0.w; a[4] += b1.x; a[5] += b1.y; a[6] += b1.z; a[7] += b1.w; }
            if (p.residual) {
                float rr[8];
                unpack8(rraw, rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += rr[e];
            }
            const uint4 pk = pack8(a);
            *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldc + n) = pk;
            unpack8(pk, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += a[e]; ss[e] = fmaf(a[e], a[e], ss[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(size_t)ty * CB + tx * 8 + e] = make_float2(s[e], ss[e]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < CB; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int y = 0; y < PY; ++y) { const float2 t = red[(size_t)y * CB + c]; a += t.x; b += t.y; }
        red[c] = make_float2(a, b);              // row 0, own column
    }
    __syncthreads();
    const int unit = p.colstat_unit, nu = p.N / unit;
    for (int u = threadIdx.x; u * unit < CB; u += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int c = u * unit; c < (u + 1) * unit; ++c) { const float2 t = red[c]; a += t.x; b += t.y; }
        ((float2*)p.colstat_out)[(size_t)blockIdx.x * nu + nb / unit + u] = make_float2(a, b);
    }
}
// column block of the reduction: the smallest multiple of lcm(8, unit) that is >= 128 and divides N (N itself otherwise)
static int cs_red_colblock(int N, int unit) {
    int l = unit;
    while (l % 8) l += unit;
    for (int cb = l; cb < N; cb += l)
        if (cb >= 128 && N % cb == 0) return cb;
    return N;
}

int launch_splitk_reduce(hipStream_t st, const GemmParams& p, int splits) {
    // its own class: the GEMM classes' event time (bench.py's live `roofline`) is the GEMM kernel alone, as rocprofv3 sees it
    GyreProfScope prof_(KC_SPLITK_REDUCE, st, 0.0, (double)splits * p.M * p.N * 4.0 + (double)p.M * p.N * 2.0 * (p.residual ? 2.0 : 1.0));
    if (p.colstat_out) {
        const int CB = cs_red_colblock(p.N, p.colstat_unit);
        const int TX = CB / 8;
        const int R = cs_red_rows(p.rows_per_sample);
        int PY = 256 / TX; if (PY < 1) PY = 1; if (PY > R) PY = R;
        const size_t lds = (size_t)PY * CB * sizeof(float2);
        hipLaunchKernelGGL(k_splitk_reduce_cs, dim3((unsigned)((p.M + R - 1) / R), (unsigned)(p.N / CB)), dim3(256), lds, st,
                           p, splits, CB, TX, PY, R);
        GYRE_LAUNCH_CHECK();
        return 0;
    }
    const size_t nthreads = (size_t)p.M * (p.N / 4);
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, p, splits);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// Weight side of the folded LayerNorm (GemmParams::ln_colsum): one wave per output row, fixed summation order.
__global__ __launch_bounds__(256) void k_ln_fold(const bf16_t* __restrict__ W, int N, int K, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ bias,
                                                 bf16_t* __restrict__ Wf, float* __restrict__ colsum, float* __restrict__ bias_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16_t* wrow = W + (size_t)n * K;
    bf16_t* frow = Wf + (size_t)n * K;
    float cs = 0.f, bb = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        float w[8], f[8];
        unpack8(*(const uint4*)(wrow + k), w);
        const float4 g0 = *(const float4*)(gamma + k), g1 = *(const float4*)(gamma + k + 4);
        const float4 b0 = *(const float4*)(beta + k), b1 = *(const float4*)(beta + k + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[j] = w[j] * g[j]; bb = fmaf(b[j], w[j], bb); }
        const uint4 pk = pack8(f);
        *(uint4*)(frow + k) = pk;
        unpack8(pk, f);                       // the column sum is taken over the ROUNDED folded weights the GEMM multiplies by
#pragma unroll
        for (int j = 0; j < 8; ++j) cs += f[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { cs += __shfl_xor(cs, off); bb += __shfl_xor(bb, off); }
    if (lane == 0) { colsum[n] = cs; bias_out[n] = bb + (bias ? bias[n] : 0.f); }
}
int launch_ln_fold(hipStream_t st, const bf16_t* W, int N, int K, const float* gamma, const float* beta, const float* bias,
                   bf16_t* Wf, float* colsum, float* bias_out) {
    if (K % 8) GYRE_FAIL(-1, "ln_fold: K must be a multiple of 8");
    hipLaunchKernelGGL(k_ln_fold, dim3((N + 3) / 4), dim3(256), 0, st, W, N, K, gamma, beta, bias, Wf, colsum, bias_out);
    GYRE_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int WN, int MODE, bool UNIFORM_TAP, bool TRANS>
__global__ __launch_bounds__(256) void k_gemm(GemmParams p, int tiles_m, int tiles_n) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int AR = BM / 32, BR = BN / 32;  // 16-byte vectors per thread per K step (A, B)
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint4* lds = (uint4*)smem_raw;  // stage s: A at s*(BM+BN)*8, B right after A; 8 x uint4 per row
    constexpr int STAGE = (BM + BN) * 8;

    if (p.batch > 1) {   // batched form: one problem per blockIdx.y
        p.A += (size_t)blockIdx.y * p.bsA; p.A2 += (size_t)blockIdx.y * p.bsA; p.W += (size_t)blockIdx.y * p.bsW;
        p.out = (bf16_t*)p.out + (size_t)blockIdx.y * p.bsC;
    }
    // ---- XCD-aware, bijective tile order: tile id -> (tm, tn) with tn fastest -------------
    const int bid = xcd_tile_id(blockIdx.x, tiles_m * tiles_n);
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kv = tid & 7;    // which 16-byte vector of the 128-byte K slice this thread stages
    const int r0 = tid >> 3;   // base row (0..31); rows r0 + 32*i

    // ---- per-row gather state for A ----------------------------------------------------------
    int a_base[AR];   // linear: row index (or -1); conv: sample base pixel index n*Hi*Wi (or -1)
    int a_y0[AR], a_x0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int row = m0 + r0 + 32 * i;
        bool ok = row < p.M;
        if (MODE == GEMM_LINEAR) {
            a_base[i] = ok ? row : -1;
            a_y0[i] = 0; a_x0[i] = 0;
        } else {
            int hw = p.Ho * p.Wo;
            int n = row / hw, rem = row - n * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = ok ? n * p.Hi * p.Wi : -1;
            a_y0[i] = oy * p.stride - p.pad;
            a_x0[i] = ox * p.stride - p.pad;
        }
    }
    const int C2 = p.Cin - p.C1;
    const int Hlim = p.ups ? (p.Hup ? p.Hup : 2 * p.Hi) : p.Hi, Wlim = p.ups ? (p.Wup ? p.Wup : 2 * p.Wi) : p.Wi;

    uint4 ra[AR], rb[BR];
    auto load_stage = [&](int kc) {
        const int k = kc * BK + kv * 8;
        if (MODE == GEMM_LINEAR) {
            const bool kok = k < p.K;
            const bool first = k < p.C1;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kok && a_base[i] >= 0) {
                    const bf16_t* src = first ? p.A + (size_t)a_base[i] * p.lda + k
                                              : p.A2 + (size_t)a_base[i] * p.lda2 + (k - p.C1);
                    v = *(const uint4*)src;
                }
                ra[i] = v;
            }
        } else {
            int tap, c;
            if (UNIFORM_TAP) {
                // same K order as k_gemm8 (64-channel chunk outer, tap inner): every tile config then sums in the
                // same order and gives bit-identical results, so the planner is free to choose by problem size
                const int chunk = kc / 9;
                tap = kc - chunk * 9;       // scalar: same tap for the whole K step
                c = chunk * BK + kv * 8;
            } else {
                tap = k / p.Cin;
                c = k - tap * p.Cin;
            }
            const bool kok = k < p.K;
            const int ky = tap / 3, kx = tap - ky * 3;
            const bool first = c < p.C1;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                int iy = a_y0[i] + ky, ix = a_x0[i] + kx;
                // circular padding (GemmParams::wrap: the reference's tiling option): the one row / column of padding wraps
                // around the (upsampled) image instead of reading zeros
                if (p.wrap & 2) iy = iy < 0 ? iy + Hlim : (iy >= Hlim && iy < 2 * Hlim ? iy - Hlim : iy);
                if (p.wrap & 1) ix = ix < 0 ? ix + Wlim : (ix >= Wlim && ix < 2 * Wlim ? ix - Wlim : ix);
                if (kok && a_base[i] >= 0 && (unsigned)iy < (unsigned)Hlim && (unsigned)ix < (unsigned)Wlim) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    size_t pix = (size_t)a_base[i] + (size_t)iy * p.Wi + ix;
                    const bf16_t* src = first ? p.A + pix * p.lda + c : p.A2 + pix * p.lda2 + (c - p.C1);
                    v = *(const uint4*)src;
                }
                ra[i] = v;
            }
        }
        {
            const bool kok = k < p.K;
            int kw = k;  // position of this lane's 8 weights inside a [taps][Cin] weight row
            if (MODE == GEMM_CONV3 && UNIFORM_TAP) {
                const int chunk = kc / 9, tap = kc - chunk * 9;
                kw = tap * p.Cin + chunk * BK + kv * 8;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                int n = n0 + r0 + 32 * i;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kok && n < p.N) v = *(const uint4*)(p.W + (size_t)n * p.K + kw);
                rb[i] = v;
            }
        }
    };
    auto store_stage = [&](int s) {
        uint4* a = lds + s * STAGE;
        uint4* b = a + BM * 8;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int r = r0 + 32 * i;
            a[r * 8 + (kv ^ (r & 7))] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            int r = r0 + 32 * i;
            b[r * 8 + (kv ^ (r & 7))] = rb[i];
        }
    };

    f32x4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nk) load_stage(kc + 1);
        const uint4* a = lds + cur * STAGE;
        const uint4* b = a + BM * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[MI], bfr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int r = wm * TM + i * 16 + fr;
                af[i] = __builtin_bit_cast(bf16x8_t, a[r * 8 + ((ks * 4 + fq) ^ (r & 7))]);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int r = wn * TN + j * 16 + fr;
                bfr[j] = __builtin_bit_cast(bf16x8_t, b[r * 8 + ((ks * 4 + fq) ^ (r & 7))]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (TRANS) acc[i][j] = GYRE_MFMA_16x16x32(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    else       acc[i][j] = GYRE_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
        if (kc + 1 < nk) store_stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if (!TRANS) {
        gemm_epilogue<MI, NI>(p, acc, m0 + wm * TM, n0 + wn * TN, fr, fq);
    } else {
        // natural operand order: lane holds column n = ..+fr and 4 consecutive rows m = ..+4*fq+{0..3}
        // -> transposed store out[(b*N + n)*ldt + tok], 4 consecutive tokens = 8 bytes
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * TN + j * 16 + fr;
            if (n >= p.N) continue;
            const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm * TM + i * 16 + 4 * fq;
                if (m >= p.M) continue;
                const int b = m / p.tokens_per_batch, tok = m - b * p.tokens_per_batch;
                bf16_t* dst = (bf16_t*)p.out + ((size_t)b * p.N + n) * p.ldt + tok;
                if (m + 3 < p.M && tok + 3 < p.tokens_per_batch && ((p.ldt | tok) & 3) == 0) {
                    uint2 pk = make_uint2(pack_bf16x2(acc[i][j][0] + bn, acc[i][j][1] + bn),
                                          pack_bf16x2(acc[i][j][2] + bn, acc[i][j][3] + bn));
                    *(uint2*)dst = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int mm = m + e;
                        if (mm < p.M) {
                            int bb = mm / p.tokens_per_batch, tt = mm - bb * p.tokens_per_batch;
                            ((bf16_t*)p.out)[((size_t)bb * p.N + n) * p.ldt + tt] = f32_to_bf16(acc[i][j][e] + bn);
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(hipStream_t st, const GemmParams& p) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int grid = tiles_m * tiles_n;
    const size_t lds = (size_t)2 * (BM + BN) * 128;
    const bool trans = p.out_mode == OUT_BF16_T;
    // algorithmic work of this launch (unpadded): 2*M*N*K flops; bytes = A read once + W + C written (+ residual)
    const int kcls = (p.mode == GEMM_CONV3 ? 0 : 3) + (BM == 128 ? 0 : BM == 256 ? 1 : 2);
    const double n_out = p.geglu ? p.N / 2.0 : (double)p.N;
    const double a_bytes = p.mode == GEMM_CONV3 ? (double)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Cin * 2.0
                                                : (double)p.M * p.K * 2.0;
    GyreProfScope prof_(kcls, st, 2.0 * p.M * (double)p.N * p.K,
                        a_bytes + (double)p.N * p.K * 2.0 + (double)p.M * n_out * 2.0 * (p.residual ? 2.0 : 1.0));
#define GYRE_GEMM_GO(MODE_, UNI_, TR_)                                                                              \
    do {                                                                                                            \
        auto kern = k_gemm<BM, BN, WM, WN, MODE_, UNI_, TR_>;                                                       \
        static std::atomic<unsigned long long> attr_done{0};                                                                               \
        if (gyre_lds_attr_needed(attr_done))                                                                                            \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        hipLaunchKernelGGL(kern, dim3(grid, p.batch > 1 ? p.batch : 1), dim3(256), lds, st, p, tiles_m, tiles_n);       \
    } while (0)
    if (p.mode == GEMM_LINEAR) {
        if (trans) GYRE_GEMM_GO(GEMM_LINEAR, true, true); else GYRE_GEMM_GO(GEMM_LINEAR, true, false);
    } else {
        const bool uni = (p.Cin % BK == 0) && (p.C1 % BK == 0);
        if (trans) GYRE_FAIL(-6, "conv with transposed output is not supported");
        if (uni) GYRE_GEMM_GO(GEMM_CONV3, true, false); else GYRE_GEMM_GO(GEMM_CONV3, false, false);
    }
#undef GYRE_GEMM_GO
    GYRE_LAUNCH_CHECK();
    return 0;
}

#include <mutex>
#include <unordered_map>
// 256-byte zero page per device: source of the zero padding for the LDS-DMA kernel (read-only after creation)
const bf16_t* gemm_zero_page_for_current_device() {
    static std::mutex mu;
    static std::unordered_map<int, void*> pages;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    auto it = pages.find(dev);
    if (it != pages.end()) return (const bf16_t*)it->second;
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
    (void)hipDeviceSynchronize();
    pages[dev] = p;
    return (const bf16_t*)p;
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg8(hipStream_t st, const GemmParams& p, int kcls_base, int splits) {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int grid = tiles_m * tiles_n * splits;
    // ring depth of the linear-mode K loop (k_gemm8): as deep as the LDS of ONE workgroup per CU allows when the grid has no second
    // workgroup for a CU anyway (or the tile is too big for two), else two stages each for two co-resident workgroups
    // (tuning bit 24: the two-stage loop everywhere; bit 25: the deep ring also where two 2-stage workgroups would fit)
    const size_t stage = (size_t)(BM + (BN + 63) / 64 * 64) * 128;
    int nst = 2;
    const bool conv_ring = ((BM == 128 && BN == 160) || (BM == 256 && BN == 128)) && p.mode == GEMM_CONV3 && p.Cin % BK == 0 && p.C1 % BK == 0 && p.K % BK == 0 && !(p.debug & 0x8000000);
    if ((p.mode == GEMM_LINEAR && !(p.debug & 0x1000000)) || conv_ring) {
        const int fit1 = (int)std::min<size_t>(4, (size_t)160 * 1024 / stage);
        const bool two_fit = 4 * stage <= (size_t)160 * 1024;
        if (grid <= 256 || !two_fit || (p.debug & 0x2000000)) nst = fit1;
        const int steps = ((p.K + BK - 1) / BK + splits - 1) / splits;
        if (nst > steps) nst = steps < 2 ? 2 : steps;
        if (nst < 2) nst = 2;
    }
    size_t lds = (size_t)nst * stage;
    if (p.mode == GEMM_LINEAR && p.ln_colsum && p.geglu && gemm_geglu_lnf_lds_bytes(BM, BN, WM, WN) <= 160 * 1024)
        lds = std::max(lds, gemm_geglu_lnf_lds_bytes(BM, BN, WM, WN));
    const int lds_attr = 160 * 1024;      // (the attribute is set once per kernel and device: the most any launch of it asks for)
    const int kcls = kcls_base + (p.mode == GEMM_CONV3 ? 0 : 4);
    const double n_out = p.geglu ? p.N / 2.0 : (double)p.N;
    const double a_bytes = p.mode == GEMM_CONV3 ? (double)(p.M / (p.Ho * p.Wo)) * p.Hi * p.Wi * p.Cin * 2.0
                                                : (double)p.M * p.K * 2.0;
    GyreProfScope prof_(kcls, st, 2.0 * p.M * (double)p.N * p.K,
                        a_bytes + (double)p.N * p.K * 2.0 + (double)p.M * n_out * 2.0 * (p.residual ? 2.0 : 1.0));
#define GYRE_GEMM8_GO(MODE_, UNI_)                                                                                  \
    do {                                                                                                            \
        auto kern = k_gemm8<BM, BN, WM, WN, MODE_, UNI_>;                                                             \
        static std::atomic<unsigned long long> attr_done{0};                                                                               \
        if (gyre_lds_attr_needed(attr_done))                                                                                            \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, nst);                      \
    } while (0)
    if (p.colstat_out && splits == 1) {
        if constexpr (BN == 320 || BN == 160) {
            const bool uni = (p.Cin % BK == 0) && (p.C1 % BK == 0);
#define GYRE_GEMM8_CS(MODE_, UNI_)                                                                                  \
    do {                                                                                                            \
        auto kern = k_gemm8<BM, BN, WM, WN, MODE_, UNI_, false, false, true>;                                       \
        static std::atomic<unsigned long long> attr_done{0};                                                        \
        if (gyre_lds_attr_needed(attr_done))                                                                        \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, nst);                      \
    } while (0)
            if (p.mode == GEMM_LINEAR) GYRE_GEMM8_CS(GEMM_LINEAR, true);
            else if (uni) GYRE_GEMM8_CS(GEMM_CONV3, true);
            else GYRE_GEMM8_CS(GEMM_CONV3, false);
#undef GYRE_GEMM8_CS
        } else {
            GYRE_FAIL(-6, "gemm: column statistics exist for the 320- and 160-wide tiles only (see gemm_colstat_rows)");
        }
    } else if (p.mode == GEMM_LINEAR && p.rowstat_out) {
        auto kern = k_gemm8<BM, BN, WM, WN, GEMM_LINEAR, true, false, true>;
        static std::atomic<unsigned long long> attr_done{0};
        if (gyre_lds_attr_needed(attr_done))
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, nst);
    } else if (p.mode == GEMM_LINEAR && p.ln_colsum) {
        auto kern = k_gemm8<BM, BN, WM, WN, GEMM_LINEAR, true, true>;
        static std::atomic<unsigned long long> attr_done{0};
        if (gyre_lds_attr_needed(attr_done))
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, p, tiles_m, tiles_n, splits, nst);
    } else if (p.mode == GEMM_LINEAR) {
        GYRE_GEMM8_GO(GEMM_LINEAR, true);
    } else {
        const bool uni = (p.Cin % BK == 0) && (p.C1 % BK == 0);
        if (uni) GYRE_GEMM8_GO(GEMM_CONV3, true); else GYRE_GEMM8_GO(GEMM_CONV3, false);
    }
#undef GYRE_GEMM8_GO
    GYRE_LAUNCH_CHECK();
    prof_.stop();
    if (splits > 1) return launch_splitk_reduce(st, p, splits);
    return 0;
}

// Tile-config ids (GemmParams::force_cfg): 1 = 4w 128x128, 2 = 4w 256x64, 3 = 4w 64x64,
// 4 = 8w 256x320, 5 = 8w 128x320, 6 = 8w 256x256, 7 = 8w 128x256, 8 = 8w 128x160.
static thread_local void* g_dbg_ar_ws = nullptr;       // gyre_debug_set_ar_workspace: packed-weight scratch of bare gyre_op_* calls
static thread_local size_t g_dbg_ar_ws_bytes = 0;
// smallest grid the A-resident kernel is taken for (GYRE_AR_GRID_MIN: tuning override, read once)
static long ar_grid_min() {
    static const long v = getenv("GYRE_AR_GRID_MIN") ? atol(getenv("GYRE_AR_GRID_MIN")) : 192;
    return v;
}
static int pick_cfg(const GemmParams& p, int* splits_out) {
    *splits_out = 1;
    const bool trans = p.out_mode == OUT_BF16_T;
    auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    // wave-quantisation efficiency on 256 CUs with `slots` resident workgroups per CU
    auto eff = [&](long t, int slots) { long cap = 256L * slots; long waves = (t + cap - 1) / cap; return (double)t / (double)(waves * cap); };
    // padding efficiency along N and along M (a 256-row tile on 128 rows does half its work on padding)
    auto neff = [&](int bn) { long nt = (p.N + bn - 1) / bn; return (double)p.N / (double)(nt * bn); };
    auto meff = [&](int bm) { long mt = (p.M + bm - 1) / bm; return (double)p.M / (double)(mt * bm); };
    double best = -1; int cfg = 3;
    auto consider = [&](int id, double speed, int bm, int bn, int slots) {
        double s = speed * eff(tiles(bm, bn), slots) * neff(bn) * meff(bm);
        if (s > best) { best = s; cfg = id; }
    };
    // relative speeds measured with tools/opbench.py on MI355X (TFLOP/s at full occupancy / 1000)
    consider(3, 0.30, 64, 64, 4);
    consider(1, 0.55, 128, 128, 2);
    consider(2, 0.50, 256, 64, 2);
    // circular padding exists in the 4-wave register-staged gather only (a niche request option: correctness over speed)
    const bool wrap_only4 = p.mode == GEMM_CONV3 && p.wrap != 0;
    // A-resident kernel (kernels_gemm_ar.hip): K = 320 / 640 linear problems with enough rows to cover the chip keep their
    // activations in registers and stream only the weights (tuning bit 21: off)
    // Taken where the N sweep is long enough to amortise the slab load (GEGLU FF1: N = 8 C, 216 -> 129 us at 64x64, 163 -> 124
    // at 32x32; a plain N = 3 C: 80 -> 62 us).  The C x C projections (5 N tiles per workgroup) are HBM-bound either way and
    // measured 5 - 7 % slower here (33 -> 35 us): they stay on the 8-wave tiles unless tuning bit 22 asks for them.
    if (!trans && p.batch <= 1 && (p.ar_ok || p.w_packed || g_dbg_ar_ws) && !p.no_ar && !(p.debug & 0x200000) && p.M >= 4096 &&
        (p.N >= 3 * p.K || (p.debug & 0x400000)) && gemm_ar_supports(p) &&
        // ... and only where its grid (256-row blocks x N-range splits of at least four tiles) covers most of the chip: the fused Q|K|V
        // at batch 2 (M = 8192, 15 N tiles -> 64 workgroups) took 31 us there against 18 us on the 8-wave tile
        ((long)((p.M + 255) / 256) * gemm_ar_nsplit(p) >= ar_grid_min() || (p.debug & 0x400000)))
        return 30;
    if (!trans && p.batch <= 1 && !wrap_only4) {
        // the 1-workgroup-per-CU big tiles only pay when the grid covers most of the chip: with few tiles the
        // serial K loop of each workgroup dominates and the small tiles' extra parallelism wins
        auto big = [&](int id, double speed, int bm, int bn) { if (tiles(bm, bn) >= 160) consider(id, speed, bm, bn, 1); };
        if (!p.geglu && p.N % 320 == 0) { big(4, 0.92, 256, 320); big(5, 0.88, 128, 320); }
        // 128x160: for the problems whose 128x256 / 128x320 tiling leaves CUs idle (16x16 level: M = 4096, N = 1280 is 160 /
        // 128 tiles of those, 256 of this one).  71 FLOP per pipe byte against 85 / 91, but every CU works and two workgroups
        // fit a CU (74 KB of LDS, < 128 registers): tuning bit 18 turns it off
        // (tools/gemm_sweep.py 16 64 sd15 8, cold and producer-warm regimes: beats the 128x320 tile wherever that one is
        //  chosen - 32x32 projections 34.7 -> 28.2 us, K = 2560 84 -> 66 us - and loses to 256x320 / 256x256 where those fill
        //  the chip, M = 65536: 41 vs 38 us)
        if (!p.geglu && p.N % 160 == 0 && !(p.debug & 0x40000)) big(8, 0.90, 128, 160);
        if (p.N % 256 == 0) { big(6, 0.95, 256, 256); big(7, 0.62, 128, 256); }
        // 256x128 (config 12): 3x3 convs into 128 channels - the 512x512 level of the VAE decoder, 25 % of a decode - had no 8-wave
        // tile (N = 128 is no multiple of 160 / 256 / 320) and ran on the register-staged 128x128 tile at 0.20 - 0.22 of the MFMA peak
        if (p.mode == GEMM_CONV3 && !p.geglu && p.N % 128 == 0 && p.N % 256 != 0 && p.N % 160 != 0 && !(p.debug & 0x40000)) big(12, 0.90, 256, 128);
        // pipelined 32x32x16 kernel (kernels_gemm4s.hip), 8 waves on the 256x320 tile: better main loop (barrier off the
        // critical path, requests issued from the MFMA gaps), heavier two-pass epilogue -> long reductions only.
        // Measured in the UNet (tools/unet_layers.py, r02): 3x3 convs at 64x64 -3...-7 % time, K = 2560 linears -19 %, the
        // K = 1280 / N = 320 linear +13 % (not taken).  The 4-wave one-wave-per-SIMD forms (configs 20-23) win isolated
        // benchmarks (8192^3: 1325 vs 1048 TFLOP/s) but lose 4-15 % inside the UNet; they stay test / tuning configs.
        const int nk_ = (p.K + BK - 1) / BK;
        if (!(p.debug & 0x400) && p.N % 320 == 0 && (p.mode == GEMM_CONV3 ? nk_ >= 20 : nk_ >= 32) && gemm4s_supports(p, 24)) {
            big(24, 1.00, 256, 320);
            // 3x3 convs with K >= 2880 whose 256x320 tiling has 128 - 159 tiles (the 48x48 level of a 768 px request at batch 8:
            // M = 18432, N = 640 -> 144 tiles; no split factor fits): the pipelined tile on 56 % of the CUs still beats the 128x160
            // tile on all of them by 20 - 30 % (tools/gemm_sweep.py 8 96 sd15, round 4: K = 5760 218 -> 174 us, K = 11520 403 -> 302;
            // UNet forward at 96x96 latents, batch 8: 26.96 -> 26.38 ms).  Lowering the 128x160 tile's constant for ALL long convs
            // instead was measured too: no gain at 96x96 and an extra split-K launch at 64x64 - not adopted
            // (tuning bit 23: rule off)
            if (!(p.debug & 0x800000) && p.mode == GEMM_CONV3 && nk_ >= 45 && tiles(256, 320) >= 128 && tiles(256, 320) < 160) consider(24, 1.25, 256, 320, 1);
        }
        // GEGLU FF1 (N = 8C, K = C): the 320-wide 8-wave tile of the 16x16x32 kernel has an odd fragment count per wave and
        // cannot pair value / gate columns, the pipelined kernel's 32-wide fragments can.  Taken only where the WEIGHTS are
        // the big operand (N > M: the 16x16 level, 26 MB of weights against 10 MB of activations; 140 -> 109 us cold, UNet
        // forward 19.45 -> 19.23 ms same-box): there every operand is cold inside the UNet, as in the cache-evicting
        // comparison (COLD=1 tools/geglu_compare.py).  At 64x64 / 32x32 the activations dominate and are still in the
        // Infinity Cache behind their producer - the warm regime, where the 256x256 tile wins: with the pipelined tile on all
        // three levels the forward went 19.56 -> 19.83 ms (debug bit 14 = this rule off).
        else if (!(p.debug & 0x4400) && p.geglu && (long)p.N > (long)p.M && p.N % 320 == 0 && gemm4s_supports(p, 24))
            big(24, 1.00, 256, 320);
        // few output tiles but a long reduction (the 8x8 / 16x16 UNet levels: K = 9*Cin up to 23040): cut K into
        // slices so that tiles x slices covers the chip; fp32 slabs are reduced by k_splitk_reduce
        const int nk = (p.K + BK - 1) / BK;
        if (p.out_mode == OUT_BF16 && !p.geglu && p.K >= 2048 && p.N % 4 == 0) {
            auto tryk = [&](int id, double speed, int bm, int bn) {
                long t = tiles(bm, bn);
                if (t >= 160 || t < 1) return;
                int sp = (int)((id == 8 ? 512 : 256) / t);          // two 128x160 workgroups fit a CU
                // K steps per slice: >= 8 for the 128-row tiles (tools/gemm_sweep.py at batch 2: 16 left 10 % on the table),
                // >= 16 for the 256-row ones (8x8 level, K = 11520: 256x320 x16 slices 56 us vs 128x320 x8 slices 50 us)
                const int min_steps = bm >= 256 ? 16 : 8;
                while (sp > 1 && nk / sp < min_steps) --sp;
                if (sp < 2) return;
                // slab traffic (write + read, fp32) against the operand traffic of the GEMM itself
                double sc = speed * eff(t * sp, 1) * neff(bn) * 0.85;
                if (sc > best) { best = sc; cfg = id; *splits_out = sp; }
            };
            if (p.N % 320 == 0) tryk(5, 0.88, 128, 320);
            if (p.N % 160 == 0 && !(p.debug & 0x40000)) tryk(8, 0.92, 128, 160);     // 8x8 convs: 60 -> 57 us (sweep)
            if (p.N % 256 == 0) tryk(7, 0.62, 128, 256);
            // very long reductions (3x3 convs over 1280+ channels at 32x32 / 16x16): the 256x320 tile's better
            // operand reuse outweighs the larger slabs - measured +3 % (K = 11520) to +10 % (K = 17280 / 23040)
            // (same-box A/B: UNet forward 20.56 -> 20.12 ms)
            // (cold-cache sweep, tools/gemm_sweep.py with COLD=1 - the regime inside the UNet: from K = 5120 on, e.g. the
            //  32x32 convs 640 -> 640: 128x320 unsplit 150 us, 256x320 in 2 slices 133 us)
            if (p.N % 320 == 0 && p.K >= 5000) {
                tryk(4, 1.05, 256, 320);
                // the pipelined main loop also wins with split K (tools/gemm_sweep.py, r02: 16x16 convs K = 11520 ... 23040,
                // 4 slices: 133 -> 127, 182 -> 170, 227 -> 216 us)
                if (!(p.debug & 0x400) && gemm4s_supports(p, 24)) tryk(24, 1.10, 256, 320);
            }
        }
    }
    // small problems (the planner's answer is one of the register-staged 4-wave tiles): the 4-stage LDS-DMA ring of
    // kernels_gemm_sm.hip hides the per-K-step memory latency those kernels expose (tuning bit 5: off)
    if (!trans && !(p.debug & 0x20) && gemm_sm_supports(p)) {
        if (cfg <= 3 && *splits_out == 1) cfg = 32;
        // ... and the long-K linear problems with few rows (FF2 of the deep levels at batch 1 - 2: M = 2048, K = 2560 -> 320 tiles of
        // 64x64; M = 512, K = 5120 -> 160) that were cut into K slices: one launch instead of slices + reduction, 30.8 -> 26.4 us
        // and 32.0 -> 25.2 us (tools/sm_bench.py, FORCE=32); with more tiles the 128x160 slices win (M = 4096: 27.9 vs 32.1)
        else if (*splits_out > 1 && p.mode == GEMM_LINEAR && tiles(64, 64) <= 512 && !(p.debug & 0x40)) { cfg = 32; *splits_out = 1; }
    }
    return cfg;
}

static thread_local int g_force_cfg = 0;
static thread_local float* g_dbg_ws = nullptr;
static thread_local size_t g_dbg_ws_bytes = 0;
static thread_local int g_gemm_debug = 0;
extern "C" int gyre_debug_gemm_ablation(int bits) { int old = g_gemm_debug; g_gemm_debug = bits; return old; }
extern "C" int gyre_debug_force_gemm_cfg(int cfg) { int old = g_force_cfg; g_force_cfg = cfg; return old; }
extern "C" int gyre_debug_set_splitk_workspace(void* ws, size_t bytes) { g_dbg_ws = (float*)ws; g_dbg_ws_bytes = bytes; return 0; }
extern "C" int gyre_debug_set_ar_workspace(void* ws, size_t bytes) { g_dbg_ar_ws = ws; g_dbg_ar_ws_bytes = bytes; return 0; }

static int pick_cfg_nosplit(const GemmParams& p) {
    GemmParams q = p; q.K = q.K < 2048 ? q.K : 2040;  // same tile scoring, split path disabled
    int sp; return pick_cfg(q, &sp);
}

// Batch-invariant planning.  Every tile config sums K in the same order, so the only way the batch size can change
// a result bit is through the split-K factor (it is chosen from the tile count, i.e. from M = batch x rows).  With
// the mode on, the factor is planned for M' = rows-per-sample x canonical batch whatever the real batch is: any
// split of a request over GPUs / sub-batches then gives bit-identical images (reference property
// tests/batch_independance.py:15-27 made exact) at the price of a worse-filled chip when batch << canonical.
// Per calling thread, like the other planner knobs: the reference serves several device slots from threads of one
// process (manager.py:2107-2139), and a process-wide switch flipped by one request between another thread's workspace
// sizing and its forward would change that thread's split-K plan under its feet.
static thread_local int g_invariant_batch = 0;
int gemm_set_batch_invariant(int n) { const int old = g_invariant_batch; g_invariant_batch = n < 0 ? 0 : n; return old; }
int gemm_get_batch_invariant() { return g_invariant_batch; }
long gemm_planner_state() {
    return (long)(unsigned)g_gemm_debug | ((long)(g_force_cfg & 0xffff) << 32) | ((long)(g_invariant_batch & 0xfff) << 48) | ((long)(g_dbg_ar_ws ? 1 : 0) << 60);
}

static int plan_cfg(const GemmParams& p, int* splits) {
    const int inv = g_invariant_batch;
    if (inv > 0 && p.samples > 0 && p.M % p.samples == 0) {
        GemmParams q = p;
        q.M = p.M / p.samples * inv;
        int c = pick_cfg(q, splits);
        if (*splits > 1) return c;            // split path: 128-row tiles, fine for any M
        return pick_cfg_nosplit(p);           // no split at the canonical size -> none here either
    }
    return pick_cfg(p, splits);
}

// the condition under which k_gemm8 takes its LDS-staged epilogue (the only one that knows the folded LayerNorm)
static bool gemm_staged_epilogue_ok(const GemmParams& p) {
    const int n_out = p.geglu ? p.N / 2 : p.N;
    return p.out_mode == OUT_BF16 && (n_out % 8) == 0 && (p.ldc % 8) == 0 && (!p.residual || (p.ldr % 8) == 0) &&
           (((size_t)p.out | (size_t)p.residual) & 15) == 0;
}
bool gemm_per_sample_w_ok(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    if (g_force_cfg || p.force_cfg || (p.debug & 0x1000) || g_invariant_batch > 0) return false;   // bit 12: GroupNorm keeps its apply pass
    if (p.mode != GEMM_LINEAR || p.A2 || p.batch > 1 || p.geglu || p.vt_out || p.ln_colsum || p.rows_per_sample <= 0 ||
        p.M % p.rows_per_sample || p.K % 8 || p.N % 8)
        return false;
    if (!gemm_staged_epilogue_ok(p)) return false;
    int splits = 1;
    const int cfg = plan_cfg(p, &splits);
    if (cfg < 4 || cfg > 8 || splits > 1) return false;
    const int bm = (cfg == 4 || cfg == 6) ? 256 : 128;
    return p.rows_per_sample % bm == 0;
}
bool gemm_conv_shortcut_ok(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    if (g_force_cfg || p.force_cfg || (p.debug & 0x80)) return false;        // tuning bit 7: the shortcut stays its own launch
    // batch-invariant planning: whether the conv gets the pipelined tile depends on M = batch x rows, and the folded form rounds once
    // where the two launches round twice - so that mode keeps the two launches for every batch size (as it keeps the separate LayerNorm)
    if (g_invariant_batch > 0) return false;
    if (p.mode != GEMM_CONV3 || !p.sc_K || p.sc_K % BK || p.Cin % BK || p.K != 9 * p.Cin + p.sc_K || p.stride != 1 || p.pad != 1 || p.ups ||
        p.Hi != p.Ho || p.Wi != p.Wo || p.wrap || p.batch > 1 || p.out_mode != OUT_BF16)
        return false;
    if (!p.A2) { p.C1 = p.Cin; p.A2 = p.A; p.lda2 = p.lda; }
    if (p.C1 % BK) return false;
    if (p.rows_per_sample <= 0) p.rows_per_sample = 1;
    int splits = 1;
    const int cfg = plan_cfg(p, &splits);
    return cfg == 24 && gemm4s_supports(p, 24);
}
bool gemm_ln_fusable(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    if (g_force_cfg || p.force_cfg || (p.debug & 0x800)) return false;      // tuning runs keep the separate LayerNorm (bit 11: off)
    // batch-invariant planning: whether a shape gets an 8-wave tile depends on M = batch x rows, and the folded and the
    // separate LayerNorm round differently - so that mode keeps the separate pass for every batch size
    if (g_invariant_batch > 0) return false;
    // tuning bit 16: GEGLU FF1 keeps its separate LayerNorm (its GELU epilogue is VALU-bound and the fold adds ~15 % to it:
    // forward 19.14 ms folded, 19.26 not, 19.52 with no fold at all - same box)
    if (p.geglu && (p.debug & 0x10000)) return false;
    if (p.mode != GEMM_LINEAR || p.A2 || p.rowbias || p.batch > 1 || p.M <= 0 || p.K % 8 || p.N % 4) return false;
    if (!gemm_staged_epilogue_ok(p)) return false;
    int splits = 1;
    const int cfg = plan_cfg(p, &splits);
    if (cfg == 24) return splits == 1 && gemm4s_supports(p, 24);      // pipelined 256x320 tile (kernels_gemm4s.hip)
    if (cfg == 30) return true;                                        // A-resident kernel (kernels_gemm_ar.hip)
    // (the small-problem kernel, config 32, had the fold and the row statistics behind a tuning bit in round 4: 47 launches fewer per
    //  UNet call at batch 2 and the call 6.01 -> 6.10 ms - the LayerNorm launches it removed are cheaper than the statistics loops
    //  and folded epilogues it added; the form was deleted in round 5)
    if (cfg == 32) return false;
    if (cfg < 4 || cfg > 8 || splits > 1) return false;
    if (p.geglu && (cfg == 4 || cfg == 5 || cfg == 8)) return false;
    if (p.vt_out) {
        const int tn = cfg == 4 ? 160 : (cfg == 5 || cfg == 8) ? 80 : cfg == 6 ? 128 : 64;
        if (p.vt_col0 % tn) return false;
    }
    return true;
}

int gemm_rowstat_parts(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    if (g_force_cfg || p.force_cfg || (p.debug & 0x8800) || g_invariant_batch > 0) return 0;    // bit 15: separate statistics pass
    if (p.mode != GEMM_LINEAR || p.A2 || p.geglu || p.vt_out || p.ln_colsum || p.batch > 1 || p.M <= 0 || p.K % 8 || p.N % 8)
        return 0;
    if (!gemm_staged_epilogue_ok(p)) return 0;
    int splits = 1;
    const int cfg = plan_cfg(p, &splits);
    if (cfg == 30) return gemm_ar_nsplit(p);          // A-resident kernel: one partial per N-range split of the row block
    if (cfg == 32) return 0;                          // the small-problem kernel leaves no row statistics
    if (cfg < 4 || cfg > 8 || splits > 1) return 0;
    const int bn = (cfg == 4 || cfg == 5) ? 320 : cfg == 8 ? 160 : 256;
    return (p.N + bn - 1) / bn;
}

// Which kernel would emit the column statistics of `p`, and with what row-block size.  Mirrors launch_gemm's decisions.
int gemm_colstat_rows(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    const int unit = p.colstat_unit;
    if (unit <= 0 || g_force_cfg || p.force_cfg || (p.debug & 0x20000) || g_invariant_batch > 0) return 0;   // bit 17: separate statistics pass
    if (p.out_mode != OUT_BF16 || p.geglu || p.vt_out || p.ln_colsum || p.rowstat_out || p.batch > 1 || p.M <= 0 || p.K % 8 || p.N % 8) return 0;
    if (p.N % unit || 320 % unit || p.rows_per_sample <= 0 || p.M % p.rows_per_sample) return 0;
    if (!gemm_staged_epilogue_ok(p)) return 0;
    int splits = 1;
    const int cfg = plan_cfg(p, &splits);
    int rows = 0;
    if (splits > 1) rows = (cs_red_colblock(p.N, unit) <= 2048) ? cs_red_rows(p.rows_per_sample) : 0;     // k_splitk_reduce_cs
    else if (cfg == 4) rows = 256;
    else if (cfg == 5) rows = 128;
    else if (cfg == 8 && p.N % 160 == 0 && 160 % unit == 0 && !(p.debug & 0x100000)) rows = 128;   // (bit 20: as before this tile had the epilogue)
    else if (cfg == 24 && p.mode == GEMM_CONV3) rows = 256;                    // pipelined 256x320 tile (kernels_gemm4s.hip)
    if (!rows || p.rows_per_sample % rows) return 0;
    return rows;
}

GemmPlan gemm_plan(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;      // same planner inputs as launch_gemm
    GemmPlan pl{3, 1, 0};
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return pl;
    pl.cfg = plan_cfg(p, &pl.splits);
    if (pl.splits > 1) pl.ws_bytes = (size_t)pl.splits * p.M * p.N * sizeof(float);
    return pl;
}

// W[N][K] -> 1-KiB blocks of 8 rows x 64 k (GemmParams::W_blk); one thread per 16 bytes
__global__ __launch_bounds__(256) void k_w_block(const bf16_t* __restrict__ W, int N, int K, bf16_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int kv = K >> 3;
    if (t >= (size_t)N * kv) return;
    const int n = (int)(t / kv), k = (int)(t - (size_t)n * kv) * 8;
    const uint4 v = *(const uint4*)(W + (size_t)n * K + k);
    *(uint4*)(out + ((size_t)(n >> 3) * (K >> 6) + (k >> 6)) * 512 + (n & 7) * 64 + (k & 63)) = v;
}
int launch_w_block(hipStream_t st, const bf16_t* W, int N, int K, bf16_t* out) {
    if (K % 64 || N % 8 || N < 8 || K < 64) GYRE_FAIL(-1, "w_block: K must be a multiple of 64 and N of 8");
    const size_t total = (size_t)N * (K / 8);
    hipLaunchKernelGGL(k_w_block, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, N, K, out);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// tests / tuning: scratch where this thread's bare gyre_op_* calls block their weights on the fly
static thread_local bf16_t* g_dbg_blk_ws = nullptr;
static thread_local size_t g_dbg_blk_ws_bytes = 0;
extern "C" int gyre_debug_set_wblk_workspace(void* ws, size_t bytes) { g_dbg_blk_ws = (bf16_t*)ws; g_dbg_blk_ws_bytes = bytes; return 0; }
static bool w_block_cfg(int cfg) { return (cfg >= 4 && cfg <= 8) || cfg == 12 || (cfg >= 20 && cfg <= 24) || cfg == 32; }
bool gemm_w_block_wanted(const GemmParams& p0) {
    GemmParams p = p0;
    p.debug = g_gemm_debug;
    if (!p.force_cfg) p.force_cfg = g_force_cfg;
    if (p.debug & 0x4000000) return false;                    // tuning bit 26: row-major weights everywhere
    // a request of 8 weight rows spans 16 K bytes: beyond 16 KB it leaves the fast path of the load unit (K > 1024)
    if (p.K % 64 || p.N % 8 || p.K <= 1024 || p.batch > 1 || p.w_sample_stride || p.out_mode == OUT_BF16_T) return false;
    if (!p.A2) { p.C1 = p.mode == GEMM_LINEAR ? p.K : p.Cin; p.A2 = p.A; p.lda2 = p.lda; }
    if (p.mode == GEMM_CONV3 && (p.Cin % 64 || p.C1 % 64)) return false;      // (the conv K order walks whole 64-channel chunks)
    if (p.rows_per_sample <= 0) p.rows_per_sample = 1;
    int splits = 1;
    const int cfg = p.force_cfg ? (p.force_cfg & 0xff) : plan_cfg(p, &splits);
    return w_block_cfg(cfg);
}

// (Round 5: a weight PREFETCHER was built and removed again - per handle, the sequence of weight buffers a call's GEMM launches read
//  was recorded and replayed one <= N MB group ahead by a one-dword-per-line kernel on a second low-priority stream, ordered behind
//  events on the main stream, so that every launch would find its weights in the Infinity Cache instead of cold in HBM.  Results
//  bit-identical, and SLOWER at every group size (16 / 48 / 128 MB): UNet call 5.70 -> 6.41 - 6.54 ms at batch 2, 16.85 -> 17.36 at
//  batch 16 (profiles/r05_weight_prefetch_ab.txt): the extra 1.7 GB of requests per call queue in front of the latency-bound loads
//  of the launches they were meant to help, and an Infinity-Cache hit is not enough faster than HBM to pay that back.)
int launch_gemm(hipStream_t st, const GemmParams& p0) {
    GemmParams p = p0;
    if (!p.force_cfg) p.force_cfg = g_force_cfg;
    p.debug = g_gemm_debug;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) GYRE_FAIL(-1, "gemm: empty problem");
    if (p.K % 8) GYRE_FAIL(-1, "gemm: K must be a multiple of 8");
    if (!p.A2) { p.C1 = p.mode == GEMM_LINEAR ? p.K : p.Cin; p.A2 = p.A; p.lda2 = p.lda; }
    if (p.C1 % 8) GYRE_FAIL(-1, "gemm: source split must be a multiple of 8");
    if (p.mode == GEMM_CONV3) {
        if (p.Cin % 8 || p.K != 9 * p.Cin + p.sc_K) GYRE_FAIL(-1, "conv3x3: Cin must be a multiple of 8 and K == 9*Cin (+ the folded shortcut's channels)");
        if (p.sc_K && (p.sc_K % BK || p.Cin % BK || !p.sc_A || p.stride != 1 || p.pad != 1 || p.ups || p.Hi != p.Ho || p.Wi != p.Wo || p.wrap ||
                       (p.sc_A2 && p.sc_A2 != p.sc_A && (p.sc_C1 <= 0 || p.sc_C1 >= p.sc_K || p.sc_C1 % BK))))
            GYRE_FAIL(-1, "conv3x3: a folded shortcut needs a stride-1 / pad-1 convolution and whole 64-channel steps on both sides");
    }
    if (p.out_mode == OUT_BF16 && (p.N % 4 || p.ldc % 4 || (p.residual && p.ldr % 4)))
        GYRE_FAIL(-1, "gemm: N / ldc / ldr must be multiples of 4 for bf16 row-major output");
    if (p.geglu && (p.N % 32)) GYRE_FAIL(-1, "gemm: GEGLU needs N % 32 == 0");
    if (p.rows_per_sample <= 0) p.rows_per_sample = 1;
    if (p.out_mode == OUT_BF16_T && p.tokens_per_batch <= 0) GYRE_FAIL(-1, "gemm: tokens_per_batch required");
    if (p.batch > 1 && (p.mode != GEMM_LINEAR || p.out_mode != OUT_BF16 || p.bias || p.residual || p.rowbias || p.geglu || p.vt_out))
        GYRE_FAIL(-1, "gemm: the batched form is a plain bf16 matrix product");
    if (p.vt_out) {
        if (p.out_mode != OUT_BF16 || p.geglu || p.residual || p.tokens_per_batch <= 0 || p.tokens_per_batch % 8 || p.ldt % 8 ||
            p.vt_col0 <= 0 || p.vt_col0 >= p.N || p.N % 8)
            GYRE_FAIL(-1, "gemm: bad fused Q|K|V arguments");
    }
    if (p.rowstat_out && (p.ln_colsum || p.mode != GEMM_LINEAR || p0.A2 || p.geglu || p.vt_out || p.out_mode != OUT_BF16 || p.batch > 1))
        GYRE_FAIL(-1, "gemm: row statistics come from a plain single-source linear problem with bf16 row-major output");
    if (p.ln_colsum && (p.mode != GEMM_LINEAR || p0.A2 || !p.bias || (!p.ln_stats && p.ln_nparts <= 0) || (p.ln_nparts > 0 && !p.ln_parts) ||
                        p.rowbias || p.out_mode != OUT_BF16 || p.batch > 1))
        GYRE_FAIL(-1, "gemm: the folded LayerNorm needs a single-source linear problem with row statistics, bias and bf16 row-major output");
    if (p.sc_K && (!p.sc_A2 || p.sc_A2 == p.sc_A)) { p.sc_A2 = p.sc_A; p.sc_lda2 = p.sc_lda; p.sc_C1 = p.sc_K; }
    int splits = 1;
    int cfg = plan_cfg(p, &splits);
    {   // tuning aid (tools/gemm_sweep.py): GYRE_GEMM_DUMP=1 prints every problem shape with the planner's choice
        static const bool dump = getenv("GYRE_GEMM_DUMP") != nullptr;
        if (dump)
            fprintf(stderr, "GYRE_GEMM mode=%d M=%d N=%d K=%d Cin=%d C1=%d Hi=%d Wi=%d stride=%d ups=%d geglu=%d res=%d rowbias=%d "
                            "out=%d vt=%d batch=%d samples=%d cfg=%d splits=%d\n", p.mode, p.M, p.N, p.K, p.Cin, p.C1, p.Hi, p.Wi,
                    p.stride, p.ups, p.geglu, p.residual ? 1 : 0, p.rowbias ? 1 : 0, p.out_mode, p.vt_out ? 1 : 0, p.batch, p.samples,
                    cfg, splits);
    }
    if (p.force_cfg) { cfg = p.force_cfg & 0xff; splits = (p.force_cfg >> 8) & 0xff; if (splits < 1) splits = 1; }
    if (splits > 1) {
        if (!p.splitk_ws) { p.splitk_ws = g_dbg_ws; p.splitk_ws_bytes = g_dbg_ws_bytes; }
        if (!p.splitk_ws || p.splitk_ws_bytes < (size_t)splits * p.M * p.N * sizeof(float)) {
            splits = 1;   // no slab space: best single-split configuration instead
            cfg = pick_cfg_nosplit(p);
        }
    }
    if (p.vt_out) {
        const int tn = cfg == 4 ? 160 : (cfg == 5 || cfg == 8) ? 80 : cfg == 6 ? 128 : (cfg == 7 || cfg == 30 || cfg == 32) ? 64 : 0;
        if (!tn || splits > 1 || p.vt_col0 % tn) GYRE_FAIL(-6, "gemm: fused Q|K|V needs an 8-wave tile config whose wave tiles align with the V columns");
    }
    if (p.colstat_out) {
        GemmParams q = p0;
        q.splitk_ws = nullptr;
        const int rows = gemm_colstat_rows(q);
        const int want = splits > 1 ? cs_red_rows(p.rows_per_sample) : cfg == 4 ? 256 : (cfg == 5 || cfg == 8) ? 128 : (cfg == 24 && p.mode == GEMM_CONV3) ? 256 : -1;
        if (rows <= 0 || rows != want)
            GYRE_FAIL(-6, "gemm: column statistics are not available for this problem / tile configuration (see gemm_colstat_rows)");
    }
    if (p.w_sample_stride) {
        const int bm = (cfg == 4 || cfg == 6) ? 256 : 128;
        if (cfg < 4 || cfg > 8 || splits > 1 || p.mode != GEMM_LINEAR || p.rows_per_sample % bm || p.M % p.rows_per_sample)
            GYRE_FAIL(-6, "gemm: per-sample weights need an unsplit 8-wave tile config whose row blocks do not straddle samples (gemm_per_sample_w_ok)");
    }
    if (p.mode == GEMM_CONV3 && p.wrap && cfg > 3) GYRE_FAIL(-6, "gemm: circular padding (tiling) exists in the 4-wave tile configs only");
    if (p.sc_K && cfg != 24) GYRE_FAIL(-6, "gemm: the folded shortcut exists in the pipelined 256x320 tile only (see gemm_conv_shortcut_ok)");
    if (!p.W_blk && g_dbg_blk_ws && gemm_w_block_wanted(p0) && g_dbg_blk_ws_bytes >= (size_t)p.N * p.K * 2) {   // tests / tuning
        int rc = launch_w_block(st, p.W, p.N, p.K, g_dbg_blk_ws);
        if (rc) return rc;
        p.W_blk = g_dbg_blk_ws;
    }
    if (p.W_blk && (!w_block_cfg(cfg) || p.K % 64 || p.N % 8 || p.w_sample_stride || (p.mode == GEMM_CONV3 && (p.Cin % 64 || p.C1 % 64)))) p.W_blk = nullptr;
    if (cfg == 30) {
        if (splits > 1 || !gemm_ar_supports(p)) GYRE_FAIL(-6, "gemm: problem outside the A-resident kernel's domain (K = 320 / 640 linear, bf16 row-major output)");
        const void* wpk = p.w_packed;
        if (!wpk) {       // tests / tuning: pack into the caller's scratch buffer on the fly
            if (!g_dbg_ar_ws || g_dbg_ar_ws_bytes < gemm_ar_packed_bytes(p.N, p.K))
                GYRE_FAIL(-6, "gemm: the A-resident kernel needs the packed weight copy (GemmParams::w_packed or gyre_debug_set_ar_workspace)");
            int rc = launch_ar_pack(st, p.W, p.N, p.K, g_dbg_ar_ws);
            if (rc) return rc;
            wpk = g_dbg_ar_ws;
        }
        return launch_gemm_ar(st, p, wpk);
    }
    if (cfg == 32) {
        if (splits > 1) GYRE_FAIL(-6, "gemm: the small-problem kernel has no split-K form");
        return launch_gemm_sm(st, p);
    }
    if (p.rowstat_out && (cfg < 4 || cfg > 8 || splits > 1 || !gemm_staged_epilogue_ok(p)))
        GYRE_FAIL(-6, "gemm: row statistics need an unsplit 8-wave tile config with the staged epilogue (see gemm_rowstat_parts)");
    if (p.ln_colsum && ((cfg != 24 && (cfg < 4 || cfg > 8)) || splits > 1 || !gemm_staged_epilogue_ok(p)))
        GYRE_FAIL(-6, "gemm: the folded LayerNorm needs an unsplit 8-wave tile config with the staged epilogue (see gemm_ln_fusable)");
    if (cfg >= 4) {
        if (p.out_mode == OUT_BF16_T) GYRE_FAIL(-6, "gemm: transposed output needs a 4-wave config");
        if (p.geglu && (cfg == 4 || cfg == 5 || cfg == 8)) GYRE_FAIL(-6, "gemm: GEGLU needs an even fragment count per wave");
        p.zero_page = gemm_zero_page_for_current_device();
        if (!p.zero_page) GYRE_FAIL(-5, "gemm: cannot allocate the zero page");
    }
    switch (cfg) {
        case 1: return launch_cfg<128, 128, 2, 2>(st, p);
        case 2: return launch_cfg<256, 64, 4, 1>(st, p);
        case 3: return launch_cfg<64, 64, 2, 2>(st, p);
        case 4: return launch_cfg8<256, 320, 4, 2>(st, p, KC_G8_CONV_256x320, splits);
        case 5: return launch_cfg8<128, 320, 2, 4>(st, p, KC_G8_CONV_128x320, splits);
        case 6: return launch_cfg8<256, 256, 4, 2>(st, p, KC_G8_CONV_256x256, splits);
        case 12:
            if (p.mode != GEMM_CONV3) GYRE_FAIL(-6, "gemm: the 256x128 tile is a convolution config");
            return launch_cfg8<256, 128, 4, 2>(st, p, KC_G8_X1, splits);
        case 7: return launch_cfg8<128, 256, 2, 4>(st, p, KC_G8_CONV_128x256, splits);
        case 8: return launch_cfg8<128, 160, 4, 2>(st, p, KC_G8_CONV_128x160, splits);
        case 20: case 21: case 22: case 23: case 24: return launch_gemm4s(st, p, cfg, splits);
        default: GYRE_FAIL(-1, "gemm: unknown tile config");
    }
}
