// HBM-bound kernels: layout changes at the NCHW boundary, GroupNorm(+SiLU), LayerNorm,
// timestep embedding, the fp32 row-vector linear used on the time-embedding path, and
// weight repacking.  All loads/stores of activations are 16-byte (8 x bf16) per lane
// and coalesced along the NHWC channel axis (guide G2/G13); reductions use wave64
// shuffles + an LDS cross-wave step (guide Appendix B "Reduction").
//
// Replaces, inside the third-party modules the reference calls at
// gyre/pipeline/unet/core.py:274 and unified_pipeline.py:309,1531:
//   torch.nn.GroupNorm + SiLU, torch.nn.LayerNorm, diffusers Timesteps/TimestepEmbedding.
#include "kernels.h"
#include <cstdlib>

// ------------------------------------------------------------------------------
// NCHW (f32/bf16/f16) -> NHWC bf16, channels zero-padded to Cpad (multiple of 8)
// ------------------------------------------------------------------------------
__global__ void k_nchw_to_nhwc(const void* __restrict__ x, int dtype, int C, int HW, int Cpad, bf16_t* __restrict__ y,
                               size_t total_pix) {
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
    if (pix >= total_pix) return;
    size_t n = pix / HW, p = pix % HW;
    const size_t base = n * (size_t)C * HW + p;
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = c0 + j;
            f[j] = c < C ? load_as_f32(x, dtype, base + (size_t)c * HW) : 0.0f;
        }
        *(uint4*)(y + pix * Cpad + c0) = pack8(f);
    }
}
// NHWC bf16 (Cpad channels per pixel) -> NCHW f32 (C channels): debug taps only
__global__ void k_nhwc_to_nchw_f32(const bf16_t* __restrict__ x, int C, int HW, int Cpad, float* __restrict__ y, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over B*C*HW (output order)
    if (i >= total) return;
    size_t p = i % HW, c = (i / HW) % C, n = i / ((size_t)HW * C);
    y[i] = bf16_to_f32(x[(n * HW + p) * Cpad + c]);
}
int launch_nhwc_to_nchw_f32(hipStream_t st, const bf16_t* x, int B, int C, int HW, int Cpad, float* y) {
    size_t total = (size_t)B * C * HW;
    hipLaunchKernelGGL(k_nhwc_to_nchw_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, C, HW, Cpad, y, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// y[b][p][c] += r[b][c][p]  (ControlNet residuals arrive NCHW from host PyTorch; one bf16 rounding of the fp32 sum)
__global__ void k_add_nchw_into_nhwc(const void* __restrict__ r, int dtype, int C, int HW, int Cpad, bf16_t* __restrict__ y,
                                     size_t total_pix) {
    size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total_pix) return;
    size_t n = pix / HW, p = pix % HW;
    const size_t base = n * (size_t)C * HW + p;
    for (int c0 = 0; c0 < C; c0 += 8) {
        float f[8];
        unpack8(*(const uint4*)(y + pix * Cpad + c0), f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < C) f[j] += load_as_f32(r, dtype, base + (size_t)(c0 + j) * HW);
        *(uint4*)(y + pix * Cpad + c0) = pack8(f);
    }
}
int launch_add_nchw_into_nhwc(hipStream_t st, const void* r, int dtype, int B, int C, int HW, int Cpad, bf16_t* y) {
    size_t total = (size_t)B * HW;
    hipLaunchKernelGGL(k_add_nchw_into_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r, dtype, C, HW, Cpad, y, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
int launch_nchw_to_nhwc(hipStream_t st, const void* x, int dtype, int B, int C, int HW, int Cpad, bf16_t* y) {
    size_t total = (size_t)B * HW;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, dtype, C, HW, Cpad, y,
                       total);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// dst[o][0 : inner] = dst[o][inner : 2 inner] = src[o][0 : inner] for o < outer (inner in 16-byte vectors): a half batch
// written twice - the two halves of a CFG-parallel call share every activation up to the first cross-attention
__global__ __launch_bounds__(256) void k_dup_batch(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t inner, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t o = i / inner, r = i - o * inner;
    const uint4 v = src[i];
    dst[o * 2 * inner + r] = v;
    dst[o * 2 * inner + inner + r] = v;
}
int launch_dup_batch(hipStream_t st, const void* src, void* dst, size_t outer, size_t inner_bytes) {
    if (inner_bytes % 16 || ((size_t)src | (size_t)dst) % 16) GYRE_FAIL(-1, "dup_batch: 16-byte granularity");
    const size_t inner = inner_bytes / 16, total = outer * inner;
    if (!total) return 0;
    hipLaunchKernelGGL(k_dup_batch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, inner, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}

__global__ void k_cast_to_bf16(const void* __restrict__ x, int dtype, size_t n, bf16_t* __restrict__ y) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = f32_to_bf16(load_as_f32(x, dtype, i));
}
int launch_ctx_to_bf16(hipStream_t st, const void* x, int dtype, size_t n, bf16_t* y) {
    hipLaunchKernelGGL(k_cast_to_bf16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, dtype, n, y);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// sinusoidal timestep embedding, fp32.  [cos | sin] when flip (SD: flip_sin_to_cos=True,
// freq_shift=0); exponent = -ln(10000) * i / (half - shift).
// ------------------------------------------------------------------------------
__global__ void k_timestep_embedding(const int64_t* __restrict__ t, int dim, int flip, float shift,
                                     float* __restrict__ out) {
    int b = blockIdx.x;
    int half = dim / 2;
    float tv = (float)t[b];
    for (int j = threadIdx.x; j < dim; j += blockDim.x) {
        int i = j < half ? j : j - half;
        float freq = expf(-9.210340371976184f * (float)i / ((float)half - shift));
        float a = tv * freq;
        bool is_cos = flip ? (j < half) : (j >= half);
        out[(size_t)b * dim + j] = is_cos ? cosf(a) : sinf(a);
    }
}
int launch_timestep_embedding(hipStream_t st, const int64_t* t, int B, int dim, int flip, float shift, float* out) {
    hipLaunchKernelGGL(k_timestep_embedding, dim3(B), dim3(256), 0, st, t, dim, flip, shift, out);
    GYRE_LAUNCH_CHECK();
    return 0;
}

__global__ void k_add_f32(float* __restrict__ y, const float* __restrict__ x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
__global__ __launch_bounds__(256) void k_concat_rows(const bf16_t* __restrict__ a, int K1, const bf16_t* __restrict__ b, int K2, int N,
                                                     bf16_t* __restrict__ out) {
    const int kv = (K1 + K2) >> 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)N * kv) return;
    const int n = (int)(t / kv), k = (int)(t - (size_t)n * kv) * 8;
    const uint4 v = k < K1 ? *(const uint4*)(a + (size_t)n * K1 + k) : *(const uint4*)(b + (size_t)n * K2 + (k - K1));
    *(uint4*)(out + (size_t)n * (K1 + K2) + k) = v;
}
int launch_concat_rows(hipStream_t st, const bf16_t* a, int K1, const bf16_t* b, int K2, int N, bf16_t* out) {
    if (K1 % 8 || K2 % 8 || N < 1) GYRE_FAIL(-1, "concat_rows: row lengths must be multiples of 8");
    const size_t total = (size_t)N * ((K1 + K2) / 8);
    hipLaunchKernelGGL(k_concat_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, K1, b, K2, N, out);
    GYRE_LAUNCH_CHECK();
    return 0;
}
int launch_add_f32(hipStream_t st, float* y, const float* x, size_t n) {
    hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, n);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// out[b][n] = sum_k f(x[b][k]) * W[n][k] + bias[n]   (f = SiLU when act_in_silu: applied to x IN PLACE by a tiny
// elementwise launch first - inside the dot products it was evaluated once per output column, 367 M exponentials for
// the batched time_emb_proj of a CFG batch of 16, which made this 46 MB weight stream take 200 us)
// One wave per output column n, rows in groups of 16.  Weight-read bound (B <= 32).
// ------------------------------------------------------------------------------
#define RV_ROWS 16
__global__ __launch_bounds__(256) void k_silu_inplace_f32(float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = silu_f(x[i]);
}
// One wave per RV_NPW consecutive output columns: the activation rows (fp32, L2-resident) are fetched once per k chunk and
// reused for all of them - with one column per wave the x re-reads (B x K x 4 bytes per column, 13x the weight bytes at
// B = 16) bounded the all-resnets time_emb_proj launch (20160 columns) at ~86 us against ~10 us of weight streaming.
#define RV_NPW 4
__global__ __launch_bounds__(256) void k_rowvec_linear(const float* __restrict__ x, int B, int K,
                                                       const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                       int N, float* __restrict__ out, int ldo) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + wave) * RV_NPW;
    if (n0 >= N) return;
    for (int b0 = 0; b0 < B; b0 += RV_ROWS) {
        float acc[RV_NPW][RV_ROWS];
#pragma unroll
        for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
            for (int r = 0; r < RV_ROWS; ++r) acc[c][r] = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            float w[RV_NPW][8];
#pragma unroll
            for (int c = 0; c < RV_NPW; ++c) {
                if (n0 + c < N) unpack8(*(const uint4*)(W + (size_t)(n0 + c) * K + k), w[c]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[c][j] = 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < RV_ROWS; ++r) {
                if (b0 + r < B) {
                    const float* xr = x + (size_t)(b0 + r) * K + k;
                    float4 x0 = *(const float4*)xr, x1 = *(const float4*)(xr + 4);
                    float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                    for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][r] = fmaf(xv[j], w[c][j], acc[c][r]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
            for (int r = 0; r < RV_ROWS; ++r) {
                float v = acc[c][r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0 && b0 + r < B && n0 + c < N) out[(size_t)(b0 + r) * ldo + n0 + c] = v + (bias ? bias[n0 + c] : 0.f);
            }
    }
}
// Round 5, one-image latency: the same dot products for B <= 4 rows (the uniform-timestep call has ONE).  The general kernel
// above carries 16 row accumulators per column whatever B is - 64 x 6 cross-lane adds per wave for a single row - and the
// time-embedding chain in front of every UNet call was six dependent launches (embedding, Linear, SiLU, Linear, SiLU, the
// all-resnets Linear: 91 us of a 5.6 ms batch-2 call, rocprofv3 timeline).  Here the input transform runs on load (INMODE 1:
// SiLU; INMODE 2: the sinusoidal embedding of t[b] itself, same expressions as k_timestep_embedding), every weight request of
// the lane (up to 4 k chunks x RV_NPW columns) is in flight before the first multiply, and only ROWS accumulators per column
// are reduced.  Per column the products are added in the order of the general kernel (k chunks ascending, 8 elements each,
// then the xor butterfly 32 .. 1): bit-identical outputs (tests/test_gpu_models.py::test_uniform_timestep_fast_path_is_bit_identical
// compares the B-row tensor form on the general kernel with this one).
#define RVS_TRIPS 4
template <int ROWS, int INMODE>
__global__ __launch_bounds__(256) void k_rowvec_small(const float* __restrict__ x, const int64_t* __restrict__ t, int B, int K,
                                                      const bf16_t* __restrict__ W, const float* __restrict__ bias, int N,
                                                      float* __restrict__ out, int ldo, int flip, float shift) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + wave) * RV_NPW;
    if (n0 >= N) return;
    float acc[RV_NPW][ROWS];
#pragma unroll
    for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[c][r] = 0.f;
    const int half = K / 2;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8 * RVS_TRIPS) {
        uint4 wv[RVS_TRIPS][RV_NPW];
#pragma unroll
        for (int i = 0; i < RVS_TRIPS; ++i) {
            const int k = k0 + i * 512;
#pragma unroll
            for (int c = 0; c < RV_NPW; ++c)
                if (k < K && n0 + c < N) wv[i][c] = *(const uint4*)(W + (size_t)(n0 + c) * K + k);
        }
#pragma unroll
        for (int i = 0; i < RVS_TRIPS; ++i) {
            const int k = k0 + i * 512;
            if (k < K) {
                float w[RV_NPW][8];
#pragma unroll
                for (int c = 0; c < RV_NPW; ++c) {
                    if (n0 + c < N) unpack8(wv[i][c], w[c]);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[c][j] = 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    if (r < B) {
                        float xv[8];
                        if (INMODE == 2) {
                            const float tv = (float)t[r];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int jj = k + j;
                                const int ii = jj < half ? jj : jj - half;
                                const float freq = expf(-9.210340371976184f * (float)ii / ((float)half - shift));
                                const float a = tv * freq;
                                const bool is_cos = flip ? (jj < half) : (jj >= half);
                                xv[j] = is_cos ? cosf(a) : sinf(a);
                            }
                        } else {
                            const float* xr = x + (size_t)r * K + k;
                            const float4 x0 = *(const float4*)xr, x1 = *(const float4*)(xr + 4);
                            xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
                            if (INMODE == 1) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) xv[j] = silu_f(xv[j]);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[c][r] = fmaf(xv[j], w[c][j], acc[c][r]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < RV_NPW; ++c)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            float v = acc[c][r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && r < B && n0 + c < N) out[(size_t)r * ldo + n0 + c] = v + (bias ? bias[n0 + c] : 0.f);
        }
}
template <int INMODE>
static int launch_rowvec_small(hipStream_t st, const float* x, const int64_t* t, int B, int K, const bf16_t* W, const float* bias,
                               int N, float* out, int ldo, int flip, float shift) {
    const dim3 grid((N + 4 * RV_NPW - 1) / (4 * RV_NPW));
    if (B == 1) hipLaunchKernelGGL((k_rowvec_small<1, INMODE>), grid, dim3(256), 0, st, x, t, B, K, W, bias, N, out, ldo, flip, shift);
    else if (B == 2) hipLaunchKernelGGL((k_rowvec_small<2, INMODE>), grid, dim3(256), 0, st, x, t, B, K, W, bias, N, out, ldo, flip, shift);
    else hipLaunchKernelGGL((k_rowvec_small<4, INMODE>), grid, dim3(256), 0, st, x, t, B, K, W, bias, N, out, ldo, flip, shift);
    GYRE_LAUNCH_CHECK();
    return 0;
}
int launch_rowvec_linear(hipStream_t st, float* x, int B, int K, const bf16_t* W, const float* bias, int N,
                         int act_in_silu, float* out, int ldo) {
    if (K % 8) GYRE_FAIL(-1, "rowvec_linear: K must be a multiple of 8");
    if (B <= 4)                                     // SiLU on load, x left as it is
        return act_in_silu ? launch_rowvec_small<1>(st, x, nullptr, B, K, W, bias, N, out, ldo, 0, 0.f)
                           : launch_rowvec_small<0>(st, x, nullptr, B, K, W, bias, N, out, ldo, 0, 0.f);
    if (act_in_silu) {
        const size_t n = (size_t)B * K;
        hipLaunchKernelGGL(k_silu_inplace_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
        GYRE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rowvec_linear, dim3((N + 4 * RV_NPW - 1) / (4 * RV_NPW)), dim3(256), 0, st, x, B, K, W, bias, N, out, ldo);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// out[b] = Linear(timestep_embedding(t[b])): the first two launches of the time-embedding chain in one for B <= 4, the two
// launches otherwise (emb_scratch: [B][dim] floats for that case)
int launch_timestep_linear(hipStream_t st, const int64_t* t, int B, int dim, int flip, float shift, float* emb_scratch,
                           const bf16_t* W, const float* bias, int N, float* out, int ldo) {
    if (B <= 4 && dim % 8 == 0) return launch_rowvec_small<2>(st, nullptr, t, B, dim, W, bias, N, out, ldo, flip, shift);
    if (int rc = launch_timestep_embedding(st, t, B, dim, flip, shift, emb_scratch)) return rc;
    return launch_rowvec_linear(st, emb_scratch, B, dim, W, bias, N, 0, out, ldo);
}

// ------------------------------------------------------------------------------
// GroupNorm.  Stage 1: per (sample, pixel-chunk) block accumulates per-channel sum / sumsq over
// its pixels with fully coalesced row reads, folds channels into groups through LDS and writes
// one (sum, sumsq) pair per group - deterministic (no float atomics; bit-identical for any batch
// composition, the reference's batch-independence property tests/batch_independance.py:15-27).
// Stage 2: per sample, fixed-order reduction over chunks -> mean/rstd -> per-channel a,b.
// Stage 3 (apply): y = act(a*x+b).
// ------------------------------------------------------------------------------
#define GN_MAXV 4  // channel vectors per thread: supports C <= 8*256*4
// 16-byte non-temporal load (streamed-once data: +11 % from HBM in tools/ubench/l2_bw.hip)
typedef unsigned gn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load16(const void* p) {
    const gn_u32x4 v = __builtin_nontemporal_load((const gn_u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__global__ __launch_bounds__(256) void k_gn_partial(GnParams p, int TX, int PY, int pix_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;  // [PY][C][2]
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int CV = p.C / 8;
    const int C2 = p.C - p.C1;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(p.HW, p0 + pix_per_chunk);
    float s[GN_MAXV][8], ss[GN_MAXV][8];
#pragma unroll
    for (int v = 0; v < GN_MAXV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[v][j] = 0.f; ss[v][j] = 0.f; }
    if (ty < PY) {
        if (CV <= TX) {
            // one channel vector per lane (C <= 2048, every UNet / VAE GroupNorm): four pixel rows requested per trip, added in pixel
            // order - the one-row loop waited for every load before issuing the next (round 5, same sums)
            if (tx < CV) {
                const int c = tx * 8;
                const bool first = c < p.C1;
                const bf16_t* base = first ? p.x + c : p.x2 + (c - p.C1);
                const size_t ld = first ? (size_t)p.C1 : (size_t)C2;
                for (int pix = p0 + ty; pix < p1; pix += 4 * PY) {
                    uint4 raw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (pix + u * PY < p1) raw[u] = *(const uint4*)(base + ((size_t)n * p.HW + pix + u * PY) * ld);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (pix + u * PY < p1) {
                            float f[8];
                            unpack8(raw[u], f);
#pragma unroll
                            for (int j = 0; j < 8; ++j) { s[0][j] += f[j]; ss[0][j] = fmaf(f[j], f[j], ss[0][j]); }
                        }
                }
            }
        } else
        for (int pix = p0 + ty; pix < p1; pix += PY) {
            size_t gp = (size_t)n * p.HW + pix;
#pragma unroll
            for (int v = 0; v < GN_MAXV; ++v) {
                int cv = tx + v * TX;
                if (cv < CV) {
                    int c = cv * 8;
                    const bf16_t* src = c < p.C1 ? p.x + gp * p.C1 + c : p.x2 + gp * C2 + (c - p.C1);
                    float f[8];
                    unpack8(*(const uint4*)src, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s[v][j] += f[j]; ss[v][j] = fmaf(f[j], f[j], ss[v][j]); }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < GN_MAXV; ++v) {
            int cv = tx + v * TX;
            if (cv < CV) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    red[((size_t)ty * p.C + cv * 8 + j) * 2 + 0] = s[v][j];
                    red[((size_t)ty * p.C + cv * 8 + j) * 2 + 1] = ss[v][j];
                }
            }
        }
    }
    __syncthreads();
    // fold in two fixed-order steps that use the whole block: per channel over the PY pixel rows, then per group over its
    // channels.  (One thread per group walking PY x C/G entries left 7/8 of the block idle for ~120 dependent LDS reads while
    // the chip - every workgroup of the launch is resident at once - streamed nothing.)
    float2* red2 = (float2*)red;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int y = 0; y < PY; ++y) { const float2 t = red2[(size_t)y * p.C + c]; a += t.x; b += t.y; }
        red2[c] = make_float2(a, b);          // row 0, own column: nobody else reads it before the barrier
    }
    __syncthreads();
    const int cpg = p.C / p.G;
    for (int g = threadIdx.x; g < p.G; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { const float2 t = red2[c]; a += t.x; b += t.y; }
        float* dst = p.partial + (((size_t)n * p.nchunks + chunk) * p.G + g) * 2;
        dst[0] = a; dst[1] = b;
    }
}

__global__ __launch_bounds__(256) void k_gn_finalize(GnParams p) {
    // 256 threads per sample: thread (g, part) sums every PARTS-th chunk of group g, then a fixed-order
    // LDS tree over the parts (deterministic; latency ~ nchunks/PARTS dependent loads instead of nchunks)
    __shared__ float red_s[256], red_q[256];
    __shared__ float mean_s[256], rstd_s[256];
    const int n = blockIdx.x;
    const int cpg = p.C / p.G;
    const float cnt = (float)p.HW * (float)cpg;
    int parts = 256 / p.G;
    if (parts < 1) parts = 1;
    const int g = threadIdx.x % p.G, part = threadIdx.x / p.G;
    float a = 0.f, b = 0.f;
    if (part < parts)
        for (int ch = part; ch < p.nchunks; ch += parts) {
            const float* src = p.partial + (((size_t)n * p.nchunks + ch) * p.G + g) * 2;
            a += src[0]; b += src[1];
        }
    red_s[threadIdx.x] = a; red_q[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < p.G) {
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < parts; ++q) { sa += red_s[q * p.G + g]; sb += red_q[q * p.G + g]; }
        float mean = sa / cnt;
        float var = fmaxf(sb / cnt - mean * mean, 0.f);
        mean_s[g] = mean;
        rstd_s[g] = rsqrtf(var + p.eps);
        if (p.mean_rstd) {
            p.mean_rstd[((size_t)n * p.G + g) * 2 + 0] = mean;
            p.mean_rstd[((size_t)n * p.G + g) * 2 + 1] = rstd_s[g];
        }
    }
    __syncthreads();
    float* sc = p.scale_shift + (size_t)n * 2 * p.C;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        int gg = c / cpg;
        float aa = rstd_s[gg] * p.gamma[c];
        sc[c] = aa;
        sc[p.C + c] = p.beta[c] - mean_s[gg] * aa;
    }
}

__global__ __launch_bounds__(256) void k_gn_apply(GnParams p, size_t total_vec) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_vec) return;
    const int CV = p.C / 8;
    size_t gp = idx / CV;             // global pixel index over B*HW
    int c = (int)(idx % CV) * 8;
    int n = (int)(gp / p.HW);
    const int C2 = p.C - p.C1;
    const bf16_t* src = c < p.C1 ? p.x + gp * p.C1 + c : p.x2 + gp * C2 + (c - p.C1);
    float f[8];
    unpack8(*(const uint4*)src, f);
    const float* sc = p.scale_shift + (size_t)n * 2 * p.C + c;
    float4 a0 = *(const float4*)sc, a1 = *(const float4*)(sc + 4);
    float4 b0 = *(const float4*)(sc + p.C), b1 = *(const float4*)(sc + p.C + 4);
    float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = fmaf(a[j], f[j], b[j]);
        f[j] = p.silu ? silu_f(v) : v;
    }
    *(uint4*)(p.y + gp * p.C + c) = pack8(f);
}

// Apply with the finalize step folded into its prologue: a workgroup = (pixel chunk, sample) like k_gn_partial; it first redoes
// k_gn_finalize's arithmetic for its sample (same order -> same bits: nchunks partial sums per group from L2, then a = rstd * gamma,
// b = beta - mean * a per channel into LDS) and then streams its pixels.  One launch and one dependent round trip per GroupNorm
// less; the extra reads are 2 KB per thread block against 40 KB of pixels.
// GN_U = rows a lane requests per trip: 12 (all of a lane's rows at the UNet's map sizes: one memory round trip per launch; VAE-sized
// tensors in many waves of workgroups: decode of 8 images 31.85 -> 31.0 ms), except where ONE resident wave of workgroups moves tens
// of MB - then every workgroup reads at the same time and writes at the same time, and trips of 4 (reads of the next trip under the
// writes of this one) are faster: 64x64x320 at batch 16, 84 MB, 19.6 us against 22.8.
template <int GN_U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_gn_apply_fin(GnParams p, int TX, int PY, int pix_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* ab = (float*)smem_raw;                 // [2][C]
    float* red_s = ab + 2 * p.C;                  // [256]
    float* red_q = red_s + 256;                   // [256]
    float* mean_s = red_q + 256;                  // [256]
    float* rstd_s = mean_s + 256;                 // [256]
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int cpg = p.C / p.G;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int CV = p.C / 8, C2 = p.C - p.C1;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(p.HW, p0 + pix_per_chunk);
    // Round 5: every global request of the kernel that does not depend on the statistics is ISSUED BEFORE the statistics prologue -
    // the lane's pixels (up to GN_U 16-byte rows: all of them at the UNet's map sizes) and gamma / beta - so the prologue's dependent
    // chain (partials -> mean / rstd -> a, b: three barriers) runs under the one memory round trip the kernel cannot avoid instead
    // of in front of it (before: partials, then gamma / beta, then the pixels four at a time = 3 + ceil(pixels / 4) round trips
    // per launch, 15 - 18 us for maps a 6 TB/s stream would move in 4 - 7).  Loads are non-temporal (x is read exactly once
    // here), the stores are not (the next conv reads y).  Same arithmetic, same order: bit-identical.
    constexpr int GN_GB = 4;
    const bool fast = CV <= TX;                      // one channel vector per lane (C <= 2048): the common case
    const bool lane_on = ty < PY && tx < CV;
    const int c_lane = tx * 8;
    const bool first_src = c_lane < p.C1;
    const bf16_t* base = first_src ? p.x + c_lane : p.x2 + (c_lane - p.C1);
    const size_t ld = first_src ? (size_t)p.C1 : (size_t)C2;
    uint4 raw[GN_U];
    if (fast && lane_on) {
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int px = p0 + ty + u * PY;
            if (px < p1) raw[u] = nt_load16(base + ((size_t)n * p.HW + px) * ld);
        }
    }
    float gm[GN_GB], bt[GN_GB];                      // gamma / beta of channels threadIdx.x + k * 256, k < GN_GB (C <= 1024 in registers)
#pragma unroll
    for (int k = 0; k < GN_GB; ++k) {
        const int c = threadIdx.x + k * 256;
        if (c < p.C) { gm[k] = p.gamma[c]; bt[k] = p.beta[c]; }
    }
    {
        const float cnt = (float)p.HW * (float)cpg;
        int parts = 256 / p.G;
        if (parts < 1) parts = 1;
        const int g = threadIdx.x % p.G, part = threadIdx.x / p.G;
        float a = 0.f, b = 0.f;
        // partial sums are fetched FOUR at a time and added in the original order (one load, one wait, one add per partial made
        // the prologue a chain of 2 - 8 dependent L2 round trips)
        auto gather = [&](const float2* src, int chunks, int row, int u0, int u1) {     // rows part, part + parts, ...; columns [u0, u1)
            const int nu = u1 - u0;
            if (nu <= 0 || part >= chunks) return;
            const int total = ((chunks - part + parts - 1) / parts) * nu;
            for (int i0 = 0; i0 < total; i0 += 4) {
                float2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j;
                    if (i < total) t[j] = src[(size_t)(part + (i / nu) * parts) * row + u0 + i % nu];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < total) { a += t[j].x; b += t[j].y; }
            }
        };
        if (p.cs_x) {
            // statistics left by the PRODUCERS of x / x2 (GemmParams::colstat_out): per source [B][chunks][C_src / unit][2].
            // Group g = channels [g * cpg, (g + 1) * cpg) of the concatenation: its units below C1 come from x, the others
            // from x2 (unit divides C1, C - C1 and cpg).  Fixed order: source, chunk (strided over the parts), unit.
            const int unit = p.cs_unit, c_lo = g * cpg, c_hi = c_lo + cpg;
            if (part < parts) {
                const int nu1 = p.C1 / unit;
                gather((const float2*)p.cs_x + (size_t)n * p.cs_x_chunks * nu1, p.cs_x_chunks, nu1,
                       min(c_lo, p.C1) / unit, min(c_hi, p.C1) / unit);
                const int nu2 = (p.C - p.C1) / unit;
                if (nu2 > 0)
                    gather((const float2*)p.cs_x2 + (size_t)n * p.cs_x2_chunks * nu2, p.cs_x2_chunks, nu2,
                           (max(c_lo, p.C1) - p.C1) / unit, (max(c_hi, p.C1) - p.C1) / unit);
            }
        } else if (part < parts)
            gather((const float2*)p.partial + (size_t)n * p.nchunks * p.G, p.nchunks, p.G, g, g + 1);
        red_s[threadIdx.x] = a; red_q[threadIdx.x] = b;
        __syncthreads();
        if (threadIdx.x < p.G) {
            float sa = 0.f, sb = 0.f;
            for (int q = 0; q < parts; ++q) { sa += red_s[q * p.G + g]; sb += red_q[q * p.G + g]; }
            const float mean = sa / cnt;
            const float var = fmaxf(sb / cnt - mean * mean, 0.f);
            mean_s[g] = mean;
            rstd_s[g] = rsqrtf(var + p.eps);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GN_GB; ++k) {
            const int c = threadIdx.x + k * 256;
            if (c < p.C) {
                const int gg = c / cpg;
                const float aa = rstd_s[gg] * gm[k];
                ab[c] = aa;
                ab[p.C + c] = bt[k] - mean_s[gg] * aa;
            }
        }
        for (int c = threadIdx.x + GN_GB * 256; c < p.C; c += blockDim.x) {
            const int gg = c / cpg;
            const float aa = rstd_s[gg] * p.gamma[c];
            ab[c] = aa;
            ab[p.C + c] = p.beta[c] - mean_s[gg] * aa;
        }
        __syncthreads();
    }
    if (ty >= PY) return;
    auto one = [&](const uint4& rw, size_t gp, int c) {
        float f[8];
        unpack8(rw, f);
        const float4 a0 = *(const float4*)(ab + c), a1 = *(const float4*)(ab + c + 4);
        const float4 b0 = *(const float4*)(ab + p.C + c), b1 = *(const float4*)(ab + p.C + c + 4);
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = fmaf(a[j], f[j], b[j]);
            f[j] = p.silu ? silu_f(y) : y;
        }
        *(uint4*)(p.y + gp * p.C + c) = pack8(f);
    };
    if (fast) {
        if (tx >= CV) return;
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {             // the trip requested above
            const int px = p0 + ty + u * PY;
            if (px < p1) one(raw[u], (size_t)n * p.HW + px, c_lane);
        }
        for (int pix = p0 + ty + GN_U * PY; pix < p1; pix += GN_U * PY) {      // VAE-sized chunks: further trips
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int px = pix + u * PY;
                if (px < p1) raw[u] = nt_load16(base + ((size_t)n * p.HW + px) * ld);
            }
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int px = pix + u * PY;
                if (px < p1) one(raw[u], (size_t)n * p.HW + px, c_lane);
            }
        }
        return;
    }
    for (int pix = p0 + ty; pix < p1; pix += PY) {
        const size_t gp = (size_t)n * p.HW + pix;
        uint4 rawv[GN_MAXV];
#pragma unroll
        for (int v = 0; v < GN_MAXV; ++v) {
            const int cv = tx + v * TX;
            if (cv < CV) {
                const int c = cv * 8;
                rawv[v] = nt_load16(c < p.C1 ? p.x + gp * p.C1 + c : p.x2 + gp * C2 + (c - p.C1));
            }
        }
#pragma unroll
        for (int v = 0; v < GN_MAXV; ++v) {
            const int cv = tx + v * TX;
            if (cv < CV) one(rawv[v], gp, cv * 8);
        }
    }
}
// the fused form needs nothing k_gn_finalize would have to publish (the backward pass asks for mean / rstd)
static bool gn_finalize_in_apply(const GnParams& p) {
    static const bool separate = getenv("GYRE_GN_SEPARATE_FINALIZE") != nullptr;      // tuning / bit-equality checks
    return !separate && !p.mean_rstd && p.G <= 256;
}

int gn_pick_chunks(int B, int HW, int C) {
    // The chunking fixes the order the per-sample statistics are summed in, so it must not depend on B (batch
    // independence): 64 chunks per sample for 1024 <= HW <= 16384 (what the batch-16 tuning used), 16-pixel chunks
    // below, 256-pixel chunks above (VAE resolutions).
    (void)B; (void)C;
    int ppc = HW / 64;
    ppc = ppc < 16 ? 16 : (ppc > 256 ? 256 : ppc);
    int n = (HW + ppc - 1) / ppc;
    return n < 1 ? 1 : n;
}
size_t gn_workspace_bytes(int B, int HW, int C, int G) {
    int nch = gn_pick_chunks(B, HW, C);
    size_t partial = (size_t)B * nch * G * 2 * sizeof(float);
    size_t ss = (size_t)B * 2 * C * sizeof(float);
    return ((partial + 255) & ~(size_t)255) + ((ss + 255) & ~(size_t)255);
}
int launch_groupnorm_stats(hipStream_t st, const GnParams& p) {
    if (p.C % 8 || p.C1 % 8 || p.C % p.G) GYRE_FAIL(-1, "groupnorm: C, C1 must be multiples of 8 and C of groups");
    if (p.G > 256) GYRE_FAIL(-1, "groupnorm: at most 256 groups");
    int CV = p.C / 8;
    int TX = CV < 256 ? CV : 256;
    if ((CV + TX - 1) / TX > GN_MAXV) GYRE_FAIL(-6, "groupnorm: C too large");
    int PY = 256 / TX; if (PY < 1) PY = 1;
    int ppc = (p.HW + p.nchunks - 1) / p.nchunks;
    size_t lds = (size_t)PY * p.C * 2 * sizeof(float);
    if (lds > 160 * 1024) GYRE_FAIL(-6, "groupnorm: LDS budget exceeded");
    GyreProfScope prof_(KC_GN_STATS, st, 0.0, (double)p.B * p.HW * p.C * 2.0);
    hipLaunchKernelGGL(k_gn_partial, dim3(p.nchunks, p.B), dim3(256), lds, st, p, TX, PY, ppc);
    GYRE_LAUNCH_CHECK();
    if (gn_finalize_in_apply(p)) return 0;           // launch_groupnorm_apply finishes the statistics itself
    hipLaunchKernelGGL(k_gn_finalize, dim3(p.B), dim3(256), 0, st, p);
    GYRE_LAUNCH_CHECK();
    return 0;
}
bool gn_accepts_colstats(const GnParams& p) {
    const int unit = p.cs_unit;
    if (unit <= 0 || !gn_finalize_in_apply(p) || gn_use_small(p.HW, p.C, p.C1, p.G)) return false;
    const int cpg = p.C / p.G;
    return p.C % p.G == 0 && cpg % unit == 0 && p.C1 % unit == 0 && (p.C - p.C1) % unit == 0;
}
int launch_groupnorm_apply(hipStream_t st, const GnParams& p) {
    if (p.cs_x && (!gn_accepts_colstats(p) || p.cs_x_chunks <= 0 || (p.C1 < p.C && (!p.cs_x2 || p.cs_x2_chunks <= 0))))
        GYRE_FAIL(-1, "groupnorm: producer statistics need the finalize-in-apply form, units that divide the groups and both sources");
    size_t total = (size_t)p.B * p.HW * (p.C / 8);
    GyreProfScope prof_(KC_GN_APPLY, st, 0.0, (double)p.B * p.HW * p.C * 4.0);
    if (gn_finalize_in_apply(p)) {
        const int CV = p.C / 8;
        const int TX = CV < 256 ? CV : 256;
        if ((CV + TX - 1) / TX > GN_MAXV) GYRE_FAIL(-6, "groupnorm: C too large");
        int PY = 256 / TX; if (PY < 1) PY = 1;
        const int ppc = (p.HW + p.nchunks - 1) / p.nchunks;
        const size_t lds = ((size_t)2 * p.C + 4 * 256) * sizeof(float);
        // one resident wave of workgroups (<= 4 per CU) over tens of MB: every workgroup is in the same phase at the same time
        if ((size_t)p.nchunks * p.B <= 1024 && (size_t)p.B * p.HW * p.C * 2 >= ((size_t)32 << 20))
            hipLaunchKernelGGL(k_gn_apply_fin<4>, dim3(p.nchunks, p.B), dim3(256), lds, st, p, TX, PY, ppc);
        else
            hipLaunchKernelGGL(k_gn_apply_fin<12>, dim3(p.nchunks, p.B), dim3(256), lds, st, p, TX, PY, ppc);
        GYRE_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_gn_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// GroupNorm folded into the 1x1 / Linear that consumes it (the affine-only norm in front of a Transformer2D's proj_in):
//   proj_in(GN(x))[b, p, n] = sum_k W[n][k] (a[b][k] x[b,p,k] + s[b][k]) + bias[n]
//                           = sum_k (W[n][k] a[b][k]) x[b,p,k]  +  (bias[n] + sum_k W[n][k] s[b][k])
// with a[b][k] = rstd[b, g(k)] gamma[k], s[b][k] = beta[k] - mean[b, g(k)] a[b][k]: per SAMPLE a C x C weight matrix (bf16,
// rounded once after the fp32 scaling) and a bias row - 3.3 MB at C = 320 for 16 samples against the 84 MB the apply pass
// reads and writes at 64 x 64.  The statistics are finished in the prologue exactly as k_gn_apply_fin does (from the
// producer's column statistics or from k_gn_partial's partials, same order); workgroup (row block, sample) then writes its
// rows of W'_b and bias'_b.  The GEMM reads W'_b through GemmParams::w_sample_stride and bias'_b as its per-sample row bias.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gn_fold(GnParams p, const bf16_t* __restrict__ W, const float* __restrict__ bias, int N,
                                                 bf16_t* __restrict__ Wf, float* __restrict__ bf, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* ab = (float*)smem_raw;                 // [2][C]
    float* red_s = ab + 2 * p.C;                  // [256]
    float* red_q = red_s + 256;
    float* mean_s = red_q + 256;
    float* rstd_s = mean_s + 256;
    const int n = blockIdx.y;
    const int cpg = p.C / p.G;
    {
        const float cnt = (float)p.HW * (float)cpg;
        int parts = 256 / p.G;
        if (parts < 1) parts = 1;
        const int g = threadIdx.x % p.G, part = threadIdx.x / p.G;
        float a = 0.f, b = 0.f;
        auto gather = [&](const float2* src, int chunks, int row, int u0, int u1) {     // as in k_gn_apply_fin: four partials in flight
            const int nu = u1 - u0;
            if (nu <= 0 || part >= chunks) return;
            const int total = ((chunks - part + parts - 1) / parts) * nu;
            for (int i0 = 0; i0 < total; i0 += 4) {
                float2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j;
                    if (i < total) t[j] = src[(size_t)(part + (i / nu) * parts) * row + u0 + i % nu];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < total) { a += t[j].x; b += t[j].y; }
            }
        };
        if (p.cs_x) {
            const int unit = p.cs_unit, nu = p.C / unit;
            if (part < parts)
                gather((const float2*)p.cs_x + (size_t)n * p.cs_x_chunks * nu, p.cs_x_chunks, nu, g * cpg / unit, (g + 1) * cpg / unit);
        } else if (part < parts)
            gather((const float2*)p.partial + (size_t)n * p.nchunks * p.G, p.nchunks, p.G, g, g + 1);
        red_s[threadIdx.x] = a; red_q[threadIdx.x] = b;
        __syncthreads();
        if (threadIdx.x < p.G) {
            float sa = 0.f, sb = 0.f;
            for (int q = 0; q < parts; ++q) { sa += red_s[q * p.G + g]; sb += red_q[q * p.G + g]; }
            const float mean = sa / cnt;
            const float var = fmaxf(sb / cnt - mean * mean, 0.f);
            mean_s[g] = mean;
            rstd_s[g] = rsqrtf(var + p.eps);
        }
        __syncthreads();
        for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
            const int gg = c / cpg;
            const float aa = rstd_s[gg] * p.gamma[c];
            ab[c] = aa;
            ab[p.C + c] = p.beta[c] - mean_s[gg] * aa;
        }
        __syncthreads();
    }
    // one wave per weight row: 8 channels per lane and step, the shift term reduced over the wave in a fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r_end = min(N, (int)(blockIdx.x + 1) * rows_per_block);
    for (int r = blockIdx.x * rows_per_block + wave; r < r_end; r += 4) {
        const bf16_t* wr = W + (size_t)r * p.C;
        bf16_t* wo = Wf + ((size_t)n * N + r) * p.C;
        float acc = 0.f;
        for (int c = lane * 8; c < p.C; c += 512) {
            float f[8];
            unpack8(*(const uint4*)(wr + c), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc = fmaf(f[j], ab[p.C + c + j], acc); f[j] *= ab[c + j]; }
            *(uint4*)(wo + c) = pack8(f);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) bf[(size_t)n * N + r] = acc + (bias ? bias[r] : 0.f);
    }
}
int launch_gn_fold(hipStream_t st, const GnParams& p, const bf16_t* W, const float* bias, int N, bf16_t* Wf, float* bf) {
    if (p.C % 8 || p.C % p.G || p.G > 256 || p.C1 != p.C) GYRE_FAIL(-1, "gn_fold: one source, C a multiple of 8 and of the groups");
    if (p.cs_x && (p.cs_unit <= 0 || (p.C / p.G) % p.cs_unit || p.cs_x_chunks <= 0)) GYRE_FAIL(-1, "gn_fold: bad producer statistics");
    // one row per wave: the row loop is a chain of dependent global round trips (32 rows per workgroup took 13 us for 3.4 MB);
    // the statistics prologue is cheap enough to repeat in every workgroup
    const int rows_per_block = 4;
    const size_t lds = ((size_t)2 * p.C + 4 * 256) * sizeof(float);
    GyreProfScope prof_(KC_GN_APPLY, st, 0.0, (double)p.B * N * p.C * 2.0 + (double)N * p.C * 2.0);
    hipLaunchKernelGGL(k_gn_fold, dim3((N + rows_per_block - 1) / rows_per_block, p.B), dim3(256), lds, st, p, W, bias, N, Wf, bf, rows_per_block);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// GroupNorm(+SiLU) in ONE launch for small feature maps (16x16 / 8x8 UNet levels): one workgroup per
// (sample, group) loads its HW x (C/G) slab once into registers (8-byte vectors), reduces mean, then the
// centred sum of squares (true two-pass variance), normalises and writes.  Replaces three latency-bound
// launches (partial / finalize / apply) where the whole tensor is a few MB.
// ------------------------------------------------------------------------------
#define GNS_MAXV 24
#define GNS_MAXC 256
__global__ __launch_bounds__(256) void k_gn_small(GnParams p) {
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float gb[2][GNS_MAXC];   // the group's gamma / beta, requested with the slab (round 5:
                                                                     // they used to be a dependent round trip behind the statistics)
    const int g = blockIdx.x, n = blockIdx.y;
    const int cpg = p.C / p.G, vpp = cpg / 4;          // 4-channel vectors per pixel
    const bool gb_lds = cpg <= GNS_MAXC;
    if (gb_lds && threadIdx.x < cpg) {                 // read after the two block_sum barriers
        gb[0][threadIdx.x] = p.gamma[g * cpg + threadIdx.x];
        gb[1][threadIdx.x] = p.beta[g * cpg + threadIdx.x];
    }
    const int nvec = p.HW * vpp;
    const int C2 = p.C - p.C1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float f[GNS_MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < GNS_MAXV; ++v) {
        const int idx = threadIdx.x + v * 256;
        if (idx < nvec) {
            const int pix = idx / vpp, c = g * cpg + (idx - pix * vpp) * 4;
            const size_t gp = (size_t)n * p.HW + pix;
            const bf16_t* src = c < p.C1 ? p.x + gp * p.C1 + c : p.x2 + gp * C2 + (c - p.C1);
            const uint2 w = *(const uint2*)src;
            f[v][0] = bf16lo(w.x); f[v][1] = bf16hi(w.x); f[v][2] = bf16lo(w.y); f[v][3] = bf16hi(w.y);
            s += (f[v][0] + f[v][1]) + (f[v][2] + f[v][3]);
        }
    }
    auto block_sum = [&](float v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);   // fixed order: deterministic
    };
    const float cnt = (float)p.HW * (float)cpg;
    const float mean = block_sum(s) / cnt;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < GNS_MAXV; ++v) {
        const int idx = threadIdx.x + v * 256;
        if (idx < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = f[v][j] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(block_sum(q) / cnt + p.eps);
#pragma unroll
    for (int v = 0; v < GNS_MAXV; ++v) {
        const int idx = threadIdx.x + v * 256;
        if (idx < nvec) {
            const int pix = idx / vpp, c = g * cpg + (idx - pix * vpp) * 4;
            const int cl = c - g * cpg;
            const float4 ga = gb_lds ? *(const float4*)(&gb[0][cl]) : *(const float4*)(p.gamma + c);
            const float4 be = gb_lds ? *(const float4*)(&gb[1][cl]) : *(const float4*)(p.beta + c);
            float o[4] = {(f[v][0] - mean) * rstd * ga.x + be.x, (f[v][1] - mean) * rstd * ga.y + be.y,
                          (f[v][2] - mean) * rstd * ga.z + be.z, (f[v][3] - mean) * rstd * ga.w + be.w};
            if (p.silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = silu_f(o[j]);
            }
            *(uint2*)(p.y + ((size_t)n * p.HW + pix) * p.C + c) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
    }
}
bool gn_use_small(int HW, int C, int C1, int G) {
    const int cpg = C / G;
    // measured: wins at 16x16 / 8x8 (16 vs 22 us, 10 vs 24 us); at 32x32 the 3-kernel path with full-row reads is as fast
    return HW <= 256 && (cpg % 4) == 0 && (C1 % 4) == 0 && (size_t)HW * (cpg / 4) <= (size_t)256 * GNS_MAXV;
}
int launch_groupnorm_small(hipStream_t st, const GnParams& p) {
    GyreProfScope prof_(KC_GN_APPLY, st, 0.0, (double)p.B * p.HW * p.C * 4.0);
    hipLaunchKernelGGL(k_gn_small, dim3(p.G, p.B), dim3(256), 0, st, p);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// LayerNorm over the last axis of [M][C] bf16, one wave per row, two-pass in registers
// (mean, then centred sum of squares - same arithmetic order class as ATen's).
// ------------------------------------------------------------------------------
#define LN_MAXV 4  // C <= 8*64*4 = 2048
// STATS: only the row statistics (rstd, rstd * mean) leave the kernel - the consuming GEMM normalises in its epilogue
// (GemmParams::ln_stats)
template <bool STATS>
__global__ __launch_bounds__(256) void k_layernorm(const bf16_t* __restrict__ x, int M, int C,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float eps, bf16_t* __restrict__ y, float* __restrict__ stats) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    size_t row = (size_t)blockIdx.x * 4 + wave;
    if (row >= (size_t)M) return;
    const int CV = C / 8;
    const bf16_t* xr = x + row * C;
    float f[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < LN_MAXV; ++v) {
        int cv = lane + v * 64;
        if (cv < CV) {
            unpack8(*(const uint4*)(xr + cv * 8), f[v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[v][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < LN_MAXV; ++v) {
        int cv = lane + v * 64;
        if (cv < CV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float d = f[v][j] - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (STATS) {
        if (lane == 0) *(float2*)(stats + row * 2) = make_float2(rstd, rstd * mean);
        return;
    }
#pragma unroll
    for (int v = 0; v < LN_MAXV; ++v) {
        int cv = lane + v * 64;
        if (cv < CV) {
            const float* g = gamma + cv * 8;
            const float* b = beta + cv * 8;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f[v][j] - mean) * rstd * g[j] + b[j];
            *(uint4*)(y + row * C + cv * 8) = pack8(o);
        }
    }
}
int launch_layernorm(hipStream_t st, const bf16_t* x, int M, int C, const float* gamma, const float* beta, float eps,
                     bf16_t* y) {
    if (C % 8 || C > 8 * 64 * LN_MAXV) GYRE_FAIL(-6, "layernorm: C must be a multiple of 8 and <= 2048");
    GyreProfScope prof_(KC_LAYERNORM, st, 0.0, (double)M * C * 4.0);
    hipLaunchKernelGGL(k_layernorm<false>, dim3((M + 3) / 4), dim3(256), 0, st, x, M, C, gamma, beta, eps, y, (float*)nullptr);
    GYRE_LAUNCH_CHECK();
    return 0;
}
int launch_layernorm_stats(hipStream_t st, const bf16_t* x, int M, int C, float eps, float* stats) {
    if (C % 8 || C > 8 * 64 * LN_MAXV) GYRE_FAIL(-6, "layernorm: C must be a multiple of 8 and <= 2048");
    GyreProfScope prof_(KC_LAYERNORM, st, 0.0, (double)M * C * 2.0);
    hipLaunchKernelGGL(k_layernorm<true>, dim3((M + 3) / 4), dim3(256), 0, st, x, M, C, (const float*)nullptr, (const float*)nullptr,
                       eps, (bf16_t*)nullptr, stats);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// weight repack (once, at load): conv OIHW -> [O][KH][KW][Ipad] bf16 (K contiguous for the
// implicit GEMM); linear [O][I] -> bf16, optionally with the GEGLU 16-row value/gate interleave:
// new row 32p+i = value row 16p+i, new row 32p+16+i = gate row F+16p+i  (F = O/2).
// ------------------------------------------------------------------------------
__global__ void k_repack_conv(const void* __restrict__ w, int dtype, int O, int I, int KH, int KW, int Ipad,
                              bf16_t* __restrict__ out, size_t total, float scale) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int ci = (int)(idx % Ipad);
    size_t r = idx / Ipad;
    int kx = (int)(r % KW); r /= KW;
    int ky = (int)(r % KH); r /= KH;
    int o = (int)r;
    float v = ci < I ? load_as_f32(w, dtype, (((size_t)o * I + ci) * KH + ky) * KW + kx) : 0.f;
    out[idx] = f32_to_bf16(v * scale);   // scale applied in fp32 before the single bf16 rounding
}
int launch_repack_conv(hipStream_t st, const void* w, int dtype, int O, int I, int KH, int KW, int Ipad, bf16_t* out,
                       float scale) {
    size_t total = (size_t)O * KH * KW * Ipad;
    hipLaunchKernelGGL(k_repack_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, dtype, O, I, KH, KW,
                       Ipad, out, total, scale);
    GYRE_LAUNCH_CHECK();
    return 0;
}
__device__ __forceinline__ int geglu_src_row(int r, int F) {
    int p = r >> 5, i = r & 31;
    return i < 16 ? p * 16 + i : F + p * 16 + (i - 16);
}
__global__ void k_repack_linear(const void* __restrict__ w, int dtype, int O, int I, int inter,
                                bf16_t* __restrict__ out, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int k = (int)(idx % I);
    int r = (int)(idx / I);
    int sr = inter ? geglu_src_row(r, O / 2) : r;
    out[idx] = f32_to_bf16(load_as_f32(w, dtype, (size_t)sr * I + k));
}
int launch_repack_linear(hipStream_t st, const void* w, int dtype, int O, int I, int inter, bf16_t* out) {
    if (inter && (O % 32)) GYRE_FAIL(-1, "geglu interleave needs O % 32 == 0");
    size_t total = (size_t)O * I;
    hipLaunchKernelGGL(k_repack_linear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, dtype, O, I, inter,
                       out, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
__global__ void k_cast_f32(const void* __restrict__ w, int dtype, size_t n, int inter, float* __restrict__ out, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t s = inter ? (size_t)geglu_src_row((int)i, (int)(n / 2)) : i;
    out[i] = load_as_f32(w, dtype, s) * scale;
}
int launch_cast_f32(hipStream_t st, const void* w, int dtype, size_t n, int inter, float* out, float scale) {
    hipLaunchKernelGGL(k_cast_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, dtype, n, inter, out, scale);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// 16 B/lane streaming copy: calibrates the achievable HBM rate on the box (bench.py roofline).
__global__ __launch_bounds__(256) void k_copy_probe(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}
int launch_copy_probe(hipStream_t st, const void* src, void* dst, size_t bytes) {
    size_t n = bytes / 16;
    hipLaunchKernelGGL(k_copy_probe, dim3(2048), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ---- debug: verify the caller's batch-structure hints on the device (GYRE_VERIFY_HINTS=1) ---------------------------------
// flag[0] |= 1 when some sample's timestep differs from sample 0's (gyre_unet_hint_uniform_timestep was wrong)
// flag[0] |= 2 when sample b and sample b + B/2 differ in any input byte (gyre_unet_hint_cfg_pairs was wrong)
__global__ __launch_bounds__(256) void k_verify_hints(const int64_t* t, int B, int check_t, const uint32_t* x, size_t words_per_half,
                                                      int check_pairs, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int bad = 0;
    if (check_t && i < (size_t)B && t[i] != t[0]) bad |= 1;
    if (check_pairs && i < words_per_half && x[i] != x[words_per_half + i]) bad |= 2;
    if (bad) atomicOr(flag, bad);
}
int launch_verify_hints(hipStream_t st, const int64_t* t, int B, int check_t, const void* x, size_t bytes_per_half, int check_pairs, int* flag) {
    if (bytes_per_half % 4) check_pairs = 0;          // (never: the boundary tensors are 2- or 4-byte types with an even element count)
    const size_t words = check_pairs ? bytes_per_half / 4 : 0;
    const size_t n = std::max<size_t>(words, check_t ? (size_t)B : 0);
    if (!n) return 0;
    hipLaunchKernelGGL(k_verify_hints, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t, B, check_t, (const uint32_t*)x, words, check_pairs, flag);
    GYRE_LAUNCH_CHECK();
    return 0;
}
