// Input-gradient (vector-Jacobian product) kernels for the CLIP-guided denoising mode.
//
// The reference's ClipGuidedMode (gyre/pipeline/unet/clipguided.py:301-338,420) asks autograd for
// d loss / d latents through the guided UNet evaluation and the VAE decoder; weights never receive gradients.
// These are the transposes of the forward kernels for that one direction:
//   k_gn_bwd_partial / k_gn_bwd_apply                       GroupNorm(+SiLU), concat-aware like the forward
//   k_ln_bwd                                                 LayerNorm (+ the residual branch's gradient)
//   k_geglu_bwd                                              GEGLU on the 16-value / 16-gate interleaved columns
//   k_attn_bwd_delta / k_attn_bwd_dq / k_attn_bwd_dkv        flash-style attention backward on v_mfma_f32_32x32x16_bf16:
//                                                            probabilities are recomputed from Q, K and a row
//                                                            log-sum-exp, never stored
//   k_pool2_sum / k_zero_stuff2                              adjoints of nearest 2x upsampling / stride-2 sampling
//   k_conv_weight_t / k_transpose                            W^T operands of the data-gradient GEMMs / convs
// Gradients are NHWC bf16 like the activations; every reduction accumulates in fp32 in a fixed order.
#include "kernels.h"
#include <type_traits>
#include "gemm_shared.h"

// ------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward.  With u = a*x + b (a = rstd*gamma, b = beta - mean*a, from the forward statistics
// kernels), g = dy * silu'(u) * gamma and xh = (x - mean) * rstd:
//     dx = rstd * (g - mean_grp(g) - xh * mean_grp(g * xh))
// Stage 1 sums g and g*xh per (sample, pixel chunk, group); stage 2 reduces the chunks in a fixed order (in its prologue) and applies.
// ------------------------------------------------------------------------------
#define GNB_MAXV 4
// d silu / du = s (1 + u (1 - s)), s = sigmoid(u): hardware reciprocal and exp2 (the IEEE division of 1 / (1 + e^-u) was a
// ten-instruction sequence per value, as in the forward kernels' silu_f)
__device__ __forceinline__ float silu_grad_f(float u) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));
    return s * fmaf(u, 1.0f - s, 1.0f);
}
// group of channel c + j for the 8 consecutive channels of one vector (cpg channels per group): one division per vector when a
// group is at least as wide as the vector (every UNet / VAE GroupNorm: cpg >= 4 ... 40), else per channel
__device__ __forceinline__ void gn_groups8(int c, int cpg, int (&grp)[8]) {
    const int g0 = c / cpg, r0 = c - g0 * cpg;
    if (cpg >= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) grp[j] = g0 + (r0 + j >= cpg ? 1 : 0);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) grp[j] = (c + j) / cpg;
    }
}
// Forward statistics of sample n, finished in the prologue of BOTH backward kernels from k_gn_partial's partial sums - the
// arithmetic and the order of k_gn_finalize / k_gn_apply_fin (thread (g, part) adds every parts-th chunk, then the parts in
// order), so mean / rstd carry the bits the forward pass normalised with.  Round 6: this replaced two launches per GroupNorm
// (k_gn_finalize, k_gn_bwd_finalize - the latter a 32-thread chain of nchunks dependent loads, 18 us at 64 x 64) of the five.
// stat: [4][256] floats of LDS (red_s, red_q, mean, rstd); all 256 threads call it.
__device__ __forceinline__ void gnb_forward_stats(const GnBwdParams& p, int n, float* stat) {
    float *red_s = stat, *red_q = stat + 256, *mean_s = stat + 512, *rstd_s = stat + 768;
    const int cpg = p.C / p.G;
    const float cnt = (float)p.HW * (float)cpg;
    int parts = 256 / p.G;
    if (parts < 1) parts = 1;
    const int g = threadIdx.x % p.G, part = threadIdx.x / p.G;
    float a = 0.f, b = 0.f;
    if (part < parts) {
        const float2* src = (const float2*)p.fwd_partial + (size_t)n * p.nchunks * p.G + g;
        const int total = (p.nchunks - part + parts - 1) / parts;
        for (int i0 = 0; i0 < total; i0 += 4) {          // four partials in flight, added in the original order
            float2 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < total) t[j] = src[(size_t)(part + (i0 + j) * parts) * p.G];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < total) { a += t[j].x; b += t[j].y; }
        }
    }
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
}
template <int NV>      // channel vectors per thread: 1 for C <= 2048
__global__ __launch_bounds__(256) void k_gn_bwd_partial(GnBwdParams p, int TX, int PY, int pix_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float stat[4 * 256];
    float* red = (float*)smem_raw;  // [PY][C][2]
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int CV = p.C / 8, C2 = p.C - p.C1, cpg = p.C / p.G;
    const int p0 = chunk * pix_per_chunk, p1 = min(p.HW, p0 + pix_per_chunk);
    gnb_forward_stats(p, n, stat);
    const float *mean_s = stat + 512, *rstd_s = stat + 768;
    float s[NV][8], ss[NV][8];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[v][j] = 0.f; ss[v][j] = 0.f; }
    if (ty < PY) {
        // per-channel constants of this thread's vectors, read once (they were re-read - and the group index re-divided - per pixel)
        float ka[NV][8], kb[NV][8], kg[NV][8], km[NV][8], kr[NV][8];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cv = tx + v * TX;
            if (cv < CV) {
                const int c = cv * 8;
                int grp[8];
                gn_groups8(c, cpg, grp);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kg[v][j] = p.gamma[c + j]; km[v][j] = mean_s[grp[j]]; kr[v][j] = rstd_s[grp[j]];
                    ka[v][j] = kr[v][j] * kg[v][j];                       // a = rstd * gamma, b = beta - mean * a (k_gn_finalize)
                    kb[v][j] = p.beta[c + j] - km[v][j] * ka[v][j];
                }
            }
        }
        for (int pix = p0 + ty; pix < p1; pix += PY) {
            const size_t gp = (size_t)n * p.HW + pix;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int cv = tx + v * TX;
                if (cv < CV) {
                    const int c = cv * 8;
                    const bf16_t* src = c < p.C1 ? p.x + gp * p.C1 + c : p.x2 + gp * C2 + (c - p.C1);
                    float f[8], d[8];
                    unpack8(*(const uint4*)src, f);
                    unpack8(*(const uint4*)(p.dy + gp * p.C + c), d);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float u = fmaf(ka[v][j], f[j], kb[v][j]);
                        const float g = d[j] * (p.silu ? silu_grad_f(u) : 1.0f) * kg[v][j];
                        s[v][j] += g;
                        ss[v][j] = fmaf(g, (f[j] - km[v][j]) * kr[v][j], ss[v][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int cv = tx + v * TX;
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
    for (int g = threadIdx.x; g < p.G; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int y = 0; y < PY; ++y)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
                a += red[((size_t)y * p.C + c) * 2 + 0];
                b += red[((size_t)y * p.C + c) * 2 + 1];
            }
        float* dst = p.partial + (((size_t)n * p.nchunks + chunk) * p.G + g) * 2;
        dst[0] = a; dst[1] = b;
    }
}
// Apply: workgroup = (pixel chunk, sample) like the partial kernel.  Prologue: the forward statistics (above) and the group means of
// g and g * xh from the partial kernel's sums (thread (g, part) adds every parts-th chunk, then the parts in order); then the lane
// keeps ONE channel vector's constants in registers and walks its chunk's pixels, four rows requested per trip.
template <int NV>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(GnBwdParams p, int TX, int PY, int pix_per_chunk) {
    __shared__ float stat[4 * 256];
    __shared__ float cred[2 * 256];
    __shared__ float coef[2 * 256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int CV = p.C / 8, C2 = p.C - p.C1, cpg = p.C / p.G;
    const int p0 = chunk * pix_per_chunk, p1 = min(p.HW, p0 + pix_per_chunk);
    {
        int parts = 256 / p.G;
        if (parts < 1) parts = 1;
        const int g = threadIdx.x % p.G, part = threadIdx.x / p.G;
        float a = 0.f, b = 0.f;
        if (part < parts) {
            const float2* src = (const float2*)p.partial + (size_t)n * p.nchunks * p.G + g;
            const int total = (p.nchunks - part + parts - 1) / parts;
            for (int i0 = 0; i0 < total; i0 += 4) {
                float2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < total) t[j] = src[(size_t)(part + (i0 + j) * parts) * p.G];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j < total) { a += t[j].x; b += t[j].y; }
            }
        }
        cred[threadIdx.x] = a; cred[256 + threadIdx.x] = b;
    }
    gnb_forward_stats(p, n, stat);                   // (its first barrier also publishes cred)
    const float *mean_s = stat + 512, *rstd_s = stat + 768;
    if (threadIdx.x < p.G) {
        int parts = 256 / p.G;
        if (parts < 1) parts = 1;
        const float cnt = (float)p.HW * (float)cpg;
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < parts; ++q) { sa += cred[q * p.G + threadIdx.x]; sb += cred[256 + q * p.G + threadIdx.x]; }
        coef[threadIdx.x] = sa / cnt; coef[256 + threadIdx.x] = sb / cnt;
    }
    __syncthreads();
    if (ty >= PY) return;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int cv = tx + v * TX;
        if (cv >= CV) continue;
        const int c = cv * 8;
        const bool first = c < p.C1;
        const bool has_add = first && p.addend;
        float ka[8], kb[8], kg[8], km[8], kr[8], k0[8], k1[8];
        int grp[8];
        gn_groups8(c, cpg, grp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kg[j] = p.gamma[c + j]; km[j] = mean_s[grp[j]]; kr[j] = rstd_s[grp[j]];
            ka[j] = kr[j] * kg[j];
            kb[j] = p.beta[c + j] - km[j] * ka[j];
            k0[j] = coef[grp[j]]; k1[j] = coef[256 + grp[j]];
        }
        const bf16_t* xs = first ? p.x + c : p.x2 + (c - p.C1);
        bf16_t* xd = first ? p.dx + c : p.dx2 + (c - p.C1);
        const size_t ld = first ? (size_t)p.C1 : (size_t)C2;
        for (int pix = p0 + ty; pix < p1; pix += 4 * PY) {
            uint4 rx[4], rd[4], ra[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * PY;
                if (px < p1) {
                    const size_t gp = (size_t)n * p.HW + px;
                    rx[u] = *(const uint4*)(xs + gp * ld);
                    rd[u] = *(const uint4*)(p.dy + gp * p.C + c);
                    if (has_add) ra[u] = *(const uint4*)(p.addend + gp * p.C1 + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = pix + u * PY;
                if (px >= p1) continue;
                const size_t gp = (size_t)n * p.HW + px;
                float f[8], d[8], o[8], ad[8];
                unpack8(rx[u], f);
                unpack8(rd[u], d);
                if (has_add) unpack8(ra[u], ad);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float uu = fmaf(ka[j], f[j], kb[j]);
                    const float g = d[j] * (p.silu ? silu_grad_f(uu) : 1.0f) * kg[j];
                    const float xh = (f[j] - km[j]) * kr[j];
                    o[j] = kr[j] * (g - k0[j] - xh * k1[j]) + (has_add ? ad[j] : 0.f);
                }
                *(uint4*)(xd + gp * ld) = pack8(o);
            }
        }
    }
}
size_t gn_bwd_workspace_bytes(int B, int HW, int C, int G) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int nch = gn_pick_chunks(B, HW, C);
    return gn_workspace_bytes(B, HW, C, G) + al((size_t)B * nch * G * 2 * 4) + 2 * al((size_t)B * G * 2 * 4);
}
int launch_groupnorm_bwd(hipStream_t st, GnBwdParams p, void* ws) {
    if (p.C % 8 || p.C1 % 8 || p.C % p.G || p.G > 256) GYRE_FAIL(-1, "groupnorm_bwd: C, C1 multiples of 8, C of groups (<= 256)");
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // forward statistics: the forward pass's own partial sums (same kernel, same order); mean / rstd are finished from them in
    // the prologues of the two kernels below
    GnParams f;
    f.x = p.x; f.x2 = p.x2 ? p.x2 : p.x; f.C1 = p.C1; f.B = p.B; f.HW = p.HW; f.C = p.C; f.G = p.G;
    f.gamma = p.gamma; f.beta = p.beta; f.eps = p.eps; f.silu = p.silu; f.y = nullptr;
    f.nchunks = gn_pick_chunks(p.B, p.HW, p.C);
    char* w = (char*)ws;
    f.partial = (float*)w;
    f.scale_shift = (float*)(w + al((size_t)p.B * f.nchunks * p.G * 2 * 4));
    w += gn_workspace_bytes(p.B, p.HW, p.C, p.G);
    p.partial = (float*)w;
    f.mean_rstd = nullptr;
    p.fwd_partial = f.partial; p.nchunks = f.nchunks;
    if (!p.x2) p.x2 = p.x;
    int rc = launch_groupnorm_stats(st, f);
    if (rc) return rc;
    GyreProfScope prof_(KC_NORM_BWD, st, 0.0, (double)p.B * p.HW * p.C * 2.0 * 5.0);     // x, dy twice; dx once
    const int CV = p.C / 8;
    const int TX = CV < 256 ? CV : 256;
    if ((CV + TX - 1) / TX > GNB_MAXV) GYRE_FAIL(-6, "groupnorm_bwd: C too large");
    int PY = 256 / TX; if (PY < 1) PY = 1;
    const int ppc = (p.HW + p.nchunks - 1) / p.nchunks;
    const size_t lds = (size_t)PY * p.C * 2 * sizeof(float);
    if (lds > 150 * 1024) GYRE_FAIL(-6, "groupnorm_bwd: LDS budget exceeded");
    const int nv = (CV + TX - 1) / TX;
    const dim3 grid(p.nchunks, p.B);
    if (nv == 1) hipLaunchKernelGGL(k_gn_bwd_partial<1>, grid, dim3(256), lds, st, p, TX, PY, ppc);
    else if (nv == 2) hipLaunchKernelGGL(k_gn_bwd_partial<2>, grid, dim3(256), lds, st, p, TX, PY, ppc);
    else hipLaunchKernelGGL(k_gn_bwd_partial<GNB_MAXV>, grid, dim3(256), lds, st, p, TX, PY, ppc);
    GYRE_LAUNCH_CHECK();
    if (nv == 1) hipLaunchKernelGGL(k_gn_bwd_apply<1>, grid, dim3(256), 0, st, p, TX, PY, ppc);
    else if (nv == 2) hipLaunchKernelGGL(k_gn_bwd_apply<2>, grid, dim3(256), 0, st, p, TX, PY, ppc);
    else hipLaunchKernelGGL(k_gn_bwd_apply<GNB_MAXV>, grid, dim3(256), 0, st, p, TX, PY, ppc);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// LayerNorm backward, one wave per row:  g = dy*gamma, dx = rstd*(g - mean(g) - xh*mean(g*xh)) (+ addend)
// ------------------------------------------------------------------------------
#define LNB_MAXV 4
__global__ __launch_bounds__(256) void k_ln_bwd(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, int M, int C,
                                                const float* __restrict__ gamma, float eps,
                                                const bf16_t* __restrict__ addend, bf16_t* __restrict__ dx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + wave;
    if (row >= (size_t)M) return;
    const int CV = C / 8;
    float f[LNB_MAXV][8], g[LNB_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < LNB_MAXV; ++v) {
        const int cv = lane + v * 64;
        if (cv < CV) {
            unpack8(*(const uint4*)(x + row * C + cv * 8), f[v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[v][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < LNB_MAXV; ++v) {
        const int cv = lane + v * 64;
        if (cv < CV) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = f[v][j] - mean; q = fmaf(d, d, q); }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < LNB_MAXV; ++v) {
        const int cv = lane + v * 64;
        if (cv < CV) {
            float d[8];
            unpack8(*(const uint4*)(dy + row * C + cv * 8), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f[v][j] = (f[v][j] - mean) * rstd;
                g[v][j] = d[j] * gamma[cv * 8 + j];
                s1 += g[v][j];
                s2 = fmaf(g[v][j], f[v][j], s2);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int v = 0; v < LNB_MAXV; ++v) {
        const int cv = lane + v * 64;
        if (cv < CV) {
            float o[8], ad[8];
            if (addend) unpack8(*(const uint4*)(addend + row * C + cv * 8), ad);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (g[v][j] - m1 - f[v][j] * m2) + (addend ? ad[j] : 0.f);
            *(uint4*)(dx + row * C + cv * 8) = pack8(o);
        }
    }
}
int launch_layernorm_bwd(hipStream_t st, const bf16_t* x, const bf16_t* dy, int M, int C, const float* gamma, float eps,
                         const bf16_t* addend, bf16_t* dx) {
    if (C % 8 || C > 8 * 64 * LNB_MAXV) GYRE_FAIL(-6, "layernorm_bwd: C must be a multiple of 8 and <= 2048");
    GyreProfScope prof_(KC_NORM_BWD, st, 0.0, (double)M * C * 2.0 * (addend ? 4.0 : 3.0));
    hipLaunchKernelGGL(k_ln_bwd, dim3((M + 3) / 4), dim3(256), 0, st, x, dy, M, C, gamma, eps, addend, dx);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// GEGLU backward on the interleaved pre-activation: columns 32p..32p+15 hold the values, 32p+16..32p+31 the gates of
// output columns 16p..16p+15 (launch_repack_linear).  out = val * gelu(gate):
//   d val = dy * gelu(gate),   d gate = dy * val * (Phi(gate) + gate * phi(gate))
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_geglu_bwd(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ dy, size_t M,
                                                   int F, bf16_t* __restrict__ dpre) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int FV = F / 8;
    if (idx >= M * FV) return;
    const size_t row = idx / FV;
    const int c = (int)(idx % FV) * 8;                       // output column of this 8-vector
    const size_t base = row * (size_t)(2 * F) + (size_t)(c >> 4) * 32 + (c & 15);
    float v[8], gt[8], d[8], dv[8], dg[8];
    unpack8(*(const uint4*)(pre + base), v);
    unpack8(*(const uint4*)(pre + base + 16), gt);
    unpack8(*(const uint4*)(dy + row * F + c), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = gt[j];
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        dv[j] = d[j] * x * cdf;
        dg[j] = d[j] * v[j] * fmaf(x, pdf, cdf);
    }
    *(uint4*)(dpre + base) = pack8(dv);
    *(uint4*)(dpre + base + 16) = pack8(dg);
}
int launch_geglu_bwd(hipStream_t st, const bf16_t* pre, const bf16_t* dy, size_t M, int F, bf16_t* dpre) {
    if (F % 16) GYRE_FAIL(-1, "geglu_bwd: F must be a multiple of 16");
    const size_t total = M * (size_t)(F / 8);
    hipLaunchKernelGGL(k_geglu_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pre, dy, M, F, dpre);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// resampling adjoints, elementwise add, operand transposes
// ------------------------------------------------------------------------------
// nearest upsampling to (Hu, Wu) <= (2H, 2W) duplicated source pixels; the adjoint sums the (up to 4) copies
__global__ __launch_bounds__(256) void k_pool2_sum(const bf16_t* __restrict__ du, int B, int H, int W, int Hu, int Wu, int C,
                                                   bf16_t* __restrict__ dx, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int CV = C / 8;
    const int c = (int)(idx % CV) * 8;
    size_t r = idx / CV;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int yy = 2 * y + a, xx = 2 * x + b;
            if (yy < Hu && xx < Wu) {
                float f[8];
                unpack8(*(const uint4*)(du + (((size_t)n * Hu + yy) * Wu + xx) * C + c), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        }
    *(uint4*)(dx + (((size_t)n * H + y) * W + x) * C + c) = pack8(acc);
}
int launch_pool2_sum(hipStream_t st, const bf16_t* du, int B, int H, int W, int Hu, int Wu, int C, bf16_t* dx) {
    if (C % 8) GYRE_FAIL(-1, "pool2_sum: C must be a multiple of 8");
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(k_pool2_sum, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, du, B, H, W, Hu, Wu, C, dx, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// stride-2 sampling picked input pixel (2oy + ky - pad, ...): its adjoint is a stride-1 convolution with the flipped
// kernel over dy spread onto the even positions of an (H, W) grid
__global__ __launch_bounds__(256) void k_zero_stuff2(const bf16_t* __restrict__ dy, int B, int Ho, int Wo, int H, int W, int C,
                                                     bf16_t* __restrict__ dz, size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int CV = C / 8;
    const int c = (int)(idx % CV) * 8;
    size_t r = idx / CV;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
        v = *(const uint4*)(dy + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
    *(uint4*)(dz + (((size_t)n * H + y) * W + x) * C + c) = v;
}
int launch_zero_stuff2(hipStream_t st, const bf16_t* dy, int B, int Ho, int Wo, int H, int W, int C, bf16_t* dz) {
    if (C % 8) GYRE_FAIL(-1, "zero_stuff2: C must be a multiple of 8");
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(k_zero_stuff2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, B, Ho, Wo, H, W, C, dz, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(256) void k_add_bf16(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvec) return;
    float a[8], b[8];
    unpack8(((const uint4*)y)[i], a);
    unpack8(((const uint4*)x)[i], b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    ((uint4*)y)[i] = pack8(a);
}
int launch_add_bf16(hipStream_t st, bf16_t* y, const bf16_t* x, size_t n) {
    if (n % 8) GYRE_FAIL(-1, "add_bf16: n must be a multiple of 8");
    hipLaunchKernelGGL(k_add_bf16, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, y, x, n / 8);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// conv weight [O][3][3][I] -> [I][3][3][O] with the window rotated by 180 degrees (the data-gradient convolution)
__global__ __launch_bounds__(256) void k_conv_weight_t(const bf16_t* __restrict__ w, int O, int I, bf16_t* __restrict__ wt,
                                                       size_t total) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % O);
    const int t = (int)((idx / O) % 9);
    const int i = (int)(idx / ((size_t)O * 9));
    wt[idx] = w[((size_t)o * 9 + (8 - t)) * I + i];
}
int launch_conv_weight_t(hipStream_t st, const bf16_t* w, int O, int I, bf16_t* wt) {
    const size_t total = (size_t)O * 9 * I;
    hipLaunchKernelGGL(k_conv_weight_t, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, O, I, wt, total);
    GYRE_LAUNCH_CHECK();
    return 0;
}
// out[b][c][r] = in[b][r][c] for r < R (zeros for R <= r < ld_out): batched 2-D transpose through LDS
__global__ __launch_bounds__(256) void k_transpose(const bf16_t* __restrict__ in, int ld_in, int R, int C, bf16_t* __restrict__ out,
                                                   int ld_out, size_t bs_in, size_t bs_out) {
    __shared__ bf16_t tile[64][66];
    const bf16_t* src = in + blockIdx.z * bs_in;
    bf16_t* dst = out + blockIdx.z * bs_out;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < ld_out) dst[(size_t)c * ld_out + r] = tile[tx][i];
    }
}
int launch_transpose(hipStream_t st, const bf16_t* in, int ld_in, int R, int C, bf16_t* out, int ld_out, int batch,
                     size_t bs_in, size_t bs_out) {
    if (ld_out < R) GYRE_FAIL(-1, "transpose: ld_out < rows");
    hipLaunchKernelGGL(k_transpose, dim3((ld_out + 63) / 64, (C + 63) / 64, batch), dim3(256), 0, st, in, ld_in, R, C, out,
                       ld_out, bs_in, bs_out);
    GYRE_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------
// Attention backward.  Per (sample, head):  S = Q K^T,  P = softmax(S * scale),  O = P V.  Given dO:
//   delta_i = sum_d dO_id O_id,   dP = dO V^T,   dS = P o (dP - delta),
//   dQ = dS K * scale,   dK = dS^T Q * scale,   dV = P^T dO.
// v_mfma_f32_32x32x16_bf16 register layouts (wave64): A / B operand lane l = row / column l & 31, eight consecutive k at
// 8 * (l >> 5); result lane l = column l & 31, register r = row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
//   k_attn_bwd_dq   (wave = 32 queries) builds S^T = K Q^T so a lane owns ONE query and 16 keys per tile: the softmax
//                   statistics, P^T and dS^T are lane-local, and dS^T is directly the B operand of dQ^T += K^T dS^T.
//   k_attn_bwd_dkv  (wave = 32 keys) builds S = Q K^T so a lane owns ONE key and 16 queries per tile: P and dS are directly
//                   the A operands of dV += P^T dO and dK += dS^T Q.
// In both, the contraction index of the second product is the set of rows a lane holds, i.e. a fixed permutation of the
// tile's 32 rows; the other operand is read in the same permuted order from a transposed copy ([C][tokens]) of K / Q / dO.
// The first pass of k_attn_bwd_dq produces the row log-sum-exp (base 2) both kernels recompute P from.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attn_bwd_delta(AttnBwdParams p) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * Nq * H
    if (idx >= (size_t)p.B * p.Nq * p.H) return;
    const int h = (int)(idx % p.H);
    const size_t row = idx / p.H;          // b * Nq + q
    const int b = (int)(row / p.Nq), q = (int)(row % p.Nq);
    const bf16_t* o = p.o + row * p.ldo + h * p.D;
    const bf16_t* d = p.d_o + row * p.lddo + h * p.D;
    float s = 0.f;
    for (int v = 0; v < p.D; v += 8) {
        float a[8], c[8];
        unpack8(*(const uint4*)(o + v), a);
        unpack8(*(const uint4*)(d + v), c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf(a[j], c[j], s);
    }
    p.delta[((size_t)b * p.H + h) * p.NqPad + q] = s;
}

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* base, int ld, int row, int nrows, int k, int kmax) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nrows && k < kmax) v = *(const uint4*)(base + (size_t)row * ld + k);
    return __builtin_bit_cast(bf16x8_t, v);
}
// eight values of one row of a transposed operand in the permuted order of MFMA step m: tokens t0 + 16m + 4hi + {0..3, 8..11}
__device__ __forceinline__ bf16x8_t ld_frag_t(const bf16_t* rowp, bool row_ok, int t) {
    uint2 a = make_uint2(0, 0), b = make_uint2(0, 0);
    if (row_ok) { a = *(const uint2*)(rowp + t); b = *(const uint2*)(rowp + t + 8); }
    return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& v, int m) {
    return __builtin_bit_cast(bf16x8_t, make_uint4(pack_bf16x2(v[8 * m + 0], v[8 * m + 1]), pack_bf16x2(v[8 * m + 2], v[8 * m + 3]),
                                                   pack_bf16x2(v[8 * m + 4], v[8 * m + 5]), pack_bf16x2(v[8 * m + 6], v[8 * m + 7])));
}

// DS = ceil(D / 16) when the loop-invariant operand fragments (this lane's query rows in the dQ kernel, its key rows in the
// dK / dV kernel) fit in registers (D <= 160), 0 = reload them every tile (VAE: one head of 512 channels)
template <int NDB, int DS>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(AttnBwdParams p, int dchunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int qblk = blockIdx.x / dchunks, dchunk = blockIdx.x % dchunks;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (qblk * 4 + wave) * 32;
    if (q0 >= p.Nq) return;
    const int D = p.D;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    const bf16_t* KT = p.kt + ((size_t)b * p.H * D + (size_t)h * D) * p.ldkt;
    const int q = q0 + col;
    const size_t stat = ((size_t)b * p.H + h) * p.NqPad + q;
    const int nkb = (p.Nk + 31) / 32;
    const float NEG = -1e30f;
    bf16x8_t qh[DS > 0 ? DS : 1], doh[DS > 0 ? DS : 1];
    if constexpr (DS > 0) {
#pragma unroll
        for (int i = 0; i < DS; ++i) {
            qh[i] = ld_frag(Q, p.ldq, q, p.Nq, i * 16 + 8 * hi, D);
            doh[i] = ld_frag(DO, p.lddo, q, p.Nq, i * 16 + 8 * hi, D);
        }
    }

    // pass 1: row maximum and sum of 2^(s - max) over all keys (lane-local over its 16 keys per tile, halves merged last)
    float lse2;
    {
        float mx = NEG, sum = 0.f;
        for (int kb = 0; kb < nkb; ++kb) {
            f32x16_t s = {};
            if constexpr (DS > 0) {
#pragma unroll
                for (int i = 0; i < DS; ++i)
                    s = GYRE_MFMA_32x32x16(ld_frag(K, p.ldk, kb * 32 + col, p.Nk, i * 16 + 8 * hi, D), qh[i], s, 0, 0, 0);
            } else {
                for (int d0 = 0; d0 < D; d0 += 16)
                    s = GYRE_MFMA_32x32x16(ld_frag(K, p.ldk, kb * 32 + col, p.Nk, d0 + 8 * hi, D),
                                                                ld_frag(Q, p.ldq, q, p.Nq, d0 + 8 * hi, D), s, 0, 0, 0);
            }
            float tmx = NEG;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                s[r] = key < p.Nk ? s[r] * p.alpha : NEG;
                tmx = fmaxf(tmx, s[r]);
            }
            const float nm = fmaxf(mx, tmx);
            float ts = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ts += __builtin_amdgcn_exp2f(s[r] - nm);
            sum = sum * __builtin_amdgcn_exp2f(mx - nm) + ts;
            mx = nm;
        }
        const float omx = __shfl_xor(mx, 32), osum = __shfl_xor(sum, 32);
        const float nm = fmaxf(mx, omx);
        sum = sum * __builtin_amdgcn_exp2f(mx - nm) + osum * __builtin_amdgcn_exp2f(omx - nm);
        lse2 = nm + __builtin_amdgcn_logf(sum);     // v_log_f32 is log2
        if (dchunk == 0 && hi == 0 && q < p.Nq) p.lse[stat] = lse2;
    }
    const float delta = q < p.Nq ? p.delta[stat] : 0.f;

    // pass 2: dQ^T[d][q] += K^T[d][keys] dS^T[keys][q]
    f32x16_t acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) acc[i] = f32x16_t{};
    const int dbase = dchunk * NDB * 32;
    for (int kb = 0; kb < nkb; ++kb) {
        f32x16_t s = {}, dp = {};
        if constexpr (DS > 0) {
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                s = GYRE_MFMA_32x32x16(ld_frag(K, p.ldk, kb * 32 + col, p.Nk, i * 16 + 8 * hi, D), qh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(ld_frag(V, p.ldv, kb * 32 + col, p.Nk, i * 16 + 8 * hi, D), doh[i], dp, 0, 0, 0);
            }
        } else {
            for (int d0 = 0; d0 < D; d0 += 16) {
                const bf16x8_t qf = ld_frag(Q, p.ldq, q, p.Nq, d0 + 8 * hi, D);
                const bf16x8_t dof = ld_frag(DO, p.lddo, q, p.Nq, d0 + 8 * hi, D);
                s = GYRE_MFMA_32x32x16(ld_frag(K, p.ldk, kb * 32 + col, p.Nk, d0 + 8 * hi, D), qf, s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(ld_frag(V, p.ldv, kb * 32 + col, p.Nk, d0 + 8 * hi, D), dof, dp, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const float pr = key < p.Nk ? __builtin_amdgcn_exp2f(s[r] * p.alpha - lse2) : 0.f;
            s[r] = pr * (dp[r] - delta);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const bf16x8_t dsf = pack_frag(s, m);
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                const int d = dbase + i * 32 + col;
                acc[i] = GYRE_MFMA_32x32x16(ld_frag_t(KT + (size_t)d * p.ldkt, d < D, kb * 32 + 16 * m + 4 * hi),
                                                                 dsf, acc[i], 0, 0, 0);
            }
        }
    }
    if (q >= p.Nq) return;
    bf16_t* out = p.dq + ((size_t)b * p.Nq + q) * p.lddq + h * D;
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dbase + i * 32 + 8 * g + 4 * hi;
            if (d < D)
                *(uint2*)(out + d) = make_uint2(pack_bf16x2(acc[i][4 * g] * p.beta, acc[i][4 * g + 1] * p.beta),
                                                pack_bf16x2(acc[i][4 * g + 2] * p.beta, acc[i][4 * g + 3] * p.beta));
        }
}

template <int NDB, int DS>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(AttnBwdParams p, int dchunks) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int kblk = blockIdx.x / dchunks, dchunk = blockIdx.x % dchunks;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = (kblk * 4 + wave) * 32;
    if (k0 >= p.Nk) return;
    const int D = p.D;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    const bf16_t* QT = p.qt + ((size_t)b * p.H * D + (size_t)h * D) * p.ldqt;
    const bf16_t* DOT = p.d_ot + ((size_t)b * p.H * D + (size_t)h * D) * p.ldqt;
    const float* LSE = p.lse + ((size_t)b * p.H + h) * p.NqPad;
    const float* DEL = p.delta + ((size_t)b * p.H + h) * p.NqPad;
    const int key = k0 + col;
    const int nqb = (p.Nq + 31) / 32;
    f32x16_t dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) { dk[i] = f32x16_t{}; dv[i] = f32x16_t{}; }
    const int dbase = dchunk * NDB * 32;
    bf16x8_t kh[DS > 0 ? DS : 1], vh[DS > 0 ? DS : 1];
    if constexpr (DS > 0) {
#pragma unroll
        for (int i = 0; i < DS; ++i) {
            kh[i] = ld_frag(K, p.ldk, key, p.Nk, i * 16 + 8 * hi, D);
            vh[i] = ld_frag(V, p.ldv, key, p.Nk, i * 16 + 8 * hi, D);
        }
    }
    for (int qb = 0; qb < nqb; ++qb) {
        f32x16_t s = {}, dp = {};
        if constexpr (DS > 0) {
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                s = GYRE_MFMA_32x32x16(ld_frag(Q, p.ldq, qb * 32 + col, p.Nq, i * 16 + 8 * hi, D), kh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(ld_frag(DO, p.lddo, qb * 32 + col, p.Nq, i * 16 + 8 * hi, D), vh[i], dp, 0, 0, 0);
            }
        } else {
            for (int d0 = 0; d0 < D; d0 += 16) {
                const bf16x8_t kf = ld_frag(K, p.ldk, key, p.Nk, d0 + 8 * hi, D);
                const bf16x8_t vf = ld_frag(V, p.ldv, key, p.Nk, d0 + 8 * hi, D);
                s = GYRE_MFMA_32x32x16(ld_frag(Q, p.ldq, qb * 32 + col, p.Nq, d0 + 8 * hi, D), kf, s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(ld_frag(DO, p.lddo, qb * 32 + col, p.Nq, d0 + 8 * hi, D), vf, dp, 0, 0, 0);
            }
        }
        // lane: key fixed, queries qb*32 + 8g + 4hi + {0..3}; NqPad is a multiple of 32 so the statistics reads stay in range
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qq = qb * 32 + 8 * g + 4 * hi;
            const float4 l4 = *(const float4*)(LSE + qq), d4 = *(const float4*)(DEL + qq);
            const float l[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (qq + r) < p.Nq && key < p.Nk;
                const float pr = ok ? __builtin_amdgcn_exp2f(s[4 * g + r] * p.alpha - l[r]) : 0.f;
                s[4 * g + r] = pr;
                dp[4 * g + r] = pr * (dp[4 * g + r] - dl[r]);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const bf16x8_t pf = pack_frag(s, m), dsf = pack_frag(dp, m);
            const int t = qb * 32 + 16 * m + 4 * hi;
#pragma unroll
            for (int i = 0; i < NDB; ++i) {
                const int d = dbase + i * 32 + col;
                dv[i] = GYRE_MFMA_32x32x16(pf, ld_frag_t(DOT + (size_t)d * p.ldqt, d < D, t), dv[i], 0, 0, 0);
                dk[i] = GYRE_MFMA_32x32x16(dsf, ld_frag_t(QT + (size_t)d * p.ldqt, d < D, t), dk[i], 0, 0, 0);
            }
        }
    }
    // result lane: column = d, register r = key (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const int d = dbase + i * 32 + col;
        if (d >= D) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kk < p.Nk) {
                p.dk[((size_t)b * p.Nk + kk) * p.lddk + h * D + d] = f32_to_bf16(dk[i][r] * p.beta);
                p.dv[((size_t)b * p.Nk + kk) * p.lddv + h * D + d] = f32_to_bf16(dv[i][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------
// LDS-tiled forms (head dim <= 160, i.e. every UNet attention): the tile of the streamed side - 32 queries for the dK / dV
// kernel, 32 keys for the dQ kernel - is fetched ONCE per workgroup into LDS, row-major for the A operands and transposed
// for the permuted B / A operands of the second products, instead of once per wave from L2; that also removes the global
// transposed copies.  Register prefetch of tile t+1 under the MFMAs of tile t, two LDS buffers, one barrier per tile.
// Row strides are 16 x odd bytes (row-major tiles, ds_read_b128) and 72 bytes (transposed tiles, ds_read_b64): conflict-free
// for the fragment reads.
// ------------------------------------------------------------------------------
__device__ __forceinline__ int attn_bwd_row_stride(int D) { return D + (((D >> 3) & 1) ? 0 : 8); }   // elements
// RT = 32-row MFMA tiles per LDS tile; transposed rows hold 32 RT tokens + 4 (72 / 136 bytes)
__host__ __device__ inline size_t attn_bwd_stage_bytes(int D, int RT) {
    const int rs = D + (((D >> 3) & 1) ? 0 : 8), ts = 32 * RT + 4;
    return ((size_t)(2 * 32 * RT * rs + 2 * D * ts) * 2 + 512 + 255) & ~(size_t)255;
}
__device__ __forceinline__ bf16x8_t lds_frag(const bf16_t* tile, int rs, int row, int k, int D) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < D) v = *(const uint4*)(tile + row * rs + k);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t lds_frag_t(const bf16_t* tile_t, int ts, int d, int D, int t) {
    uint2 a = make_uint2(0, 0), b = make_uint2(0, 0);
    if (d < D) { a = *(const uint2*)(tile_t + d * ts + t); b = *(const uint2*)(tile_t + d * ts + t + 8); }
    return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}
// one tile of two row-major [ROWS][D] operands (rows r0 ..) -> registers; then registers -> LDS (row-major + transposed copies)
// register sets of the request ring of the LDS-tile kernels below (a set = 8 registers per 5 row vectors of a 32-row tile pair)
#define ABW_WAVES(DS) ((DS) <= 5 ? 2 : 1)       // resident waves per SIMD the LDS-tile dK / dV kernels are compiled for
#define ABW_WAVES_DQ(DS) ((DS) <= 6 ? 2 : 1)    // ... the dQ kernels (no dK / dV accumulators: D = 80 fits two)
__host__ __device__ constexpr int abw_pf(int DS) { return DS <= 4 ? 4 : (DS <= 6 ? 2 : 1); }
template <int MAXIT, int ROWS>
struct AbwTile {
    uint4 a[MAXIT], b[MAXIT];
    __device__ __forceinline__ void fetch(const bf16_t* A, int lda, const bf16_t* B, int ldb, int r0, int nrows, int D, int tid) {
        const int nv = D >> 3;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + it * 256;
            a[it] = make_uint4(0, 0, 0, 0); b[it] = make_uint4(0, 0, 0, 0);
            if (idx < ROWS * nv) {
                const int r = idx / nv, v = idx - r * nv;
                if (r0 + r < nrows) {
                    a[it] = *(const uint4*)(A + (size_t)(r0 + r) * lda + v * 8);
                    if (B) b[it] = *(const uint4*)(B + (size_t)(r0 + r) * ldb + v * 8);
                }
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* As, bf16_t* Bs, bf16_t* At, bf16_t* Bt, int D, int tid) const {
        const int nv = D >> 3, rs = attn_bwd_row_stride(D);
        constexpr int TS = ROWS + 4;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int idx = tid + it * 256;
            if (idx < ROWS * nv) {
                const int r = idx / nv, v = idx - r * nv;
                *(uint4*)(As + r * rs + v * 8) = a[it];
                if (Bs) *(uint4*)(Bs + r * rs + v * 8) = b[it];
                const uint32_t wa[4] = {a[it].x, a[it].y, a[it].z, a[it].w}, wb[4] = {b[it].x, b[it].y, b[it].z, b[it].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (At) At[(v * 8 + j) * TS + r] = (bf16_t)(wa[j >> 1] >> (16 * (j & 1)));
                    if (Bt) Bt[(v * 8 + j) * TS + r] = (bf16_t)(wb[j >> 1] >> (16 * (j & 1)));
                }
            }
        }
    }
};

template <int NDB, int DS, int RT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ABW_WAVES(DS), ABW_WAVES(DS)))) void k_attn_bwd_dkv_lds(AttnBwdParams p) {
    constexpr int ROWS = 32 * RT, TS = ROWS + 4;
    constexpr int MAXIT = (ROWS * (DS * 2) + 255) / 256;         // D <= 16 DS -> D/8 <= 2 DS vectors per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, D = p.D, rs = attn_bwd_row_stride(D);
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    const float* LSE = p.lse + ((size_t)b * p.H + h) * p.NqPad;
    const float* DEL = p.delta + ((size_t)b * p.H + h) * p.NqPad;
    const size_t stage = attn_bwd_stage_bytes(D, RT);
    auto part = [&](int buf, int which) -> bf16_t* {              // 0 Q, 1 dO, 2 Q^T, 3 dO^T
        bf16_t* base = (bf16_t*)(smem + buf * stage);
        return which < 2 ? base + which * ROWS * rs : base + 2 * ROWS * rs + (which - 2) * D * TS;
    };
    auto stats = [&](int buf) -> float* { return (float*)(smem + buf * stage + (size_t)(2 * ROWS * rs + 2 * D * TS) * 2); };
    const int key = k0 + col;
    bf16x8_t kh[DS], vh[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) {
        kh[i] = ld_frag(K, p.ldk, key, p.Nk, i * 16 + 8 * hi, D);
        vh[i] = ld_frag(V, p.ldv, key, p.Nk, i * 16 + 8 * hi, D);
    }
    f32x16_t dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) { dk[i] = f32x16_t{}; dv[i] = f32x16_t{}; }
    const int nqb = (p.Nq + ROWS - 1) / ROWS;
    // Round 6: a ring of PF register sets.  Tile qb + 1 + PF is requested when tile qb + 1 moves from its registers to LDS, so a
    // request has PF tile computations to arrive (one tile - 14 MFMAs - did not cover an L2 round trip: the loop ran at the load
    // latency, 2.1 us per 32 query rows, the matrix cores 19 % busy).  Eight registers per set at D = 40.
    constexpr int PF = DS == 5 ? 1 : abw_pf(DS);        // (D = 80: one set keeps the kernel at two waves per SIMD without spilling)
    AbwTile<MAXIT, ROWS> tiles[PF];
    float st_pref[PF];
    auto fetch = [&](int qb, AbwTile<MAXIT, ROWS>& tile, float& sp) {
        tile.fetch(Q, p.ldq, DO, p.lddo, qb * ROWS, p.Nq, D, tid);
        sp = 0.f;
        if (tid < 2 * ROWS) { const int q = qb * ROWS + (tid % ROWS); sp = q < p.Nq ? (tid < ROWS ? LSE[q] : DEL[q]) : 0.f; }
    };
    auto store = [&](int buf, const AbwTile<MAXIT, ROWS>& tile, float sp) {
        tile.store(part(buf, 0), part(buf, 1), part(buf, 2), part(buf, 3), D, tid);
        if (tid < 2 * ROWS) stats(buf)[tid] = sp;            // [0,ROWS) LSE, [ROWS,2 ROWS) delta
    };
    auto compute = [&](int qb, int buf) {
        const bf16_t *Qs = part(buf, 0), *DOs = part(buf, 1), *QTs = part(buf, 2), *DOTs = part(buf, 3);
        const float* sts = stats(buf);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x16_t s = {}, dp = {};
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                s = GYRE_MFMA_32x32x16(lds_frag(Qs, rs, rt * 32 + col, i * 16 + 8 * hi, D), kh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(lds_frag(DOs, rs, rt * 32 + col, i * 16 + 8 * hi, D), vh[i], dp, 0, 0, 0);
            }
            // (whole tiles - all but the last of a ragged problem - skip the per-element range tests: wave-uniform branch)
            const bool whole = (qb * ROWS + rt * 32 + 32) <= p.Nq && (k0 + 32) <= p.Nk;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = rt * 32 + 8 * g + 4 * hi;
                const float4 l4 = *(const float4*)(sts + ql), d4 = *(const float4*)(sts + ROWS + ql);
                const float l[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
                if (whole) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + r], p.alpha, -l[r]));
                        s[4 * g + r] = pr;
                        dp[4 * g + r] = pr * (dp[4 * g + r] - dl[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (qb * ROWS + ql + r) < p.Nq && key < p.Nk;
                        const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(s[4 * g + r], p.alpha, -l[r])) : 0.f;
                        s[4 * g + r] = pr;
                        dp[4 * g + r] = pr * (dp[4 * g + r] - dl[r]);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8_t pf = pack_frag(s, m), dsf = pack_frag(dp, m);
                const int t = rt * 32 + 16 * m + 4 * hi;
#pragma unroll
                for (int i = 0; i < NDB; ++i) {
                    const int d = i * 32 + col;
                    dv[i] = GYRE_MFMA_32x32x16(pf, lds_frag_t(DOTs, TS, d, D, t), dv[i], 0, 0, 0);
                    dk[i] = GYRE_MFMA_32x32x16(dsf, lds_frag_t(QTs, TS, d, D, t), dk[i], 0, 0, 0);
                }
            }
        }
    };
    fetch(0, tiles[0], st_pref[0]);
    store(0, tiles[0], st_pref[0]);
#pragma unroll
    for (int j = 1; j <= PF; ++j)
        if (j < nqb) fetch(j, tiles[j % PF], st_pref[j % PF]);
    __syncthreads();
    for (int qb0 = 0; qb0 < nqb; qb0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int qb = qb0 + j;
            if (qb < nqb) {                                       // (workgroup-uniform)
                compute(qb, qb & 1);
                const int nx = (j + 1) % PF;        // (compile-time after the unroll)                  // register set of tile qb + 1
                if (qb + 1 < nqb) store((qb + 1) & 1, tiles[nx], st_pref[nx]);
                if (qb + 1 + PF < nqb) fetch(qb + 1 + PF, tiles[nx], st_pref[nx]);
                __syncthreads();
            }
        }
    }
    if (k0 >= p.Nk) return;
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const int d = i * 32 + col;
        if (d >= D) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kk < p.Nk) {
                p.dk[((size_t)b * p.Nk + kk) * p.lddk + h * D + d] = f32_to_bf16(dk[i][r] * p.beta);
                p.dv[((size_t)b * p.Nk + kk) * p.lddv + h * D + d] = f32_to_bf16(dv[i][r]);
            }
        }
    }
}

template <int NDB, int DS, int RT>
__device__ __forceinline__ void abw_dq_lds_body(const AttnBwdParams& p) {
    constexpr int ROWS = 32 * RT, TS = ROWS + 4;
    constexpr int MAXIT = (ROWS * (DS * 2) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z, D = p.D, rs = attn_bwd_row_stride(D);
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    const size_t stage = attn_bwd_stage_bytes(D, RT);
    auto part = [&](int buf, int which) -> bf16_t* {              // 0 K, 1 V, 2 K^T
        bf16_t* base = (bf16_t*)(smem + buf * stage);
        return which < 2 ? base + which * ROWS * rs : base + 2 * ROWS * rs;
    };
    const int q = q0 + col;
    const size_t stat = ((size_t)b * p.H + h) * p.NqPad + q;
    const int nkb = (p.Nk + ROWS - 1) / ROWS;
    const float NEG = -1e30f;
    bf16x8_t qh[DS], doh[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) {
        qh[i] = ld_frag(Q, p.ldq, q, p.Nq, i * 16 + 8 * hi, D);
        doh[i] = ld_frag(DO, p.lddo, q, p.Nq, i * 16 + 8 * hi, D);
    }
    constexpr int PF = abw_pf(DS);                // request ring, see k_attn_bwd_dkv_lds
    AbwTile<MAXIT, ROWS> tiles[PF];
    const float delta = q < p.Nq ? p.delta[stat] : 0.f;
    // ---- ONE pass (round 6, as k_attn_bwd_dq_dma): dQ^T[d][q] += K^T[d][keys] dS^T[keys][q] against a running reference, divided
    // by the row sum at the end; the separate log-sum-exp pass over K is gone ----
#ifdef GYRE_STORE_F16
    constexpr float RECENTRE = 8.f;
#else
    constexpr float RECENTRE = 20.f;
#endif
    float mref = NEG, lsum = 0.f;
    f32x16_t acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) acc[i] = f32x16_t{};
    auto pass2 = [&](int kb, int buf) {
        const bf16_t *Ks = part(buf, 0), *Vs = part(buf, 1), *KTs = part(buf, 2);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x16_t s = {}, dp = {};
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                s = GYRE_MFMA_32x32x16(lds_frag(Ks, rs, rt * 32 + col, i * 16 + 8 * hi, D), qh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(lds_frag(Vs, rs, rt * 32 + col, i * 16 + 8 * hi, D), doh[i], dp, 0, 0, 0);
            }
            float tmx = NEG;
            if (kb * ROWS + rt * 32 + 32 <= p.Nk) {          // whole key tile: no per-element range test
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] *= p.alpha; tmx = fmaxf(tmx, s[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * ROWS + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[r] = key < p.Nk ? s[r] * p.alpha : NEG;
                    tmx = fmaxf(tmx, s[r]);
                }
            }
            tmx = fmaxf(tmx, __shfl_xor(tmx, 32));          // lanes l and l + 32 hold the same query: one reference
            if (__any(tmx > mref + RECENTRE)) {
                const float nm = fmaxf(mref, tmx);
                const float sc = __builtin_amdgcn_exp2f(mref - nm);
                mref = nm; lsum *= sc;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= sc;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(s[r] - mref);          // masked keys: 2^(-1e30 - m) = 0
                lsum += pr;
                s[r] = pr * (dp[r] - delta);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8_t dsf = pack_frag(s, m);
#pragma unroll
                for (int i = 0; i < NDB; ++i)
                    acc[i] = GYRE_MFMA_32x32x16(lds_frag_t(KTs, TS, i * 32 + col, D, rt * 32 + 16 * m + 4 * hi), dsf,
                                                                     acc[i], 0, 0, 0);
            }
        }
    };
    tiles[0].fetch(K, p.ldk, V, p.ldv, 0, p.Nk, D, tid);
    tiles[0].store(part(0, 0), part(0, 1), part(0, 2), nullptr, D, tid);
#pragma unroll
    for (int j = 1; j <= PF; ++j)
        if (j < nkb) tiles[j % PF].fetch(K, p.ldk, V, p.ldv, j * ROWS, p.Nk, D, tid);
    __syncthreads();
    for (int kb0 = 0; kb0 < nkb; kb0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int kb = kb0 + j;
            if (kb < nkb) {
                pass2(kb, kb & 1);
                const int nx = (j + 1) % PF;        // (compile-time after the unroll)
                if (kb + 1 < nkb) tiles[nx].store(part((kb + 1) & 1, 0), part((kb + 1) & 1, 1), part((kb + 1) & 1, 2), nullptr, D, tid);
                if (kb + 1 + PF < nkb) tiles[nx].fetch(K, p.ldk, V, p.ldv, (kb + 1 + PF) * ROWS, p.Nk, D, tid);
                __syncthreads();
            }
        }
    }
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0 && q < p.NqPad) p.lse[stat] = q < p.Nq ? mref + __builtin_amdgcn_logf(lsum) : 1e30f;
    if (q >= p.Nq) return;
    const float fin = p.beta / lsum;
    bf16_t* out = p.dq + ((size_t)b * p.Nq + q) * p.lddq + h * D;
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = i * 32 + 8 * g + 4 * hi;
            if (d < D)
                *(uint2*)(out + d) = make_uint2(pack_bf16x2(acc[i][4 * g] * fin, acc[i][4 * g + 1] * fin),
                                                pack_bf16x2(acc[i][4 * g + 2] * fin, acc[i][4 * g + 3] * fin));
        }
}

template <int NDB, int DS, int RT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ABW_WAVES_DQ(DS), ABW_WAVES_DQ(DS)))) void k_attn_bwd_dq_lds(AttnBwdParams p) { abw_dq_lds_body<NDB, DS, RT>(p); }

// ==========================================================================================================================
// Round 6: the same two kernels on an LDS-DMA ring with transposing LDS reads - head dims whose rows are an ODD number of 16-byte
// vectors up to 48 (D = 40: the 64 x 64 self-attentions of SD1.x, 85 % of config 5's attention-backward time).
//   * the streamed side arrives by global_load_lds (no registers, no staging stores): a stage = one 64-row tile pair, row-major
//     and contiguous ([64][D], 80-byte rows at D = 40: conflict-free 16-byte column reads) + the row statistics; NS = 4 stages,
//     three tiles in flight behind counted vmcnt waits, one barrier per 64 rows (28 / 20 MFMAs per wave) instead of one per 32;
//   * the second products' operands (dO^T, Q^T, K^T fragments in the permuted row order of the 32x32x16 result layout) come
//     straight out of the row-major tile through ds_read_b64_tr_b16 (each 16-lane group transposes a [4 rows][16 columns] block:
//     lane c hands in the address of row c / 4, columns 4 (c % 4) .. + 3 and receives column c of the four rows), so the
//     transposed LDS copies - sixteen 2-byte stores per lane and tile - are gone;
//   * rows past the end: the request is clamped to the last row (finite data) and the statistics make it vanish - the dQ kernel
//     leaves log-sum-exp = 1e30 on the pad rows of the 64-row statistics arrays, so P = 2^(s - 1e30) = 0 there; ragged KEY tiles keep
//     the masked form of the dQ kernel, and need nothing in the dK / dV kernel (a key past the end only feeds its own, unwritten, row);
//   * LDS is zeroed once: the 16-byte column reads of the last k-step run past column D (zero B operand, but 0 x NaN = NaN).
// ==========================================================================================================================
typedef short abw_v4s __attribute__((ext_vector_type(4)));
template <int D> struct Abw2 {
    static_assert(D % 8 == 0 && D <= 48 && ((D / 8) & 1), "odd number of 16-byte vectors per row");
    static constexpr int NV = D / 8, RS = D;                  // slots / elements per LDS row
    static constexpr int DS = (D + 15) / 16, NDB = (D + 31) / 32;
    static constexpr int TILE = 64 * NV * 16;                 // bytes of a 64-row tile = NV wave-wide 16-byte DMA instructions
    static constexpr int OFF_A = 512, OFF_B = 512 + TILE, OFF_PAD = 512 + 2 * TILE, OFF_DUMP = OFF_PAD + 256;
    static constexpr int STAGE = OFF_DUMP + 256;              // [lse 64 | delta 64][tile A][tile B][zero pad][dump of the filler requests]
    static constexpr int NS = 4;
};
__device__ __forceinline__ void abw_dma16(const void* src, unsigned dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
}
__device__ __forceinline__ void abw_dma4(const void* src, unsigned dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(src) : "memory");
}
// fragment of the transposed operand for MFMA step (rows t .. t + 3 and t + 8 .. t + 11 of this lane's 16-lane group's block)
__device__ __forceinline__ bf16x8_t abw_tr_frag(const char* lane_base, int byte_off, int rs_bytes) {
    typedef __attribute__((address_space(3))) abw_v4s* lp;
    const abw_v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lane_base + byte_off));
    const abw_v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lane_base + byte_off + 8 * rs_bytes));
    struct { abw_v4s x, y; } both = {a, b};
    return __builtin_bit_cast(bf16x8_t, both);
}

__device__ __forceinline__ float abw_keep(bf16x8_t v) { const uint4 u = __builtin_bit_cast(uint4, v); return __builtin_bit_cast(float, u.x ^ u.y ^ u.z ^ u.w); }
// ABL (timing experiments, wrong results): 1 no softmax arithmetic, 2 no second products, 4 no requests inside the loop, 8 no barrier / wait
template <int D, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_attn_bwd_dkv_dma(AttnBwdParams p) {
    using T = Abw2<D>;
    constexpr int DS = T::DS, NDB = T::NDB, RS = T::RS, NV = T::NV, NS = T::NS, STAGE = T::STAGE;
    constexpr int PPW = (2 * NV + 2 + 3) / 4;                 // requests per wave and tile (fillers make it the same for every wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    const float* LSE = p.lse + ((size_t)b * p.H + h) * p.NqPad;
    const float* DEL = p.delta + ((size_t)b * p.H + h) * p.NqPad;
    for (int i = tid; i < NS * STAGE / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    const int key = k0 + col;
    bf16x8_t kh[DS], vh[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) {
        kh[i] = ld_frag(K, p.ldk, key, p.Nk, i * 16 + 8 * hi, D);
        vh[i] = ld_frag(V, p.ldv, key, p.Nk, i * 16 + 8 * hi, D);
    }
    // this lane's part of request i of a tile: wave-instruction id = wave + 4 i; ids [0, NV) tile A (Q), [NV, 2 NV) tile B (dO):
    // slot n = 64 id' + lane of the tile -> row n / NV, vector n % NV
    int prow[PPW]; const char* pbase[PPW]; unsigned pld[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int id = wave + 4 * i;
        const bool second = id >= NV;
        const int n = (second ? id - NV : id) * 64 + lane;
        prow[i] = n / NV;
        pbase[i] = (const char*)((second ? DO : Q) + (n % NV) * 8);
        pld[i] = (unsigned)(second ? p.lddo : p.ldq) * 2u;
    }
    const int ntiles = (p.Nq + 63) / 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int t, int slot) {
        const unsigned sb = lds0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int id = wave + 4 * i;                           // (wave-uniform: scalar branches)
            if (t < ntiles && id < 2 * NV) {
                const int r = min(t * 64 + prow[i], p.Nq - 1);
                abw_dma16(pbase[i] + (size_t)((unsigned)r * pld[i]), sb + (id < NV ? T::OFF_A + id * 1024 : T::OFF_B + (id - NV) * 1024));
            } else if (t < ntiles && id < 2 * NV + 2) {
                abw_dma4((id == 2 * NV ? LSE : DEL) + t * 64 + lane, sb + (id - 2 * NV) * 256);
            } else {
                abw_dma4(LSE + lane, sb + T::OFF_DUMP);            // filler: keeps the vmcnt arithmetic the same in every wave and at the tail
            }
        }
    };
    f32x16_t dk[NDB], dv[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) { dk[i] = f32x16_t{}; dv[i] = f32x16_t{}; }
    // lane part of the transposing reads: row (c >> 2) + 4 hi, columns 16 ((lane >> 4) & 1) + 4 (c & 3), c = lane & 15
    const int tr_lane = (((lane & 15) >> 2) + 4 * hi) * RS * 2 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // a USE of the loop-invariant fragments in front of the loop: hipcc's wait-count pass puts the wait for their loads at the first
    // use, and inside the loop that is an s_waitcnt vmcnt(0) on every trip - it drains the request ring (seen in the ISA)
#pragma unroll
    for (int i = 0; i < DS; ++i) asm volatile("" : "+v"(kh[i]), "+v"(vh[i]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    for (int t = 0; t < ntiles; ++t) {
        if (!(ABL & 8)) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PPW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (!(ABL & 4)) issue(t + NS - 1, (t + NS - 1) % NS);
        const char* stg = smem + (t % NS) * STAGE;
        const float* sts = (const float*)stg;
        const bf16_t *Qs = (const bf16_t*)(stg + T::OFF_A), *DOs = (const bf16_t*)(stg + T::OFF_B);
        const char *qtr = stg + T::OFF_A + tr_lane, *dotr = stg + T::OFF_B + tr_lane;
        // (A hand-pipelined order - both halves' S / dP products first, every transposing read a phase ahead of its MFMA, the softmax
        // of one half between MFMA groups of the other - needs 221 registers, i.e. two waves per SIMD instead of three, and measured
        // 614 - 638 us against 585: the loop is bound by LDS reads - 44 KB per wave and tile for 28 MFMAs, one operand fragment per
        // MFMA is what a 32-key wave tile costs - and more resident waves overlap the three pipes better than a longer schedule.
        // Ablations, `tools/abw_abl.sh`: no softmax arithmetic -81 us, no second products -106, no transposing reads -56, no
        // statistics reads -43, no requests -76; the first-product skeleton alone 293 us.  A 64-keys-per-wave form - every fragment
        // feeding two MFMAs, 198 VGPRs + 192 AGPRs, ONE wave per SIMD - halves the LDS reads and measured 720 us against 603.)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16_t s = {}, dp = {};
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                const bf16x8_t qa = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Qs + (rt * 32 + col) * RS + i * 16 + 8 * hi));
                const bf16x8_t da = __builtin_bit_cast(bf16x8_t, *(const uint4*)(DOs + (rt * 32 + col) * RS + i * 16 + 8 * hi));
                s = GYRE_MFMA_32x32x16(qa, kh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(da, vh[i], dp, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = rt * 32 + 8 * g + 4 * hi;
                float4 l4, d4;
                if (ABL & 32) { l4 = make_float4(p.alpha, p.beta, p.alpha, p.beta); d4 = l4; }
                else { l4 = *(const float4*)(sts + ql); d4 = *(const float4*)(sts + 64 + ql); }
                const float l[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ABL & 1) { s[4 * g + r] += l[r]; dp[4 * g + r] += dl[r]; continue; }
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[4 * g + r], p.alpha, -l[r]));
                    s[4 * g + r] = pr;
                    dp[4 * g + r] = pr * (dp[4 * g + r] - dl[r]);
                }
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8_t pf = pack_frag(s, m), dsf = pack_frag(dp, m);
#pragma unroll
                for (int i = 0; i < NDB; ++i) {
                    const int off = (rt * 32 + 16 * m) * RS * 2 + i * 64;
                    const bf16x8_t dof = (ABL & 16) ? pf : abw_tr_frag(dotr, off, RS * 2);
                    const bf16x8_t qf = (ABL & 16) ? dsf : abw_tr_frag(qtr, off, RS * 2);
                    if (ABL & 2) { dv[i][m] += abw_keep(pf) + abw_keep(dof); dk[i][m] += abw_keep(dsf) + abw_keep(qf); continue; }
                    dv[i] = GYRE_MFMA_32x32x16(pf, dof, dv[i], 0, 0, 0);
                    dk[i] = GYRE_MFMA_32x32x16(dsf, qf, dk[i], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the filler requests still in flight target this workgroup's LDS
    if (k0 >= p.Nk) return;
#pragma unroll
    for (int i = 0; i < NDB; ++i) {
        const int d = i * 32 + col;
        if (d >= D) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (kk < p.Nk) {
                p.dk[((size_t)b * p.Nk + kk) * p.lddk + h * D + d] = f32_to_bf16(dk[i][r] * p.beta);
                p.dv[((size_t)b * p.Nk + kk) * p.lddv + h * D + d] = f32_to_bf16(dv[i][r]);
            }
        }
    }
}

// dQ in ONE pass over the keys (the register-staged kernel streams K twice: a log-sum-exp pass, then the products): the row
// normalisation is linear in the accumulator - dQ = (1 / l) sum_k 2^(s_k - m) (dP_k - delta) K_k with l = sum_k 2^(s_k - m) for ANY
// reference m - so the kernel keeps a reference per query, accumulates against it and divides at the end, like the forward kernel's
// output; the reference is re-centred (accumulators and l rescaled, wave-uniform branch) on the first tile and whenever a score
// exceeds it by more than 2^RECENTRE (p <= 2^20: harmless in fp32 sums and in the bf16 dS operand; 2^8 in the fp16 build) - in practice once per row.
// Lanes l and l + 32 hold the same query (different keys of the tile) and meet in the MFMA contraction, so they share the reference.
// The row log-sum-exp m + log2(l) goes to p.lse for the dK / dV kernel.  26 -> 20 MFMAs and half the exponentials per 64 keys.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_attn_bwd_dq_dma(AttnBwdParams p) {
    using T = Abw2<D>;
    constexpr int DS = T::DS, NDB = T::NDB, RS = T::RS, NV = T::NV, NS = T::NS, STAGE = T::STAGE;
    constexpr int PPW = (2 * NV + 3) / 4;
#ifdef GYRE_STORE_F16
    constexpr float RECENTRE = 8.f;      // dS = p (dP - delta) is an fp16 MFMA operand: p <= 2^8 leaves |dP - delta| < 255 before 65504
#else
    constexpr float RECENTRE = 20.f;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const bf16_t* Q = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* K = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* V = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const bf16_t* DO = p.d_o + (size_t)b * p.Nq * p.lddo + h * D;
    for (int i = tid; i < NS * STAGE / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
    const int q = q0 + col;
    const size_t stat = ((size_t)b * p.H + h) * p.NqPad + q;
    const int ntiles = (p.Nk + 63) / 64;
    bf16x8_t qh[DS], doh[DS];
#pragma unroll
    for (int i = 0; i < DS; ++i) {
        qh[i] = ld_frag(Q, p.ldq, q, p.Nq, i * 16 + 8 * hi, D);
        doh[i] = ld_frag(DO, p.lddo, q, p.Nq, i * 16 + 8 * hi, D);
    }
    float delta = q < p.Nq ? p.delta[stat] : 0.f;
    int prow[PPW]; const char* pbase[PPW]; unsigned pld[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int id = wave + 4 * i;
        const bool second = id >= NV;
        const int n = (second ? id - NV : id) * 64 + lane;
        prow[i] = n / NV;
        pbase[i] = (const char*)((second ? V : K) + (n % NV) * 8);
        pld[i] = (unsigned)(second ? p.ldv : p.ldk) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* filler = (const char*)(p.delta + ((size_t)b * p.H + h) * p.NqPad);
    auto issue = [&](int t, int slot) {
        const unsigned sb = lds0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int id = wave + 4 * i;                           // (wave-uniform: scalar branches)
            if (t < ntiles && id < 2 * NV) {
                const int r = min(t * 64 + prow[i], p.Nk - 1);
                abw_dma16(pbase[i] + (size_t)((unsigned)r * pld[i]), sb + (id < NV ? T::OFF_A + id * 1024 : T::OFF_B + (id - NV) * 1024));
            } else {
                abw_dma4(filler + lane * 4, sb + T::OFF_DUMP);     // filler: the same vmcnt arithmetic in every wave and at the tail
            }
        }
    };
    f32x16_t acc[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i) acc[i] = f32x16_t{};
    float mref = -1e30f, lsum = 0.f;
    const int tr_lane = (((lane & 15) >> 2) + 4 * hi) * RS * 2 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
    for (int i = 0; i < DS; ++i) asm volatile("" : "+v"(qh[i]), "+v"(doh[i]));      // (see k_attn_bwd_dkv_dma)
    asm volatile("" : "+v"(delta));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t, t);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(t + NS - 1, (t + NS - 1) % NS);
        const char* stg = smem + (t % NS) * STAGE;
        const bf16_t *Ks = (const bf16_t*)(stg + T::OFF_A), *Vs = (const bf16_t*)(stg + T::OFF_B);
        const char* ktr = stg + T::OFF_A + tr_lane;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16_t s = {}, dp = {};
#pragma unroll
            for (int i = 0; i < DS; ++i) {
                const bf16x8_t ka = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Ks + (rt * 32 + col) * RS + i * 16 + 8 * hi));
                const bf16x8_t va = __builtin_bit_cast(bf16x8_t, *(const uint4*)(Vs + (rt * 32 + col) * RS + i * 16 + 8 * hi));
                s = GYRE_MFMA_32x32x16(ka, qh[i], s, 0, 0, 0);
                dp = GYRE_MFMA_32x32x16(va, doh[i], dp, 0, 0, 0);
            }
            const bool whole = t * 64 + rt * 32 + 32 <= p.Nk;       // (wave-uniform)
            float tmx = -1e30f;
            if (whole) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] *= p.alpha; tmx = fmaxf(tmx, s[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[r] = key < p.Nk ? s[r] * p.alpha : -1e30f;
                    tmx = fmaxf(tmx, s[r]);
                }
            }
            tmx = fmaxf(tmx, __shfl_xor(tmx, 32));
            if (__any(tmx > mref + RECENTRE)) {
                const float nm = fmaxf(mref, tmx);                  // (a lane whose row is fine moves too: any reference is exact)
                const float sc = __builtin_amdgcn_exp2f(mref - nm);
                mref = nm; lsum *= sc;
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= sc;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(s[r] - mref);          // masked keys: 2^(-1e30 - m) = 0
                lsum += pr;
                s[r] = pr * (dp[r] - delta);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8_t dsf = pack_frag(s, m);
#pragma unroll
                for (int i = 0; i < NDB; ++i)
                    acc[i] = GYRE_MFMA_32x32x16(abw_tr_frag(ktr, (rt * 32 + 16 * m) * RS * 2 + i * 64, RS * 2), dsf, acc[i], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the filler requests still in flight target this workgroup's LDS
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0 && q < p.NqPad) p.lse[stat] = q < p.Nq ? mref + __builtin_amdgcn_logf(lsum) : 1e30f;      // pad rows: P = 0 in the dK / dV kernel
    if (q >= p.Nq) return;
    const float fin = p.beta / lsum;
    bf16_t* out = p.dq + ((size_t)b * p.Nq + q) * p.lddq + h * D;
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = i * 32 + 8 * g + 4 * hi;
            if (d < D)
                *(uint2*)(out + d) = make_uint2(pack_bf16x2(acc[i][4 * g] * fin, acc[i][4 * g + 1] * fin),
                                                pack_bf16x2(acc[i][4 * g + 2] * fin, acc[i][4 * g + 3] * fin));
        }
}

bool attn_bwd_needs_transposes(int D) { return D > 160; }
size_t attn_bwd_stats_bytes(int B, int H, int Nq) {
    const size_t npad = (size_t)(Nq + 63) / 64 * 64;     // 64-row tiles of the LDS-DMA kernels
    return 2 * (((size_t)B * H * npad * 4 + 255) & ~(size_t)255);
}
int launch_attention_bwd(hipStream_t st, AttnBwdParams p) {
    if (p.D % 8 || p.ldq % 8 || p.ldk % 8 || p.ldv % 8 || p.lddo % 8 || p.ldo % 8 || p.lddq % 4 || p.ldkt % 4 || p.ldqt % 4)
        GYRE_FAIL(-1, "attention_bwd: head dim and row strides must be multiples of 8 elements");
    if (p.B < 1 || p.H < 1 || p.Nq < 1 || p.Nk < 1) GYRE_FAIL(-1, "attention_bwd: empty problem");
    const bool lds_path = !attn_bwd_needs_transposes(p.D);
    if (!lds_path && (!p.kt || p.ldkt < (p.Nk + 31) / 32 * 32 || (p.dk && (!p.qt || !p.d_ot || p.ldqt < (p.Nq + 31) / 32 * 32))))
        GYRE_FAIL(-1, "attention_bwd: transposed operands need token strides padded to a multiple of 32");
    p.NqPad = (p.Nq + 63) / 64 * 64;
    const size_t half = ((size_t)p.B * p.H * p.NqPad * 4 + 255) & ~(size_t)255;
    p.lse = (float*)p.stats; p.delta = (float*)((char*)p.stats + half);
    const float sc = 1.0f / sqrtf((float)p.D);
    // K prescaled by log2(e) / sqrt(D) (UNet to_k weights): logits are already base-2; d logit / d (Q K'^T) = ln 2
    p.alpha = p.k_prescaled ? 1.0f : sc * 1.4426950408889634f;
    p.beta = p.k_prescaled ? 0.6931471805599453f : sc;
    GYRE_HIP_CHECK(hipMemsetAsync(p.stats, 0, 2 * half, st));
    // S, dP, dQ (and dK, dV) products; the recomputation of S / dP in the second kernel and the statistics pass are not counted
    GyreProfScope prof_(KC_ATTN_BWD, st, (p.dk ? 5.0 : 3.0) * 2.0 * p.B * p.H * (double)p.Nq * p.Nk * p.D,
                        2.0 * ((double)p.B * p.Nq * p.H * p.D * 4 + (double)p.B * p.Nk * p.H * p.D * (p.dk ? 4 : 2)));
    const size_t nd = (size_t)p.B * p.Nq * p.H;
    hipLaunchKernelGGL(k_attn_bwd_delta, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, p);
    GYRE_LAUNCH_CHECK();
    static const bool no_dma = getenv("GYRE_ABW_NO_DMA") != nullptr;      // tuning / A-B: the register-staged kernels for D = 40 too
    if (p.D == 40 && !no_dma) {
        const size_t lds = (size_t)Abw2<40>::NS * Abw2<40>::STAGE;
        const dim3 gq((unsigned)((p.Nq + 127) / 128), p.H, p.B), gk((unsigned)((p.Nk + 127) / 128), p.H, p.B);
        hipLaunchKernelGGL((k_attn_bwd_dq_dma<40>), gq, dim3(256), lds, st, p);
        GYRE_LAUNCH_CHECK();
        if (!p.dk) return 0;
#ifdef GYRE_ABW_ABLATIONS       // timing experiments (tools/abw_abl.sh; build the library with -DGYRE_ABW_ABLATIONS): wrong results by design
        static const int abl = getenv("GYRE_ABW_ABL") ? atoi(getenv("GYRE_ABW_ABL")) : 0;
        switch (abl) {
            case 1: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 1>), gk, dim3(256), lds, st, p); break;
            case 2: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 2>), gk, dim3(256), lds, st, p); break;
            case 3: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 3>), gk, dim3(256), lds, st, p); break;
            case 4: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 4>), gk, dim3(256), lds, st, p); break;
            case 12: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 12>), gk, dim3(256), lds, st, p); break;
            case 16: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 16>), gk, dim3(256), lds, st, p); break;
            case 32: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 32>), gk, dim3(256), lds, st, p); break;
            case 48: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 48>), gk, dim3(256), lds, st, p); break;
            case 63: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 63>), gk, dim3(256), lds, st, p); break;
            case 15: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40, 15>), gk, dim3(256), lds, st, p); break;
            default: hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40>), gk, dim3(256), lds, st, p); break;
        }
#else
        hipLaunchKernelGGL((k_attn_bwd_dkv_dma<40>), gk, dim3(256), lds, st, p);
#endif
        GYRE_LAUNCH_CHECK();
        return 0;
    }
    if (lds_path) {
        const int sel = p.D <= 48 ? 0 : (p.D <= 64 ? 1 : (p.D <= 80 ? 4 : (p.D <= 96 ? 2 : 3)));      // 4: D = 80 with five k-steps (the 32 x 32 level)
        // 32-row LDS tiles: 64-row tiles (RT = 2) were measured slower (2.71 vs 2.14 ms at N = 4096, D = 40: fewer resident
        // workgroups outweigh the halved barrier count)
        const size_t lds = 2 * attn_bwd_stage_bytes(p.D, 1);
        const dim3 gq((unsigned)((p.Nq + 127) / 128), p.H, p.B), gk((unsigned)((p.Nk + 127) / 128), p.H, p.B);
#define GYRE_ABW_GO(KERN, GRID)                                                                                   \
        do {                                                                                                          \
            auto kern = KERN;                                                                                         \
            static std::atomic<unsigned long long> attr_done{0};                                                      \
            if (gyre_lds_attr_needed(attr_done))                                                                      \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
            hipLaunchKernelGGL(kern, GRID, dim3(256), lds, st, p);                                                    \
            GYRE_LAUNCH_CHECK();                                                                                      \
        } while (0)
        switch (sel) {
            case 0: GYRE_ABW_GO((k_attn_bwd_dq_lds<2, 3, 1>), gq); break;
            case 1: GYRE_ABW_GO((k_attn_bwd_dq_lds<2, 4, 1>), gq); break;
            case 2: GYRE_ABW_GO((k_attn_bwd_dq_lds<3, 6, 1>), gq); break;
            case 4: GYRE_ABW_GO((k_attn_bwd_dq_lds<3, 5, 1>), gq); break;
            default: GYRE_ABW_GO((k_attn_bwd_dq_lds<5, 10, 1>), gq); break;
        }
        if (!p.dk) return 0;
        switch (sel) {
            case 0: GYRE_ABW_GO((k_attn_bwd_dkv_lds<2, 3, 1>), gk); break;
            case 1: GYRE_ABW_GO((k_attn_bwd_dkv_lds<2, 4, 1>), gk); break;
            case 2: GYRE_ABW_GO((k_attn_bwd_dkv_lds<3, 6, 1>), gk); break;
            case 4: GYRE_ABW_GO((k_attn_bwd_dkv_lds<3, 5, 1>), gk); break;
            default: GYRE_ABW_GO((k_attn_bwd_dkv_lds<5, 10, 1>), gk); break;
        }
#undef GYRE_ABW_GO
        return 0;
    }
    const int ndb_total = (p.D + 31) / 32;
    const int sel = p.D <= 48 ? 0 : (p.D <= 64 ? 1 : (p.D <= 96 ? 2 : (p.D <= 160 ? 3 : 4)));
    const int ndb = sel <= 1 ? 2 : (sel == 2 ? 3 : (sel == 3 ? 5 : 4));
    const int dchunks = (ndb_total + ndb - 1) / ndb;
    const dim3 gq((unsigned)((p.Nq + 127) / 128 * dchunks), p.H, p.B), gk((unsigned)((p.Nk + 127) / 128 * dchunks), p.H, p.B);
    switch (sel) {
        case 0: hipLaunchKernelGGL((k_attn_bwd_dq<2, 3>), gq, dim3(256), 0, st, p, dchunks); break;
        case 1: hipLaunchKernelGGL((k_attn_bwd_dq<2, 4>), gq, dim3(256), 0, st, p, dchunks); break;
        case 2: hipLaunchKernelGGL((k_attn_bwd_dq<3, 6>), gq, dim3(256), 0, st, p, dchunks); break;
        case 3: hipLaunchKernelGGL((k_attn_bwd_dq<5, 10>), gq, dim3(256), 0, st, p, dchunks); break;
        default: hipLaunchKernelGGL((k_attn_bwd_dq<4, 0>), gq, dim3(256), 0, st, p, dchunks); break;
    }
    GYRE_LAUNCH_CHECK();
    if (!p.dk) return 0;
    switch (sel) {
        case 0: hipLaunchKernelGGL((k_attn_bwd_dkv<2, 3>), gk, dim3(256), 0, st, p, dchunks); break;
        case 1: hipLaunchKernelGGL((k_attn_bwd_dkv<2, 4>), gk, dim3(256), 0, st, p, dchunks); break;
        case 2: hipLaunchKernelGGL((k_attn_bwd_dkv<3, 6>), gk, dim3(256), 0, st, p, dchunks); break;
        case 3: hipLaunchKernelGGL((k_attn_bwd_dkv<5, 10>), gk, dim3(256), 0, st, p, dchunks); break;
        default: hipLaunchKernelGGL((k_attn_bwd_dkv<4, 0>), gk, dim3(256), 0, st, p, dchunks); break;
    }
    GYRE_LAUNCH_CHECK();
    return 0;
}
