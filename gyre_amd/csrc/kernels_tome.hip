// Token merging (ToMe) for self-attention keys / values.
//
// Reference: nonfree/tome_unet.py:138-182 (ToMeCrossAttention.forward) - after the K and V projections of a SELF
// attention, `merge, _ = bipartite_soft_matching(key, r, ...)`, `key = merge_wavg(merge, key)`,
// `value = merge_wavg(merge, value)`: the r most redundant keys are averaged into their best match, attention then
// runs against N - r keys (queries are untouched).  The algorithm itself lives in facebookresearch/ToMe
// (git submodule, absent from the reference tree -> parity unpinned); restated from the published method
// (Bolya et al., "Token Merging: Your ViT But Faster", 2023, tome/merge.py):
//
//   metric = key / |key|                                  (full C-wide key vector, all heads)
//   a, b   = metric[::2], metric[1::2]                    (alternating bipartition)
//   scores = a @ b^T ; node_max, node_idx = scores.max(-1) (best b partner of every a token)
//   edge   = argsort(node_max, descending)                (a tokens by similarity of their best edge)
//   src = edge[:r] are merged into dst = node_idx[src]; unm = edge[r:] stay
//   merge(x) = cat([ x_a[unm], scatter_reduce(x_b, dst, x_a[src], "sum") / counts ])       (merge_wavg, sizes 1)
//
// Deterministic tie rules (the published code leaves them to torch.max / torch.argsort): the FIRST maximal b index wins,
// equal scores rank by ascending a index.  The similarity matrix is a bf16 MFMA GEMM (fp32 accumulate) whose output is
// rounded to bf16 - the oracle (oracle/tome_ref.py) applies the same two roundings so that selections can be compared.
//
// Kernels (all HBM / latency bound, tiny next to the attention they shorten):
//   k_tome_normalize_split  one wave per token: 1/|k| (fp32), normalised bf16 row to the a or b half
//   (batched k_gemm)        scores[b] = a[b] b[b]^T  ->  bf16 [B][N/2][N/2]
//   k_tome_rowmax           one wave per a token: max / first argmax over its score row
//   k_tome_sort             one workgroup per sample: bitonic sort of (score desc, index asc) keys in LDS
//   k_tome_merge_rows       one wave per output row: copy (unmerged a) or mean of a b token and its merged a tokens
//                           (sources found by scanning the r-entry destination list, accumulated in rank order)
//   k_tome_transpose        merged V rows -> V^T [B][C][N' padded to 8], the layout the attention kernel streams
#include "kernels.h"

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
int tome_effective_r(int N, int r) { const int mx = N / 2; return r < 0 ? 0 : (r > mx ? mx : r); }

__global__ __launch_bounds__(256) void k_tome_normalize_split(const bf16_t* k, int ldk, int B, int N, int C, bf16_t* a, bf16_t* b) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * N) return;
    const int bi = row / N, n = row - bi * N;
    const bf16_t* src = k + ((size_t)bi * N + n) * ldk;
    const int nv = C / 8;
    float ss = 0.f;
    for (int v = lane; v < nv; v += 64) {
        float f[8];
        unpack8(*(const uint4*)(src + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = rsqrtf(fmaxf(ss, 1e-30f));
    const int half = N / 2;
    if ((n >> 1) >= half) return;                         // odd N: the last token is not part of the bipartition
    bf16_t* dst = ((n & 1) ? b : a) + ((size_t)bi * half + (n >> 1)) * C;
    for (int v = lane; v < nv; v += 64) {
        float f[8];
        unpack8(*(const uint4*)(src + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= inv;
        *(uint4*)(dst + v * 8) = pack8(f);
    }
}

// scores [B][half][lds] bf16 -> node_max [B][half] (fp32 value of the bf16 maximum), node_idx [B][half] (first argmax)
__global__ __launch_bounds__(256) void k_tome_rowmax(const bf16_t* scores, int B, int half, int lds, float* node_max, int* node_idx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * half) return;
    const bf16_t* s = scores + (size_t)row * lds;
    float best = -3.0e38f;
    int bi = 0x7fffffff;
    for (int j0 = lane * 8; j0 < half; j0 += 512) {
        if (j0 + 8 <= half) {
            float f[8];
            unpack8(*(const uint4*)(s + j0), f);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (f[e] > best) { best = f[e]; bi = j0 + e; }
        } else {
            for (int j = j0; j < half; ++j) {
                const float f = bf16_to_f32(s[j]);
                if (f > best) { best = f; bi = j; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { node_max[row] = best; node_idx[row] = bi; }
}

// ascending sort of 64-bit keys (inverted order-preserving score bits << 32 | a index): descending score, ascending index
__global__ __launch_bounds__(1024) void k_tome_sort(const float* node_max, const int* node_idx, int half, int P, int r,
                                                     int* order, int* dstlist) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < P; i += nt) {
        unsigned long long key = ~0ull;
        if (i < half) {
            const unsigned u = __float_as_uint(node_max[(size_t)b * half + i]);
            const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // monotone in the float value
            key = ((unsigned long long)(~ord) << 32) | (unsigned)i;
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < half; i += nt) {
        const int ai = (int)(unsigned)(keys[i] & 0xffffffffu);
        order[(size_t)b * half + i] = ai;
        if (i < r) dstlist[(size_t)b * half + i] = node_idx[(size_t)b * half + ai];
    }
}

// x [B][N][ldx] -> y [B][N - r][C]:  rows [0, half - r) = unmerged a tokens (rank order), then the N - half b-side tokens
// (b tokens averaged with the a tokens merged into them; an odd trailing token is appended unchanged)
__global__ __launch_bounds__(256) void k_tome_merge_rows(const bf16_t* x, int ldx, int B, int N, int C, int r, const int* order,
                                                         const int* dstlist, bf16_t* y) {
    const int half = N / 2, nu = half - r, nout = N - r;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * nout) return;
    const int b = row / nout, t = row - b * nout;
    const bf16_t* xb = x + (size_t)b * N * ldx;
    bf16_t* dst = y + ((size_t)b * nout + t) * C;
    const int nv = C / 8;
    if (t < nu || t >= nu + half) {
        const int tok = t < nu ? 2 * order[(size_t)b * half + r + t] : N - 1;
        for (int v = lane; v < nv; v += 64) *(uint4*)(dst + v * 8) = *(const uint4*)(xb + (size_t)tok * ldx + v * 8);
        return;
    }
    const int j = t - nu;
    constexpr int MAXV = 3;                               // C <= 1536
    float acc[MAXV][8];
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
        const int v = lane + 64 * q;
        if (v < nv) unpack8(*(const uint4*)(xb + (size_t)(2 * j + 1) * ldx + v * 8), acc[q]);
    }
    int cnt = 1;
    const int* dl = dstlist + (size_t)b * half;
    const int* od = order + (size_t)b * half;
    for (int k0 = 0; k0 < r; k0 += 64) {
        const int k = k0 + lane;
        const bool hit = k < r && dl[k] == j;
        unsigned long long m = __ballot(hit);
        while (m) {                                       // ascending rank: fixed accumulation order
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int tok = 2 * od[k0 + bit];
#pragma unroll
            for (int q = 0; q < MAXV; ++q) {
                const int v = lane + 64 * q;
                if (v < nv) {
                    float f[8];
                    unpack8(*(const uint4*)(xb + (size_t)tok * ldx + v * 8), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[q][e] += f[e];
                }
            }
            ++cnt;
        }
    }
    const float inv = 1.0f / (float)cnt;
#pragma unroll
    for (int q = 0; q < MAXV; ++q) {
        const int v = lane + 64 * q;
        if (v < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] *= inv;
            *(uint4*)(dst + v * 8) = pack8(acc[q]);
        }
    }
}

// v [B][T][C] -> vt [B][C][ldt] (columns T .. ldt-1 zero)
__global__ __launch_bounds__(256) void k_tome_transpose(const bf16_t* v, int B, int T, int C, bf16_t* vt, int ldt) {
    __shared__ bf16_t tile[64][66];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int t = t0 + i, c = c0 + tx;
        tile[i][tx] = (t < T && c < C) ? v[((size_t)b * T + t) * C + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < ldt) vt[((size_t)b * C + c) * ldt + t] = tile[tx][i];
    }
}

// workspace layout: a | b (normalised halves) | scores | node_max | node_idx | order | dstlist | merged V rows
size_t tome_workspace_bytes(int B, int N, int C) {
    const size_t half = N / 2, lds = (half + 7) / 8 * 8;        // score rows padded to 16 bytes
    return 2 * al256((size_t)B * half * C * 2) + al256((size_t)B * half * lds * 2) + 4 * al256((size_t)B * half * 4) +
           al256((size_t)B * N * C * 2);
}

int launch_tome_merge(hipStream_t st, const TomeParams& p) {
    const int half = p.N / 2, r = tome_effective_r(p.N, p.r);
    if (p.B <= 0 || p.N < 2 || p.C <= 0 || p.C % 8 || p.C > 1536 || p.ldk % 8 || p.ldv % 8 || r <= 0)
        GYRE_FAIL(-1, "tome: needs N >= 2, C a multiple of 8 (<= 1536), 16-byte aligned rows and r > 0");
    if (half > 16384) GYRE_FAIL(-6, "tome: more than 32768 tokens per sample are not supported (sort runs in one workgroup's LDS)");
    if (!p.ws || p.ws_bytes < tome_workspace_bytes(p.B, p.N, p.C)) GYRE_FAIL(-4, "tome: workspace too small");
    if (p.ldvt < p.N - r) GYRE_FAIL(-1, "tome: ldvt smaller than the merged token count");
    char* w = (char*)p.ws;
    bf16_t* a = (bf16_t*)w; w += al256((size_t)p.B * half * p.C * 2);
    bf16_t* b = (bf16_t*)w; w += al256((size_t)p.B * half * p.C * 2);
    const int lds = (half + 7) / 8 * 8;
    bf16_t* scores = (bf16_t*)w; w += al256((size_t)p.B * half * lds * 2);
    float* node_max = (float*)w; w += al256((size_t)p.B * half * 4);
    int* node_idx = (int*)w; w += al256((size_t)p.B * half * 4);
    int* order = (int*)w; w += al256((size_t)p.B * half * 4);
    int* dstlist = (int*)w; w += al256((size_t)p.B * half * 4);
    bf16_t* vrows = p.vrows_out ? p.vrows_out : (bf16_t*)w;
    {
        GyreProfScope prof_(KC_OTHER, st, 0, (double)p.B * p.N * p.C * 4.0);
        hipLaunchKernelGGL(k_tome_normalize_split, dim3((p.B * p.N + 3) / 4), dim3(256), 0, st, p.k, p.ldk, p.B, p.N, p.C, a, b);
        GYRE_LAUNCH_CHECK();
    }
    GemmParams g;
    g.A = a; g.lda = p.C; g.mode = GEMM_LINEAR; g.W = b; g.K = p.C; g.N = half; g.M = half;
    g.out = scores; g.ldc = lds; g.out_mode = OUT_BF16;
    g.batch = p.B; g.bsA = (size_t)half * p.C; g.bsW = (size_t)half * p.C; g.bsC = (size_t)half * lds;
    if (half % 4) GYRE_FAIL(-1, "tome: N / 2 must be a multiple of 4");
    int rc = launch_gemm(st, g);
    if (rc) return rc;
    GyreProfScope prof_(KC_OTHER, st, 0, (double)p.B * half * half * 2.0 + (double)p.B * p.N * p.C * 6.0);
    hipLaunchKernelGGL(k_tome_rowmax, dim3((p.B * half + 3) / 4), dim3(256), 0, st, scores, p.B, half, lds, node_max, node_idx);
    GYRE_LAUNCH_CHECK();
    int P = 1;
    while (P < half) P <<= 1;
    const int threads = P / 2 < 64 ? 64 : (P / 2 > 1024 ? 1024 : P / 2);
    {
        auto kern = k_tome_sort;
        static std::atomic<unsigned long long> attr_done{0};
        if (gyre_lds_attr_needed(attr_done))
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
        hipLaunchKernelGGL(kern, dim3(p.B), dim3(threads), (size_t)P * 8, st, node_max, node_idx, half, P, r, order, dstlist);
        GYRE_LAUNCH_CHECK();
    }
    const int nout = p.N - r;
    hipLaunchKernelGGL(k_tome_merge_rows, dim3((p.B * nout + 3) / 4), dim3(256), 0, st, p.k, p.ldk, p.B, p.N, p.C, r, order, dstlist, p.k_out);
    GYRE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tome_merge_rows, dim3((p.B * nout + 3) / 4), dim3(256), 0, st, p.v, p.ldv, p.B, p.N, p.C, r, order, dstlist, vrows);
    GYRE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tome_transpose, dim3((p.ldvt + 63) / 64, (p.C + 63) / 64, p.B), dim3(256), 0, st, vrows, p.B, nout, p.C, p.vt_out, p.ldvt);
    GYRE_LAUNCH_CHECK();
    if (p.order_out) (void)hipMemcpyAsync(p.order_out, order, (size_t)p.B * half * 4, hipMemcpyDeviceToDevice, st);
    if (p.node_idx_out) (void)hipMemcpyAsync(p.node_idx_out, node_idx, (size_t)p.B * half * 4, hipMemcpyDeviceToDevice, st);
    if (p.dstlist_out) (void)hipMemcpyAsync(p.dstlist_out, dstlist, (size_t)p.B * half * 4, hipMemcpyDeviceToDevice, st);
    return 0;
}

// ---- adjoint of k_tome_merge_rows (input-gradient pass of the CLIP-guided mode; the matching itself carries no gradient) ----
__global__ __launch_bounds__(256) void k_tome_inv_rank(const int* order, int B, int half, int* inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half;
    inv[(size_t)b * half + order[i]] = i - b * half;
}
__global__ __launch_bounds__(256) void k_tome_unmerge_rows(const bf16_t* dy, int B, int N, int C, int r, const int* inv,
                                                           const int* dstlist, bf16_t* dx, int ldx) {
    const int half = N / 2, nu = half - r, nout = N - r;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * N) return;
    const int b = row / N, n = row - b * N;
    const int* dl = dstlist + (size_t)b * half;
    int src, j = -1;                                      // source row of dy; j >= 0: a merged b-side row (divide by its count)
    if (n >= 2 * half) src = nout - 1;                    // unpaired trailing token
    else if (n & 1) { j = n >> 1; src = nu + j; }
    else {
        const int rank = inv[(size_t)b * half + (n >> 1)];
        if (rank >= r) src = rank - r;
        else { j = dl[rank]; src = nu + j; }
    }
    float w = 1.0f;
    if (j >= 0) {
        int cnt = 1;
        for (int k0 = 0; k0 < r; k0 += 64) {
            const int k = k0 + lane;
            cnt += __popcll(__ballot(k < r && dl[k] == j));
        }
        w = 1.0f / (float)cnt;
    }
    const bf16_t* s = dy + ((size_t)b * nout + src) * C;
    bf16_t* d = dx + ((size_t)b * N + n) * ldx;
    for (int v = lane; v < C / 8; v += 64) {
        float f[8];
        unpack8(*(const uint4*)(s + v * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= w;
        *(uint4*)(d + v * 8) = pack8(f);
    }
}
int launch_tome_unmerge(hipStream_t st, const bf16_t* dy, int B, int N, int C, int r, const int* order, const int* dstlist,
                        int* inv, bf16_t* dx, int ldx) {
    const int half = N / 2;
    if (C % 8 || ldx % 8 || r < 1 || r > half) GYRE_FAIL(-1, "tome_unmerge: C, ldx multiples of 8 and 1 <= r <= N / 2");
    hipLaunchKernelGGL(k_tome_inv_rank, dim3((B * half + 255) / 256), dim3(256), 0, st, order, B, half, inv);
    GYRE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_tome_unmerge_rows, dim3((B * N + 3) / 4), dim3(256), 0, st, dy, B, N, C, r, inv, dstlist, dx, ldx);
    GYRE_LAUNCH_CHECK();
    return 0;
}
