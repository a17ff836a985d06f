// Flash-style attention forward (no mask, scale D^-1/2) on bf16 MFMA, fp32 online softmax.
//
//   O[b,q,h*D+d] = sum_kv softmax_kv( Q[b,q,h,:] . K[b,kv,h,:] * D^-1/2 ) V[b,kv,h,d]
//
// Same reshape/scale contract as the reference's xformers path
// (gyre/pipeline/models/memory_efficient_cross_attention.py:32-60: [B,N,h*d] -> [B*h,N,d], no mask,
// scale d^-1/2), which reaches xformers.ops.memory_efficient_attention [3P].
//
// CDNA4 mapping:
//   * one workgroup = 4 waves = 64*QI query rows of one (batch, head); K/V^T tiles of 64 keys staged
//     in LDS; Q fragments live in registers for the whole kernel.
//   * both products are issued "swapped" so the reduction axis of the softmax is lane-local:
//       S^T[kv][q] = mfma(A = K rows, B = Q rows)      lane: q = lane&15, 4 keys per fragment
//       O^T[d][q]  = mfma(A = V^T rows, B = P)          lane: q = lane&15, 4 output dims per fragment
//     so row max / row sum are 15 in-lane ops + 2 wave64 shuffles (xor 16, 32), the rescale factor
//     is lane-local for O^T, and P never leaves registers: the K rows of fragment pair (2s, 2s+1) are
//     permuted (krow below) so that the 8 P values a lane holds for k-step s are keys 32s+8g..+8 -
//     exactly the contiguous 16 bytes of a V^T row that ds_read_b128 delivers as the A operand.
//   * V is consumed transposed ([B][h*D][token], token contiguous); the V-projection GEMM writes it
//     in that layout directly (OUT_BF16_T epilogue), so no in-kernel transpose is needed.
//   * head dims 40/80/160 are zero-padded in LDS only (QK^T contraction to a multiple of 32, PV
//     output to a multiple of 16); global traffic is the unpadded Q/K/V/O.
#include "kernels.h"

template <int D, int QI>
__global__ __launch_bounds__(256) void k_attn(AttnParams p) {
    constexpr int DP = (D + 31) / 32 * 32;   // QK^T contraction length (padded)
    constexpr int KS = DP / 32;              // k-steps for QK^T
    constexpr int DO = (D + 15) / 16;        // output d fragments
    constexpr int KROW = DP * 2 + 16;        // K tile row stride in bytes (padded)
    constexpr int VROW = 64 * 2 + 16;        // V^T tile row stride in bytes
    constexpr int KVEC = DP / 8;             // 16-byte vectors per K row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_lds = smem;                      // [64][KROW]
    char* v_lds = smem + 64 * KROW;          // [DO*16][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int q0 = blockIdx.x * (64 * QI) + wave * (16 * QI);

    const bf16_t* qb = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* kb = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* vb = p.vt + ((size_t)b * p.H * D + (size_t)h * D) * p.ldvt;

    // ---- Q fragments (MFMA B operand: column q = fr, slots d = 32*ks + 8*fq + j) --------------
    bf16x8_t qf[QI][KS];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const int q = q0 + qi * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + fq * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.Nq && d < D) v = *(const uint4*)(qb + (size_t)q * p.ldq + d);
            qf[qi][ks] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    f32x4_t o[QI][DO];
    float m_run[QI], l_run[QI];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        m_run[qi] = -1e30f; l_run[qi] = 0.f;
#pragma unroll
        for (int di = 0; di < DO; ++di) o[qi][di] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const float sc = rsqrtf((float)D) * 1.4426950408889634f;  // scale * log2(e)

    for (int kv0 = 0; kv0 < p.Nk; kv0 += 64) {
        __syncthreads();  // everyone is done reading the previous tile
        // ---- stage K tile [64][DP] (zero padded) --------------------------------------------
        for (int v = tid; v < 64 * KVEC; v += 256) {
            const int row = v / KVEC, dv = v - row * KVEC;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (kv0 + row < p.Nk && dv * 8 < D) val = *(const uint4*)(kb + (size_t)(kv0 + row) * p.ldk + dv * 8);
            *(uint4*)(k_lds + row * KROW + dv * 16) = val;
        }
        // ---- stage V^T tile [DO*16][64]; keys >= Nk and dims >= D are zeroed ------------------
        for (int v = tid; v < DO * 16 * 8; v += 256) {
            const int row = v >> 3, c = v & 7;
            const int kv = kv0 + c * 8;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (row < D && kv < p.Nk) {
                val = *(const uint4*)(vb + (size_t)row * p.ldvt + kv);
                const int nvalid = p.Nk - kv;  // >= 1
                if (nvalid < 8) {
                    uint32_t w[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (2 * e >= nvalid) w[e] = 0;
                        else if (2 * e + 1 >= nvalid) w[e] &= 0xffffu;
                    }
                    val = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            *(uint4*)(v_lds + row * VROW + c * 16) = val;
        }
        __syncthreads();

        // ---- S^T = K Q^T ---------------------------------------------------------------------
        f32x4_t s[QI][4];
#pragma unroll
        for (int qi = 0; qi < QI; ++qi)
#pragma unroll
            for (int ki = 0; ki < 4; ++ki) s[qi][ki] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
            // permuted key row: MFMA row i of fragment ki <- key 32*(ki>>1) + 8*(i>>2) + 4*(ki&1) + (i&3)
            const int krow = 32 * (ki >> 1) + 8 * (fr >> 2) + 4 * (ki & 1) + (fr & 3);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t kf = __builtin_bit_cast(bf16x8_t, *(const uint4*)(k_lds + krow * KROW + (ks * 4 + fq) * 16));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    s[qi][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], s[qi][ki], 0, 0, 0);
            }
        }
        // lane now holds, for q = fr: s[qi][ki][r] = key 32*(ki>>1) + 8*fq + 4*(ki&1) + r
        const bool tail = kv0 + 64 > p.Nk;
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            float mx = -1e30f;
#pragma unroll
            for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = s[qi][ki][r] * sc;
                    if (tail) {
                        const int kvl = 32 * (ki >> 1) + 8 * fq + 4 * (ki & 1) + r;
                        if (kv0 + kvl >= p.Nk) t = -1e30f;
                    }
                    s[qi][ki][r] = t;
                    mx = fmaxf(mx, t);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qi], mx);
            const float alpha = exp2f(m_run[qi] - m_new);
            m_run[qi] = m_new;
            float ls = 0.f;
#pragma unroll
            for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = exp2f(s[qi][ki][r] - m_new);
                    s[qi][ki][r] = pv;
                    ls += pv;
                }
            l_run[qi] = l_run[qi] * alpha + ls;  // per-lane partial row sum; reduced over fq at the end
#pragma unroll
            for (int di = 0; di < DO; ++di) {
                o[qi][di][0] *= alpha; o[qi][di][1] *= alpha; o[qi][di][2] *= alpha; o[qi][di][3] *= alpha;
            }
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            bf16x8_t pf[QI];
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                uint4 w;
                w.x = pack_bf16x2(s[qi][2 * ks2][0], s[qi][2 * ks2][1]);
                w.y = pack_bf16x2(s[qi][2 * ks2][2], s[qi][2 * ks2][3]);
                w.z = pack_bf16x2(s[qi][2 * ks2 + 1][0], s[qi][2 * ks2 + 1][1]);
                w.w = pack_bf16x2(s[qi][2 * ks2 + 1][2], s[qi][2 * ks2 + 1][3]);
                pf[qi] = __builtin_bit_cast(bf16x8_t, w);
            }
#pragma unroll
            for (int di = 0; di < DO; ++di) {
                bf16x8_t vf = __builtin_bit_cast(
                    bf16x8_t, *(const uint4*)(v_lds + (di * 16 + fr) * VROW + (ks2 * 32 + fq * 8) * 2));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    o[qi][di] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qi], o[qi][di], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: O[q][h*D + 16*di + 4*fq + r] = o / l -------------------------------------------
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + qi * 16 + fr;
        if (q >= p.Nq) continue;
        bf16_t* orow = p.o + ((size_t)b * p.Nq + q) * p.ldo + h * D;
#pragma unroll
        for (int di = 0; di < DO; ++di) {
            const int d = di * 16 + 4 * fq;
            if (d < D) {
                uint2 pk = make_uint2(pack_bf16x2(o[qi][di][0] * inv, o[qi][di][1] * inv),
                                      pack_bf16x2(o[qi][di][2] * inv, o[qi][di][3] * inv));
                *(uint2*)(orow + d) = pk;
            }
        }
    }
}

template <int D, int QI>
static int launch_attn_t(hipStream_t st, const AttnParams& p) {
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int DO = (D + 15) / 16;
    const size_t lds = (size_t)64 * (DP * 2 + 16) + (size_t)DO * 16 * (64 * 2 + 16);
    auto kern = k_attn<D, QI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid((p.Nq + 64 * QI - 1) / (64 * QI), p.B * p.H);
    GyreProfScope prof_(KC_ATTN, st, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * D,
                        2.0 * p.B * p.H * D * (2.0 * p.Nq + 2.0 * p.Nk));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    GYRE_LAUNCH_CHECK();
    return 0;
}

int launch_attention(hipStream_t st, const AttnParams& p) {
    if (p.D % 8) GYRE_FAIL(-1, "attention: head dim must be a multiple of 8");
    if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4) GYRE_FAIL(-1, "attention: strides must be multiples of 8");
    if (p.ldvt < (p.Nk + 7) / 8 * 8) GYRE_FAIL(-1, "attention: ldvt must cover Nk rounded up to 8");
    if (p.Nk < 1 || p.Nq < 1) GYRE_FAIL(-1, "attention: empty sequence");
    switch (p.D) {
        case 8: return launch_attn_t<8, 2>(st, p);
        case 16: return launch_attn_t<16, 2>(st, p);
        case 32: return launch_attn_t<32, 2>(st, p);
        case 40: return launch_attn_t<40, 2>(st, p);
        case 64: return launch_attn_t<64, 2>(st, p);
        case 80: return launch_attn_t<80, 2>(st, p);
        case 128: return launch_attn_t<128, 2>(st, p);
        case 160: return launch_attn_t<160, 2>(st, p);
        case 512: return launch_attn_t<512, 1>(st, p);
        default: GYRE_FAIL(-6, "attention: unsupported head dim " + std::to_string(p.D));
    }
}
