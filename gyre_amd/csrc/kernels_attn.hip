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
//
// Three kernels, chosen in launch_attention():
//   k_attn   (v1)  register-staged K / V^T tiles; any head dim up to 512 (the VAE mid block)
//   k_attn2  (v2)  LDS-DMA ring; plain online softmax, or FOLD (K prescaled: scale in the to_k weights, running max in
//                  the MFMA C input) for short key sequences (cross-attention) and D = 160
//   k_attn3  (v3)  FOLD + software pipeline inside the wave (exp of tile t overlaps QK^T of tile t+1); all UNet
//                  self-attention with D in {16,32,40,64,80}
#include "kernels.h"
#include <type_traits>

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
    const float sc = p.k_prescaled ? 1.0f : rsqrtf((float)D) * 1.4426950408889634f;  // scale * log2(e)

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
                    s[qi][ki] = GYRE_MFMA_16x16x32(kf, qf[qi][ks], s[qi][ki], 0, 0, 0);
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
                    o[qi][di] = GYRE_MFMA_16x16x32(vf, pf[qi], o[qi][di], 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------
// v2 (head dims <= 160): K / V^T tiles arrive by LDS-DMA (global_load_lds, 16 B per lane, no VGPR round
// trip) into a 2-deep ring, the loads of tile t+1 are in flight while tile t is computed (counted
// s_waitcnt vmcnt + raw s_barrier so the DMA survives the barrier).  The LDS image of a DMA is
// lane-linear, so every layout decision is made on the per-lane SOURCE address:
//   * K rows are stored densely ([64][D] bf16, no padding) in the permuted key order the S^T/P register
//     layout needs (see krow in v1); the zero padding of the contraction (D -> multiple of 32) lives in
//     the Q fragments only: a K fragment read past a row end sees the next row's (finite) data times 0.
//   * V^T rows are 128 B ([d][64 keys]) with the 16-byte slots XOR-swizzled by (d & 7) -> conflict-free
//     ds_read_b128 of the MFMA A operand.
//   * key / dim tails fetch from a zero page.
// Softmax: p = exp2(fma(s, scale*log2e, -m*scale*log2e)) - one FMA + one v_exp per score.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_a;
typedef __attribute__((address_space(1))) const void gbl_void_a;

// FOLD (needs AttnParams::k_prescaled): K carries log2(e)/sqrt(D), so the MFMA output is already the exp2 argument
// up to the row offset - and that offset (the reference maximum of the row) is subtracted by the MFMA itself through its
// C input.  p = exp2(S') then costs one v_exp per score and nothing else; rows are re-centred in a wave-uniform branch
// on the first tile and whenever a score exceeds the reference maximum by 2^TAU (practically never afterwards).
// QLOOP (short key sequences - the 77-token text context: every key tile fits its own ring slot): a workgroup stages K / V^T of
// its (batch, head) ONCE and then walks `qiter` consecutive query blocks with no barrier and no further DMA, the next block's
// Q fragments being fetched while the current one is computed.  The one-block form spends most of its time in the serial
// chain Q load -> tile DMA -> barrier -> 2 tiles -> store (47 us for the 64x64 cross-attention, 84 MB: 1.8 TB/s).
template <int D, int QI, int PD, bool FOLD, bool QLOOP = false>
__global__ __launch_bounds__(256, (D <= 80 ? 2 : 1)) void k_attn2(AttnParams p, const bf16_t* zero, int qiter) {
    constexpr int NS = PD + 2;               // ring depth: PD tiles in flight + the one being computed + one spare,
                                             // so a refill never targets a stage a slower wave may still be reading
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int KS = DP / 32;
    constexpr int DO = (D + 15) / 16;
    constexpr int KVEC = D / 8;              // 16-byte granules per K row
    constexpr int VR = DO * 16;              // V^T rows staged (rows >= D come from the zero page)
    constexpr int KBYTES = 64 * D * 2;
    constexpr int RAW = KBYTES + VR * 128;
    constexpr int STAGE = (RAW + 4095) / 4096 * 4096;   // whole number of 4-wave DMA rounds
    constexpr int NW = STAGE / 4096;         // DMA instructions per wave per tile
    constexpr int KG = 64 * KVEC, VG = VR * 8;
    static_assert(KG % 64 == 0, "K granules fill whole wave instructions");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    int q0 = (QLOOP ? blockIdx.x * qiter : blockIdx.x) * (64 * QI) + wave * (16 * QI);
    const bf16_t* qb = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* kb = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* vb = p.vt + ((size_t)b * p.H * D + (size_t)h * D) * p.ldvt;

    // FOLD with a padded head dim (D = 40: V^T rows 40..47 are padding): row D of the staged V^T tile is fetched from a
    // page of ones, so PV row D accumulates sum_k p = the softmax denominator inside the matrix core - no VALU adds,
    // and the denominator sees exactly the bf16-rounded p the numerator does.
    constexpr bool ONES = FOLD && (D % 16 != 0);
    // ---- per-lane DMA descriptors: element offset at tile 0, key index used for the tail check ----------
    int off[NW], kq[NW];
    bool one_row[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int g = (i * 4 + wave) * 64 + lane;
        if (g < KG) {
            const int rho = g / KVEC, vec = g - rho * KVEC;
            // LDS row rho of fragment ki = rho>>4 holds key 32*(ki>>1) + 8*(i>>2) + 4*(ki&1) + (i&3), i = rho&15
            const int ki = rho >> 4, ii = rho & 15;
            const int key = 32 * (ki >> 1) + 8 * (ii >> 2) + 4 * (ki & 1) + (ii & 3);
            off[i] = key * p.ldk + vec * 8;
            kq[i] = key;
        } else if (g - KG < VG) {
            const int gv = g - KG, d = gv >> 3, sl = gv & 7;
            const int kg = sl ^ (d & 7);
            off[i] = d * p.ldvt + kg * 8;
            kq[i] = d < D ? kg * 8 : (1 << 28);
            one_row[i] = ONES && d == D;
        } else {
            off[i] = 0; kq[i] = 1 << 28; one_row[i] = false;
        }
        if (g < KG) one_row[i] = false;
    }
    auto issue = [&](int kv0, int st) {
        char* sbase = smem + st * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const bool isk = (i * 4 + wave) * 64 < KG;   // wave-uniform
            const bf16_t* src = (ONES && one_row[i]) ? zero + 128 : zero;
            if (kv0 + kq[i] < p.Nk) src = isk ? kb + (size_t)kv0 * p.ldk + off[i] : vb + kv0 + off[i];
            __builtin_amdgcn_global_load_lds((gbl_void_a*)src, (lds_void_a*)(sbase + i * 4096), 16, 0, 0);
        }
    };

    // ---- Q fragments ---------------------------------------------------------------------------------------
    bf16x8_t qf[QI][KS];
    auto load_q = [&](int qq0, bf16x8_t (&dst)[QI][KS]) {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            const int q = qq0 + qi * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = ks * 32 + fq * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (q < p.Nq && d < D) v = *(const uint4*)(qb + (size_t)q * p.ldq + d);
                dst[qi][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
    };
    load_q(q0, qf);
    f32x4_t o[QI][DO];
    float m_run[QI], l_run[QI];
    auto reset = [&]() {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            m_run[qi] = FOLD ? 0.f : -1e30f; l_run[qi] = 0.f;
#pragma unroll
            for (int di = 0; di < DO; ++di) o[qi][di] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    reset();
    const float sc = p.k_prescaled ? 1.0f : rsqrtf((float)D) * 1.4426950408889634f;
    // FOLD: a row is re-centred only when a score exceeds its reference maximum by 2^TAU.  p <= 2^60, row sums
    // <= 2^60 * Nk and the fp32 PV accumulators stay far inside the fp32 range; every row keeps a term >= 1 from
    // the tile that set its reference, so nothing underflows either.
    constexpr float TAU = GYRE_ATTN_TAU;       // 60 (bf16 storage); 14 with fp16 storage, whose P operand ends at 65504 (common.h)

    // half_tag: a tail tile whose keys 32..63 all lie past Nk (the second tile of a 77-key context holds 13 keys): the two key
    // fragments and the PV step of that half are skipped instead of masked
    auto tile = [&](int t, int kv0, auto tail_tag, auto half_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        constexpr bool HALF = decltype(half_tag)::value;
        constexpr int NKI = HALF ? 2 : 4;
        const char* k_lds = smem + (t % NS) * STAGE;
        const char* v_lds = k_lds + KBYTES;

        f32x4_t s[QI][4];
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            const float c0 = FOLD ? -m_run[qi] : 0.f;     // FOLD: S' = S - m_ref straight out of the matrix core
#pragma unroll
            for (int ki = 0; ki < NKI; ++ki) s[qi][ki] = f32x4_t{c0, c0, c0, c0};
        }
#pragma unroll
        for (int ki = 0; ki < NKI; ++ki) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t kf = __builtin_bit_cast(
                    bf16x8_t, *(const uint4*)(k_lds + (ki * 16 + fr) * (D * 2) + (ks * 4 + fq) * 16));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    s[qi][ki] = GYRE_MFMA_16x16x32(kf, qf[qi][ks], s[qi][ki], 0, 0, 0);
            }
        }
        if (FOLD) {
            // p = exp2(S') with no per-score arithmetic in the common case.  The reference maximum of a row moves only
            // on the first tile (to the true maximum) or when a score exceeds it by more than 2^TAU - a wave-uniform
            // branch that is essentially never taken after the first tiles.
            float mxs[QI];
            bool upd = t == 0;
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                float mx = -1e30f;
#pragma unroll
                for (int ki = 0; ki < NKI; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (TAIL) {
                            const int kvl = 32 * (ki >> 1) + 8 * fq + 4 * (ki & 1) + r;
                            if (kv0 + kvl >= p.Nk) s[qi][ki][r] = -1e30f;
                        }
                        mx = fmaxf(mx, s[qi][ki][r]);
                    }
                mxs[qi] = mx;              // this lane's 16 keys only: enough for the overflow check (ballot below)
                upd |= mx > TAU;
            }
            if (__any(upd)) {
                const float lo = t == 0 ? -1e30f : 0.f;
#pragma unroll
                for (int qi = 0; qi < QI; ++qi) {
                    float mx = mxs[qi];
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float delta = fmaxf(mx, lo);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    m_run[qi] += delta;
                    l_run[qi] *= alpha;
#pragma unroll
                    for (int ki = 0; ki < NKI; ++ki)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qi][ki][r] -= delta;
#pragma unroll
                    for (int di = 0; di < DO; ++di) {
                        o[qi][di][0] *= alpha; o[qi][di][1] *= alpha; o[qi][di][2] *= alpha; o[qi][di][3] *= alpha;
                    }
                }
            }
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                float ls = 0.f;
#pragma unroll
                for (int ki = 0; ki < NKI; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(s[qi][ki][r]);
                        s[qi][ki][r] = pv;
                        if (!ONES) ls += pv;
                    }
                if (!ONES) l_run[qi] += ls;
            }
        }
#pragma unroll
        for (int qi = 0; qi < (FOLD ? 0 : QI); ++qi) {
            float mx = -1e30f;
#pragma unroll
            for (int ki = 0; ki < NKI; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (TAIL) {
                        const int kvl = 32 * (ki >> 1) + 8 * fq + 4 * (ki & 1) + r;
                        if (kv0 + kvl >= p.Nk) s[qi][ki][r] = -1e30f;
                    }
                    mx = fmaxf(mx, s[qi][ki][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qi], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qi] - m_new) * sc);
            const float nm = -m_new * sc;
            m_run[qi] = m_new;
            float ls = 0.f;
#pragma unroll
            for (int ki = 0; ki < NKI; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(s[qi][ki][r], sc, nm));
                    s[qi][ki][r] = pv;
                    ls += pv;
                }
            l_run[qi] = l_run[qi] * alpha + ls;
#pragma unroll
            for (int di = 0; di < DO; ++di) {
                o[qi][di][0] *= alpha; o[qi][di][1] *= alpha; o[qi][di][2] *= alpha; o[qi][di][3] *= alpha;
            }
        }
#pragma unroll
        for (int ks2 = 0; ks2 < NKI / 2; ++ks2) {
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
                const int d = di * 16 + fr;
                uint4 vraw = *(const uint4*)(v_lds + d * 128 + (((ks2 * 4 + fq) ^ (d & 7)) * 16));
                if (TAIL) {  // keys >= Nk inside a partially valid granule: whatever the V^T pad columns hold, use 0
                    const int nvalid = p.Nk - (kv0 + ks2 * 32 + fq * 8);
                    uint32_t w[4] = {vraw.x, vraw.y, vraw.z, vraw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (2 * e >= nvalid) w[e] = 0;
                        else if (2 * e + 1 >= nvalid) w[e] &= 0xffffu;
                    }
                    vraw = make_uint4(w[0], w[1], w[2], w[3]);
                }
                bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vraw);
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    o[qi][di] = GYRE_MFMA_16x16x32(vf, pf[qi], o[qi][di], 0, 0, 0);
            }
        }
    };
    const int nt = (p.Nk + 63) / 64;
    auto finish = [&](int qq0) {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            float l;
            if (ONES) {   // O^T row D (block D/16, local row D%16 -> lanes fq = (D%16)/4, register (D%16)%4) for query fr
                l = __shfl(o[qi][D / 16][(D % 16) % 4], ((D % 16) / 4) * 16 + fr);
            } else {
                l = l_run[qi];
                l += __shfl_xor(l, 16);
                l += __shfl_xor(l, 32);
            }
            const float inv = 1.0f / l;
            const int q = qq0 + qi * 16 + fr;
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
    };
    if constexpr (QLOOP) {
        // nt <= NS (launcher): tile t lives in slot t for the whole kernel
        for (int t0 = 0; t0 < nt; ++t0) issue(t0 * 64, t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < qiter; ++it) {
            if (q0 >= p.Nq) break;                               // per wave: nothing below synchronises
            bf16x8_t qn[QI][KS];
            const bool more = it + 1 < qiter;
            if (more) load_q(q0 + 64 * QI, qn);                  // in flight under this block's tiles
            for (int t = 0; t < nt; ++t) {
                const int kv0 = t * 64;
                if (kv0 + 32 >= p.Nk) tile(t, kv0, std::true_type{}, std::true_type{});
                else if (kv0 + 64 > p.Nk) tile(t, kv0, std::true_type{}, std::false_type{});
                else tile(t, kv0, std::false_type{}, std::false_type{});
            }
            finish(q0);
            if (!more) break;
#pragma unroll
            for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[qi][ks] = qn[qi][ks];
            reset();
            q0 += 64 * QI;
        }
        return;
    }
#pragma unroll
    for (int t0 = 0; t0 < PD; ++t0)
        if (t0 < nt) issue(t0 * 64, t0);
    for (int t = 0; t < nt; ++t) {
        const int kv0 = t * 64;
        if (t + PD < nt) issue(kv0 + PD * 64, (t + PD) % NS);
        // wait until tile t has landed; the (up to PD) younger tiles stay in flight (counted vmcnt)
        const int ahead = min(PD, nt - 1 - t);
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NW) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's share of tile t is in LDS (the only barrier per tile)
        if (kv0 + 32 >= p.Nk) tile(t, kv0, std::true_type{}, std::true_type{});
                else if (kv0 + 64 > p.Nk) tile(t, kv0, std::true_type{}, std::false_type{});
                else tile(t, kv0, std::false_type{}, std::false_type{});
    }
    finish(q0);
}


// ------------------------------------------------------------------------------------------------
// v3 (prescaled K, D <= 64): the folded-softmax kernel, software-pipelined inside each wave.  One step of the loop
// holds two score tiles in registers:
//     A  p = exp2(S'_t), pack to bf16                       (VALU / transcendental)
//     B  S'_{t+1} = K_{t+1} Q^T - m_ref                     (MFMA, independent of A -> overlaps it)
//     C  O^T += V^T_t P^T_t                                 (MFMA, needs A)
//     D  overflow check of S'_{t+1} (+ rare re-centre)      (VALU, overlaps C)
// so the matrix core works on the next tile's scores while the vector unit exponentiates the current one, instead
// of the two phases alternating.  Same LDS-DMA ring as v2; tile t+1 is waited for (counted vmcnt + one barrier) at
// the top of step t and the slot of tile t-1 is refilled right after that barrier.  Loads past the last tile are
// issued against the zero page so that the in-flight count stays constant (drained before the wave ends).
// ------------------------------------------------------------------------------------------------
// ABL: timing ablations (GYRE_ATTN_ABLATIONS builds, results are garbage): 1 = no exponentials, 2 = no tile requests after the
// prologue, 4 = no per-tile barrier / wait, 8 = no MFMAs, 16 = no K / V^T fragment reads (one stale fragment), 32 = no bf16 packing
// Ring discipline (round 5; the cause of the round-4 "two concurrent handles are not bit-reproducible" anomaly).  A refill of ring
// slot X is a WRITE into LDS that nothing orders behind an earlier ds_read of X except the reader's own lgkmcnt wait followed by a
// barrier the refilling wave has passed.  The round-4 loop refilled, right after the barrier of step t+1, the slot whose V^T tile
// step t had just read - and WITHOUT the per-tile check (whose v_max3 chain needs every score, hence every read, before the step
// ends) hipcc sinks the last PV MFMAs of step t and the lgkmcnt wait in front of them BELOW that barrier (the barrier builtin
// orders memory operations, not register-only MFMAs or the waits the compiler derives for them): a wave could pass the barrier with
// its last two V^T ds_read_b128 still queued while another wave's LDS-DMA request for the same slot was already on its way.  On an
// idle CU the read wins by hundreds of cycles; beside a second UNet's kernels sharing the CU's LDS / TA queues it occasionally lost,
// and a few rows of one attention launch read the NEXT ring generation of V^T (finite, plausible values: "a rounding-path sized"
// difference at the UNet output, ~1 call in 12).  The checked pass never showed it because its check pins the reads inside the
// step.  Fix, by construction instead of by timing:
//   * LDSX (the ring has a slot to spare: NS = PD + 3): the refill after barrier B_t targets the slot last read in step t-2; every
//     wave executes s_waitcnt lgkmcnt(0) right AFTER B_{t-1} (the reads are a barrier old by then - the wait is free), so all reads of
//     that slot have RETURNED before any wave can pass B_t.
//   * otherwise (NS = PD + 2; D = 80, whose stage is 24 KB): s_waitcnt lgkmcnt(0) in FRONT of the barrier.
// The redo flag lives in the dynamic LDS region too (word 0 of slot 0, three barriers once per workgroup): a second __shared__
// object made hipcc put s_waitcnt vmcnt(0) in front of the first V^T read of EVERY step (LDS-DMA alias scopes), which drained the
// whole DMA lookahead - the reason the checked kernel got slower between rounds 3 and 4.
__host__ __device__ constexpr int attn3_stage_bytes(int D) { return (64 * D * 2 + ((D + 15) / 16) * 16 * 128 + 4095) / 4096 * 4096; }
__host__ __device__ constexpr bool attn3_spare_slot(int D, int PD) { return 2 * (PD + 3) * attn3_stage_bytes(D) <= 160 * 1024; }
struct AttnRtCheck { bool on; };
template <class T> __device__ __forceinline__ constexpr bool attn_check_on(T) { return T::value; }
__device__ __forceinline__ bool attn_check_on(AttnRtCheck c) { return c.on; }

template <int D, int PD, int QI = 2, int ABL = 0>
__global__ __launch_bounds__(256, (QI >= 4 ? 1 : 2)) void k_attn3(AttnParams p, const bf16_t* zero) {
    // (a 16-wide v_mfma_f32_16x16x16_bf16 step for the head-dim remainder - D = 40 as 32 + 16 instead of 64 - was
    // measured: no faster (the matrix core is not the limiter) and a dependent x32 -> x16 chain on one accumulator
    // gave wrong results for D = 80 under this compiler, so the contraction stays padded to 32)
    constexpr int KS = (D + 31) / 32;
    constexpr int DO = (D + 15) / 16;
    constexpr int KVEC = D / 8;
    constexpr int VR = DO * 16;
    constexpr int KBYTES = 64 * D * 2;
    constexpr int RAW = KBYTES + VR * 128;
    constexpr int STAGE = (RAW + 4095) / 4096 * 4096;
    constexpr int NW = STAGE / 4096;
#ifdef GYRE_ATTN_R4_RING      // reproducer build (tools/attn_race_repro.sh): the round-4 ring - the refill targets the slot the previous
                              // step read and nothing waits for that step's fragment reads - under the round-5 code otherwise
    constexpr bool LDSX = false;
#else
    constexpr bool LDSX = attn3_spare_slot(D, PD);      // two workgroups per CU still fit with one more slot
#endif
    constexpr int NS = PD + (LDSX ? 3 : 2);
    constexpr int KG = 64 * KVEC, VG = VR * 8;
    constexpr bool ONES = (D % 16 != 0);
    constexpr float TAU = GYRE_ATTN_TAU;       // 60 (bf16 storage); 14 with fp16 storage, whose P operand ends at 65504 (common.h)
    // D = 80 does not fit 256 registers without a few spills.  Scratch stores count in vmcnt and may retire before
    // older loads, so a counted wait is not safe there: drain completely instead (one tile less of DMA lookahead).
    constexpr bool DRAIN = D > 64;
    static_assert(KG % 64 == 0, "K granules fill whole wave instructions");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // 1-D grid, XCD-aware order: workgroups w, w+8, ... share an XCD (and its L2); give each XCD a contiguous run of
    // (batch*head, query block) ids with the query block fastest, so the query blocks that stream the same K / V^T
    // hit one L2 instead of fetching them into all eight (PMC: 755 MB -> K/V read once per XCD run)
    const int nqb = (p.Nq + 64 * QI - 1) / (64 * QI);
    const int ntot = nqb * p.B * p.H;
    int wid;
    {
        const int qn = ntot >> 3, rn = ntot & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + loc;
    }
    const int bh = wid / nqb, b = bh / p.H, h = bh % p.H;
    const int q0 = (wid - bh * nqb) * (64 * QI) + wave * (16 * QI);
    const bf16_t* qb = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const bf16_t* kb = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* vb = p.vt + ((size_t)b * p.H * D + (size_t)h * D) * p.ldvt;

    int off[NW], kq[NW];
    bool one_row[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int g = (i * 4 + wave) * 64 + lane;
        one_row[i] = false;
        if (g < KG) {
            const int rho = g / KVEC, vec = g - rho * KVEC;
            const int ki = rho >> 4, ii = rho & 15;
            const int key = 32 * (ki >> 1) + 8 * (ii >> 2) + 4 * (ki & 1) + (ii & 3);
            off[i] = key * p.ldk + vec * 8;
            kq[i] = key;
        } else if (g - KG < VG) {
            const int gv = g - KG, d = gv >> 3, sl = gv & 7;
            const int kg = sl ^ (d & 7);
            off[i] = d * p.ldvt + kg * 8;
            kq[i] = d < D ? kg * 8 : (1 << 28);
            one_row[i] = ONES && d == D;
        } else {
            off[i] = 0; kq[i] = 1 << 28;
        }
    }
    // Tiles are requested in order (kv0 = 0, 64, 128, ...), so every lane keeps a RUNNING source pointer per DMA piece: for a
    // tile that lies wholly inside the key range the request is the pointer itself and one 64-bit add advances it (padding
    // rows / the ones row point at the zero page with a zero step) - the per-lane bounds test + pointer select of the general
    // form (2 x v_cmp + 4 x v_cndmask + address arithmetic per piece: ~70 of the ~230 vector instructions of a tile) is left
    // to the tail tile and to the past-the-end requests.
    // (head dims up to 40 only: three pieces per wave; the nine / twelve extra registers make the D = 64 / 80 forms spill,
    // and a counted-vmcnt kernel must not spill)
    constexpr bool RUNPTR = D <= 40;
    const char* pcur[NW];
    unsigned pinc[NW];
    auto reset_pointers = [&]() {
#pragma unroll
        for (int i = 0; RUNPTR && i < NW; ++i) {
            const bool isk = (i * 4 + wave) * 64 < KG;   // wave-uniform
            const bool live = kq[i] < (1 << 27);
            pcur[i] = live ? (const char*)(isk ? kb + off[i] : vb + off[i]) : (const char*)((ONES && one_row[i]) ? zero + 128 : zero);
            pinc[i] = live ? (isk ? 64u * (unsigned)p.ldk * 2u : 128u) : 0u;
        }
    };
    // The requests are inline asm (M0 written in the statement that uses it), NOT the builtin: for a builtin LDS-DMA hipcc's
    // wait-count pass makes the next ds_read of ANY LDS address wait for it - an s_waitcnt vmcnt(0) in front of the first V^T read
    // of every step, which drained the whole lookahead of this ring (found in round 5 in the ISA of rounds 3 / 4).  Completion
    // is tracked by the counted waits of step() alone.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const void* src, unsigned dst) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
    };
    auto issue = [&](int kv0, int st) {
        const unsigned sbase = lds0 + st * STAGE + wave * 1024;
        if (RUNPTR && kv0 + 64 <= p.Nk) {                // wave-uniform: a full tile
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                dma16(pcur[i], sbase + i * 4096);
                pcur[i] += pinc[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const bool isk = (i * 4 + wave) * 64 < KG;   // wave-uniform
            const bf16_t* src = (ONES && one_row[i]) ? zero + 128 : zero;
            if (kv0 + kq[i] < p.Nk) src = isk ? kb + (size_t)kv0 * p.ldk + off[i] : vb + kv0 + off[i];
            dma16(src, sbase + i * 4096);
        }
    };

    // Q fragments: loaded by hand as well (inline asm, rows / head-dim padding past the end from the zero page) and waited for by
    // the first pass's own prologue wait.  A compiler-visible global_load anywhere in the kernel makes hipcc put an s_waitcnt
    // vmcnt(0) in front of the first use of its registers INSIDE the tile loop (the loop header merges "maybe still pending"), and
    // the compiler's vmcnt(0) cannot see - so it drains - the hand-counted DMA queue.
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));     // (a native vector: HIP's uint4 struct cannot be an asm register operand)
    u32x4_t qraw[QI][KS];
    auto load_q = [&]() {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            const int q = q0 + qi * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = ks * 32 + fq * 8;
                const bf16_t* src = (q < p.Nq && d < D) ? qb + (size_t)q * p.ldq + d : zero;
                u32x4_t v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");
                qraw[qi][ks] = v;
            }
        }
    };
    bf16x8_t qf[QI][KS];
    f32x4_t o[QI][DO];
    float m_ref[QI], l_run[QI];
    // -m_ref of each query block as the MFMA's C operand, kept as a register quad for the whole run: the reference only changes
    // in the (rare, wave-uniform) re-centring branch, so no tile has to re-materialise it (8 x v_mov + 2 x v_xor per tile before)
    // (head dims up to 40: the eight registers make the D = 64 form spill, and a counted-vmcnt kernel must not)
    constexpr bool CNEG = D <= 40;
    f32x4_t cneg[QI];

    // B: scores of tile t (relative to the rows' reference maxima), keys past Nk masked when TAIL
    auto qk = [&](int t, f32x4_t (&s)[QI][4], auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const char* k_lds = smem + (t % NS) * STAGE;
        if constexpr (!CNEG) {
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                const float c0 = -m_ref[qi];
#pragma unroll
                for (int ki = 0; ki < 4; ++ki) s[qi][ki] = f32x4_t{c0, c0, c0, c0};
            }
        }
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8_t kf = __builtin_bit_cast(
                    bf16x8_t, *(const uint4*)(k_lds + (((ABL & 16) ? 0 : ki) * 16 + fr) * (D * 2) + (((ABL & 16) ? 0 : ks) * 4 + fq) * 16));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi) {
                    if constexpr (ABL & 8) { if (ks == 0 && CNEG) s[qi][ki] = cneg[qi]; asm volatile("" ::"v"(kf), "v"(qf[qi][ks])); }
                    else s[qi][ki] = GYRE_MFMA_16x16x32(kf, qf[qi][ks], (CNEG && ks == 0) ? cneg[qi] : s[qi][ki], 0, 0, 0);
                }
            }
        }
        if (TAIL) {
            const int kv0 = t * 64;
#pragma unroll
            for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kvl = 32 * (ki >> 1) + 8 * fq + 4 * (ki & 1) + r;
                        if (kv0 + kvl >= p.Nk) s[qi][ki][r] = -1e30f;
                    }
        }
    };
    // D: overflow check; re-centre the rows (wave-uniform branch: first tile, or a score above 2^TAU)
    auto check = [&](bool first, f32x4_t (&s)[QI][4]) {
        float mxs[QI];
        bool upd = first;
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            float mx = -1e30f;
#pragma unroll
            for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qi][ki][r]);
            mxs[qi] = mx;
            upd |= mx > TAU;
        }
        if (__any(upd)) {
            const float lo = first ? -1e30f : 0.f;
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                float mx = mxs[qi];
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float delta = fmaxf(mx, lo);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_ref[qi] += delta;
                if constexpr (CNEG) { const float c0 = -m_ref[qi]; cneg[qi] = f32x4_t{c0, c0, c0, c0}; }
                l_run[qi] *= alpha;
#pragma unroll
                for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[qi][ki][r] -= delta;
#pragma unroll
                for (int di = 0; di < DO; ++di) {
                    o[qi][di][0] *= alpha; o[qi][di][1] *= alpha; o[qi][di][2] *= alpha; o[qi][di][3] *= alpha;
                }
            }
        }
    };
    // A: p = exp2(S') in place, packed to the bf16 MFMA operand
    auto softmax_pack = [&](f32x4_t (&s)[QI][4], bf16x8_t (&pf)[2][QI]) {
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            float ls = 0.f;
#pragma unroll
            for (int ki = 0; ki < 4; ++ki)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = (ABL & 1) ? s[qi][ki][r] * 0.5f : __builtin_amdgcn_exp2f(s[qi][ki][r]);
                    s[qi][ki][r] = pv;
                    if (!ONES) ls += pv;
                }
            if (!ONES) l_run[qi] += ls;
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) {
                uint4 w;
                if constexpr (ABL & 32) { w = make_uint4(__float_as_uint(s[qi][2 * ks2][0]), __float_as_uint(s[qi][2 * ks2][2]), __float_as_uint(s[qi][2 * ks2 + 1][0]), __float_as_uint(s[qi][2 * ks2 + 1][2])); pf[ks2][qi] = __builtin_bit_cast(bf16x8_t, w); continue; }
                w.x = pack_bf16x2(s[qi][2 * ks2][0], s[qi][2 * ks2][1]);
                w.y = pack_bf16x2(s[qi][2 * ks2][2], s[qi][2 * ks2][3]);
                w.z = pack_bf16x2(s[qi][2 * ks2 + 1][0], s[qi][2 * ks2 + 1][1]);
                w.w = pack_bf16x2(s[qi][2 * ks2 + 1][2], s[qi][2 * ks2 + 1][3]);
                pf[ks2][qi] = __builtin_bit_cast(bf16x8_t, w);
            }
    };
    // C: O^T += V^T P^T for tile t
    auto pv = [&](int t, bf16x8_t (&pf)[2][QI], auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const char* v_lds = smem + (t % NS) * STAGE + KBYTES;
        const int kv0 = t * 64;
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
#pragma unroll
            for (int di = 0; di < DO; ++di) {
                const int d = di * 16 + fr;
                uint4 vraw = *(const uint4*)(v_lds + ((ABL & 16) ? fr : d) * 128 + (((((ABL & 16) ? 0 : ks2) * 4 + fq) ^ (d & 7)) * 16));
                if (TAIL) {
                    const int nvalid = p.Nk - (kv0 + ks2 * 32 + fq * 8);
                    uint32_t w[4] = {vraw.x, vraw.y, vraw.z, vraw.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (2 * e >= nvalid) w[e] = 0;
                        else if (2 * e + 1 >= nvalid) w[e] &= 0xffffu;
                    }
                    vraw = make_uint4(w[0], w[1], w[2], w[3]);
                }
                bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vraw);
#pragma unroll
                for (int qi = 0; qi < QI; ++qi) {
                    if constexpr (ABL & 8) asm volatile("" ::"v"(vf), "v"(pf[ks2][qi]));
                    else o[qi][di] = GYRE_MFMA_16x16x32(vf, pf[ks2][qi], o[qi][di], 0, 0, 0);
                }
            }
        }
    };
    // one pipeline step: cur = scores of tile t (checked when CHECK), next receives tile t+1
    auto step = [&](int t, f32x4_t (&cur)[QI][4], f32x4_t (&next)[QI][4], auto has_next, auto tail_next, auto tail_cur, auto check_tag) {
        constexpr bool HASNEXT = decltype(has_next)::value;
        if (HASNEXT) {
            // tile t+1 landed (tiles t+2 .. t+PD may still be in flight); ring discipline: see the kernel's header
#ifndef GYRE_ATTN_R4_RING
            if constexpr (!LDSX) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            if constexpr (ABL & 6) {}
            else if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NW) : "memory");
            if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
            if constexpr (LDSX) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(ABL & 2)) issue((t + 1 + PD) * 64, (t + 1 + PD) % NS);
        }
        bf16x8_t pf[2][QI];
        softmax_pack(cur, pf);
        if (HASNEXT) qk(t + 1, next, tail_next);
        pv(t, pf, tail_cur);
        if (HASNEXT && attn_check_on(check_tag)) check(false, next);
    };
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    const int nt = (p.Nk + 63) / 64;

    // One pass over the keys.  CHECK = false (the first, optimistic pass): the rows are centred on the maximum of the FIRST key
    // tile and no later tile is examined - the per-tile overflow check (16 x v_max3 + compare + branch per wave tile: every
    // vector instruction beyond the ~4 an MFMA covers costs the SIMD ~3 cycles, tools/ubench/mfma_mix.hip) is paid by no
    // tile.  A score more than ~127 above the first tile's maximum would overflow exp2 to inf; any smaller excess is exact
    // floating-point arithmetic (P is rounded to bf16 relative to its own magnitude, O and the row sum stay finite).  The
    // overflow leaves a non-finite (or zero) row sum, which the end of the pass detects; the workgroup then repeats the whole
    // pass with CHECK = true (re-centring whenever a score exceeds the reference by 2^TAU) - never observed on SD activations,
    // exercised by tests/test_gpu_kernels.py::test_attention_folded_softmax_recentres.
    auto pass = [&](auto check_tag, auto loadq_tag) {
        constexpr bool LOADQ = decltype(loadq_tag)::value;
        reset_pointers();
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            m_ref[qi] = 0.f; l_run[qi] = 0.f;
            if constexpr (CNEG) cneg[qi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int di = 0; di < DO; ++di) o[qi][di] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t0 = 0; t0 <= PD; ++t0) issue(t0 * 64, t0);
        if constexpr (LOADQ) {
            // the first pass: Q behind the prologue's tile requests, everything waited for at once; the statement names every
            // destination so that no use of a Q register can be scheduled above the wait (guide 5.7 item 1, form ii)
            load_q();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    u32x4_t v = qraw[qi][ks];
                    asm volatile("" : "+v"(v));
                    qf[qi][ks] = __builtin_bit_cast(bf16x8_t, v);
                }
        } else if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * NW) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x4_t sa[QI][4], sb[QI][4];
        if (nt == 1) qk(0, sa, std::true_type{}); else qk(0, sa, std::false_type{});
        check(true, sa);

        const int n_main = nt >= 2 ? nt - 2 : 0;
        int t = 0;
        for (; t + 1 < n_main; t += 2) {
            step(t, sa, sb, T, F, F, check_tag);
            step(t + 1, sb, sa, T, F, F, check_tag);
        }
        bool flip = false;
        if (t < n_main) { step(t, sa, sb, T, F, F, check_tag); ++t; flip = true; }
        if (nt >= 2) {
            if (!flip) step(t, sa, sb, T, T, F, check_tag); else step(t, sb, sa, T, T, F, check_tag);
            ++t; flip = !flip;
        }
        if (!flip) step(t, sa, sb, F, F, T, check_tag); else step(t, sb, sa, F, F, T, check_tag);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the past-the-end zero-page DMAs before the wave exits / the ring restarts
    };
    auto row_sum = [&](int qi) {
        float l;
        if (ONES) {
            l = __shfl(o[qi][D / 16][(D % 16) % 4], ((D % 16) / 4) * 16 + fr);
        } else {
            l = l_run[qi];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        return l;
    };

    // Optimistic first, checked only if needed (head dims up to 40: the 64x64 level of SD1.x).  The D = 64 form with two copies of
    // the loop sits at 256 registers + 9 spilled SGPRs and its redo faulted on the GPU in round 4 (address 0, never isolated); ONE
    // copy with a run-time flag around the check, inside a retry loop, was tried in round 5 and spills for every head dim (the
    // whole kernel becomes a loop body: 24 - 112 spilled VGPRs) - so D >= 64 keeps the checked pass it has always had.
    // The end-of-pass test accepts a row only if its sum is finite, positive and below 2^TAU (round 6; 1e25 ~ 2^83 before): then no p
    // reached 2^TAU, i.e. no score of the row exceeded its first-tile reference by TAU - exactly the condition under which the checked
    // pass never takes its re-centring branch after the first tile.  A workgroup whose rows are all accepted has therefore executed
    // the checked pass's arithmetic instruction for instruction, and a workgroup with one rejected row REPEATS its tile with the
    // checked pass: the default path is bit-identical to variant 7 (always checked) on EVERY input, the data only decides how many
    // workgroups pay twice (gyre_debug_attn_redo_count; 0 on the synthetic weights, bench.py prints it).  With the old bound a row
    // whose excess lay between 2^60 and 2^83 was accepted here and re-centred there: equal to rounding, not to the bit.
    // (p rounds to bf16 monotonically and 2^TAU is representable, so a score above TAU always leaves a sum >= 2^TAU.)
    constexpr bool OPTIMISTIC = D <= 40;
    if (!OPTIMISTIC || p.always_check) pass(T, T);
    else {
        pass(F, T);
        bool bad = false;
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            const float l = row_sum(qi);
            bad |= (q0 + qi * 16 + fr < p.Nq) && !(l > 0.f && l < GYRE_ATTN_BOUND);      // 2^TAU (common.h)
        }
        const bool wave_bad = __any(bad);
        int* flags = (int*)smem;                        // four words of ring slot 0, used between the two passes only
        __syncthreads();                                // every wave is done reading the ring (its DMAs were drained by pass())
        if (lane == 0) flags[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        const int redo = __builtin_amdgcn_readfirstlane(flags[0] | flags[1] | flags[2] | flags[3]);
        if (redo && ABL == 0) {                         // workgroup-uniform: the ring and its barriers are shared by the four waves
            __syncthreads();                            // the flags have been read before the next prologue's DMA lands on them
            if (p.redo_counter && tid == 0) atomicAdd(p.redo_counter, 1u);      // (tuning: how often does it happen?)
            pass(T, F);
        }
    }

#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const float l = row_sum(qi);
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

#include <mutex>
#include <unordered_map>
static const bf16_t* attn_zero_page() {
    static std::mutex mu;
    static std::unordered_map<int, void*> pages;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    auto it = pages.find(dev);
    if (it != pages.end()) return (const bf16_t*)it->second;
    void* p = nullptr;   // 256 B of zeros followed by 256 B of ones in the storage type (GYRE_ONE_BITS), then the device's redo counter (one zeroed word)
    if (hipMalloc(&p, 512 + 16) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess || hipMemset((char*)p + 512, 0, 16) != hipSuccess) return nullptr;
    if (hipMemsetD16((hipDeviceptr_t)((char*)p + 256), GYRE_ONE_BITS, 128) != hipSuccess) return nullptr;
    (void)hipDeviceSynchronize();
    pages[dev] = p;
    return (const bf16_t*)p;
}

// Device counter of the workgroups that repeated their pass with the per-tile check (round 6: always on, one word per device behind
// that device's zero / ones page - the increment only happens on the redo path, so counting costs the optimistic pass nothing;
// bench.py prints the count of its timed region as `attn_redo_count`)
static unsigned* attn_redo_counter() {
    const bf16_t* page = attn_zero_page();
    return page ? (unsigned*)((char*)page + 512) : nullptr;
}
extern "C" long gyre_debug_attn_redo_count() {
    unsigned* c = attn_redo_counter();
    if (!c) return -1;
    unsigned v = 0;
    if (hipMemcpy(&v, c, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)v;
}
static thread_local int g_attn_variant = 0;  // tests / tuning: 0 auto, 1 = v1, 2 = v2 plain, 3 = v2 folded, 4 = v2 QI=4 (D <= 32), 5 = v3,
                                             // 6 = auto without the several-query-blocks-per-workgroup form of short key sequences
extern "C" int gyre_debug_force_attn_variant(int v) { int o = g_attn_variant; g_attn_variant = v; return o; }

template <int D, int QI, bool FOLD = false>
static int launch_attn2_t(hipStream_t st, const AttnParams& p) {
    constexpr int DO = (D + 15) / 16;
    constexpr int RAW = 64 * D * 2 + DO * 16 * 128;
    constexpr int STAGE = (RAW + 4095) / 4096 * 4096;
    constexpr int PD = D <= 80 ? 2 : 1;      // tiles in flight ahead of the one being computed
    const size_t lds = (size_t)(PD + 2) * STAGE;
    const bf16_t* zero = attn_zero_page();
    if (!zero) GYRE_FAIL(-5, "attention: cannot allocate the zero page");
    const int nblk = (p.Nq + 64 * QI - 1) / (64 * QI);
    GyreProfScope prof_(KC_ATTN, st, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * D,
                        2.0 * p.B * p.H * D * (2.0 * p.Nq + 2.0 * p.Nk));
    // short key sequence with many query blocks (cross-attention against the text context): K / V^T staged once per
    // workgroup, several query blocks per workgroup - as many as still leave ~2 workgroups per CU
    if constexpr (D == 40 || D == 64 || D == 80 || D == 160) {
        int qiter = (int)((long)nblk * p.B * p.H / 512);
        if (qiter > 8) qiter = 8;
        if (qiter > nblk) qiter = nblk;
        if ((p.Nk + 63) / 64 <= PD + 2 && qiter >= 2 && ((g_attn_variant & 255) == 0 || (g_attn_variant & 255) == 7 || (g_attn_variant & 255) == 8)) {
            auto kern = k_attn2<D, QI, PD, FOLD, true>;
            static std::atomic<unsigned long long> attr_done{0};
            if (gyre_lds_attr_needed(attr_done))
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3((nblk + qiter - 1) / qiter, p.B * p.H), dim3(256), lds, st, p, zero, qiter);
            GYRE_LAUNCH_CHECK();
            return 0;
        }
    }
    auto kern = k_attn2<D, QI, PD, FOLD>;
    static std::atomic<unsigned long long> attr_done{0};
    if (gyre_lds_attr_needed(attr_done))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nblk, p.B * p.H), dim3(256), lds, st, p, zero, 1);
    GYRE_LAUNCH_CHECK();
    return 0;
}

template <int D, int QI = 2>
static int launch_attn3_t(hipStream_t st, const AttnParams& p) {
    constexpr int DO = (D + 15) / 16;
    constexpr int RAW = 64 * D * 2 + DO * 16 * 128;
    constexpr int STAGE = (RAW + 4095) / 4096 * 4096;
    // D = 80 (24 KB per stage, a few spills: its loop drains the DMA queue every step anyway): one tile of lookahead and a 3-slot
    // ring = 72 KB, TWO workgroups per CU instead of the one its 4 x 24 KB ring allowed (the 32x32 level of SD1.x)
    constexpr int PD = D > 64 ? 1 : 2;
    static_assert(STAGE == attn3_stage_bytes(D), "ring stage size");
#ifdef GYRE_ATTN_R4_RING
    const size_t lds = (size_t)(PD + 2) * STAGE;
#else
    const size_t lds = (size_t)(PD + (attn3_spare_slot(D, PD) ? 3 : 2)) * STAGE;
#endif
    const bf16_t* zero = attn_zero_page();
    if (!zero) GYRE_FAIL(-5, "attention: cannot allocate the zero page");
    auto kern = k_attn3<D, PD, QI>;
    static std::atomic<unsigned long long> attr_done{0};
    if (gyre_lds_attr_needed(attr_done))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)(((p.Nq + 64 * QI - 1) / (64 * QI)) * p.B * p.H));
    GyreProfScope prof_(KC_ATTN, st, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * D,
                        2.0 * p.B * p.H * D * (2.0 * p.Nq + 2.0 * p.Nk));
    AttnParams q = p;
    // optimistic first pass by default (variant 7 = always the per-tile check; 8 = a synonym of the default, kept for the tools)
    q.always_check = (g_attn_variant & 255) == 7 ? 1 : 0;
    q.redo_counter = (unsigned*)((char*)zero + 512);        // the device's counter sits behind its zero / ones page
#ifdef GYRE_ATTN_ABLATIONS
    if constexpr (D == 40) {
        const int abl = g_attn_variant >> 8;              // gyre_debug_force_attn_variant(abl << 8)
#define GYRE_ATTN_ABL(A_) if (abl == A_) { hipLaunchKernelGGL((k_attn3<D, PD, QI, A_>), grid, dim3(256), lds, st, q, zero); GYRE_LAUNCH_CHECK(); return 0; }
        GYRE_ATTN_ABL(1) GYRE_ATTN_ABL(2) GYRE_ATTN_ABL(4) GYRE_ATTN_ABL(6) GYRE_ATTN_ABL(8) GYRE_ATTN_ABL(16) GYRE_ATTN_ABL(32) GYRE_ATTN_ABL(33)
        GYRE_ATTN_ABL(9) GYRE_ATTN_ABL(24) GYRE_ATTN_ABL(57) GYRE_ATTN_ABL(63) GYRE_ATTN_ABL(22)
#undef GYRE_ATTN_ABL
    }
#endif
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q, zero);
    GYRE_LAUNCH_CHECK();
    return 0;
}

template <int D, int QI>
static int launch_attn_t(hipStream_t st, const AttnParams& p) {
    constexpr int DP = (D + 31) / 32 * 32;
    constexpr int DO = (D + 15) / 16;
    const size_t lds = (size_t)64 * (DP * 2 + 16) + (size_t)DO * 16 * (64 * 2 + 16);
    auto kern = k_attn<D, QI>;
    static std::atomic<unsigned long long> attr_done{0};
    if (gyre_lds_attr_needed(attr_done))
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
    const int var = ((g_attn_variant & 255) == 6 || (g_attn_variant & 255) == 7 || (g_attn_variant & 255) == 8) ? 0 : (g_attn_variant & 255);
    // software-pipelined folded kernel; with only a couple of key tiles (cross-attention, Nk = 77) its longer prologue
    // costs more than the overlap wins (measured 47.8 vs 41.1 us), so short key sequences stay on the v2 form
    // (48 query rows per wave, QI = 3, was tried for D = 40: 232 B/lane of spills at 2 waves/SIMD - not built)
    // (64 rows per wave at ONE wave per SIMD, QI = 4 with 412 registers: 625 vs 619 us at 64x64, 10-40 % slower on
    //  the smaller shapes - not built)
    if (p.k_prescaled && ((var == 0 && p.Nk >= 256) || var == 5)) {
        switch (p.D) {
            case 16: return launch_attn3_t<16>(st, p);
            case 32: return launch_attn3_t<32>(st, p);
            case 40: return launch_attn3_t<40>(st, p);
            case 64: return launch_attn3_t<64>(st, p);
            case 80: return launch_attn3_t<80>(st, p);
            default: break;
        }
    }
    if (p.k_prescaled && var != 1 && var != 2 && var != 4) {
        switch (p.D) {
            case 16: return launch_attn2_t<16, 2, true>(st, p);
            case 32: return launch_attn2_t<32, 2, true>(st, p);
            case 40: return launch_attn2_t<40, 2, true>(st, p);
            case 64: return launch_attn2_t<64, 2, true>(st, p);
            case 160: return launch_attn2_t<160, 2, true>(st, p);
            default: break;   // D = 80: the folded form spills 7 registers at 2 waves/SIMD and is no faster (measured)
        }
    }
    if (var != 1) {
        const bool q4 = var == 4;  // 64 query rows per wave: measured slower than 32 at every SD shape (register pressure)
        switch (p.D) {
            case 16: return q4 ? launch_attn2_t<16, 4>(st, p) : launch_attn2_t<16, 2>(st, p);
            case 32: return q4 ? launch_attn2_t<32, 4>(st, p) : launch_attn2_t<32, 2>(st, p);
            // (64 query rows per wave spill from head dim 40 up, which a counted-vmcnt kernel must not: not built)
            case 40: return launch_attn2_t<40, 2>(st, p);
            case 64: return launch_attn2_t<64, 2>(st, p);
            case 80: return launch_attn2_t<80, 2>(st, p);
            case 128: return launch_attn2_t<128, 2>(st, p);
            case 160: return launch_attn2_t<160, 2>(st, p);
            default: break;
        }
    }
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
