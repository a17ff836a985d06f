// Pieces shared by the GEMM translation units (kernels_gemm.hip: 4-/8-wave kernels, kernels_gemm4s.hip: the
// one-wave-per-SIMD kernel).
#pragma once
#include "kernels.h"

#define BK 64

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// XCD-aware, bijective tile order: blocks b, b+8, ... run on the same XCD (private L2); give each XCD a
// contiguous chunk of tile ids, tile id -> (tm, tn) with tn fastest so neighbours share the A rows.
__device__ __forceinline__ int xcd_tile_id(int bid, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// 256-byte zero page per device: source of the zero padding for the LDS-DMA kernels (read-only after creation)
const bf16_t* gemm_zero_page_for_current_device();

int launch_splitk_reduce(hipStream_t st, const GemmParams& p, int splits);   // kernels_gemm.hip

// kernels_gemm4s.hip: tile configs 20 (192x320), 21 (256x256), 22 (128x320), 23 (128x256)
bool gemm4s_supports(const GemmParams& p, int cfg);
int launch_gemm4s(hipStream_t st, const GemmParams& p, int cfg, int splits);

// kernels_gemm_ar.hip: tile config 30, the A-resident kernel for K = 320 / 640 linear problems (declarations the model runtime
// needs are in kernels.h)
bool gemm_ar_supports(const GemmParams& p);
int launch_gemm_ar(hipStream_t st, const GemmParams& p, const void* wpk);
int gemm_ar_nsplit(const GemmParams& p);      // N-range splits per row block = partial sums per row in rowstat_out
bool gemm_sm_supports(const GemmParams& p);    // small-problem kernel (kernels_gemm_sm.hip, tile config 32)
int launch_gemm_sm(hipStream_t st, const GemmParams& p);
