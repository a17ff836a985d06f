// Shared device/host helpers for libgyre_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>

// ---- storage flavour --------------------------------------------------------------------------------------------------------------
// The library is built twice from the same sources (gyre_amd/build.py): libgyre_hip.so stores activations and weights as bf16,
// libgyre_hip_f16.so (-DGYRE_STORE_F16) as IEEE fp16 - the reference's own GPU arithmetic (manager.py:146-151,1199-1200 loads fp16) with
// three more mantissa bits at the same MFMA rate (v_mfma_f32_*_f16 = v_mfma_f32_*_bf16 on gfx950); accumulation, statistics and
// every epilogue stay fp32 in both.  Everything type-specific is in this block: the raw 16-bit element type keeps the name bf16_t
// ("the half-width storage element") throughout the kernels, conversions go through the helpers below, the MFMAs through the
// GYRE_MFMA_* names.  gyre_storage_dtype() (include/gyre_hip.h) says which flavour a loaded library is.
typedef uint16_t bf16_t;  // raw bits of the 16-bit storage element (bf16, or fp16 with GYRE_STORE_F16); all activations are NHWC of it
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define GYRE_WAVE 64

#ifdef GYRE_STORE_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;      // MFMA operand fragment (8 storage elements)
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
#define GYRE_STORAGE_DTYPE 2                                          /* GYRE_F16 */
#define GYRE_ONE_BITS 0x3c00                                          /* 1.0 */
#define GYRE_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define GYRE_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
// softmax probabilities are packed to the storage type: fp16 ends at 65504, so rows are re-centred (and the optimistic attention pass
// accepted) at 2^14 instead of 2^60
#define GYRE_ATTN_TAU 14.f
#define GYRE_ATTN_BOUND 16384.f
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }    // round to nearest even
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return (float)__builtin_bit_cast(bf16x2_t, w)[0]; }
__device__ __forceinline__ float bf16hi(uint32_t w) { return (float)__builtin_bit_cast(bf16x2_t, w)[1]; }
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define GYRE_STORAGE_DTYPE 1                                          /* GYRE_BF16 */
#define GYRE_ONE_BITS 0x3f80
#define GYRE_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define GYRE_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define GYRE_ATTN_TAU 60.f
#define GYRE_ATTN_BOUND 1.152921504606846976e18f                      /* 2^60 */
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even via the gfx950 hardware conversion (v_cvt_pk_bf16_f32, one instruction per pair)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf16lo(v.x); f[1] = bf16hi(v.x); f[2] = bf16lo(v.y); f[3] = bf16hi(v.y);
    f[4] = bf16lo(v.z); f[5] = bf16hi(v.z); f[6] = bf16lo(v.w); f[7] = bf16hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}
// x * sigmoid(x) with the hardware reciprocal (1 ulp): the IEEE division of `x / (1 + e^-x)` is a ten-instruction sequence
// (v_div_scale / v_rcp / four v_fma / v_div_fmas / v_div_fixup) per value - 136 of the ~400 vector instructions of a GroupNorm-apply
// loop body were division scaffolding
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf-GELU (the diffusers GEGLU uses the exact erf form), branch-free.  With z = |x|/sqrt(2) and
//   q = erfc(z) ~= (1 + a1 z + ... + a6 z^6)^-16          (Abramowitz-Stegun 7.1.28, |err| <= 3e-7: one reciprocal, no exponential)
//   gelu(x) = max(x, 0) - 0.5 |x| q                        - no cancellation in the negative tail, no compare / select
// (|gelu - exact| <= 7.1e-7 over [-12, 12], relative 2.8e-4 where |gelu| > 1e-3: below bf16 resolution; large |x|: the power
// overflows to inf, q = 0).  OCML's erff is a branchy piecewise routine; in the GEGLU epilogue it cost more than the GEMM main
// loop, and the previous 7.1.26 form (reciprocal AND exponential per value) was still 25 % of the 64x64 FF1 launch.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    float p = fmaf(z, 0.0000430638f, 0.0002765672f);
    p = fmaf(z, p, 0.0001520143f);
    p = fmaf(z, p, 0.0092705272f);
    p = fmaf(z, p, 0.0422820123f);
    p = fmaf(z, p, 0.0705230784f);
    p = fmaf(z, p, 1.0f);
    p *= p; p *= p; p *= p; p *= p;
    const float q = __builtin_amdgcn_rcpf(p);
    return fmaf(-0.70710678118654752f * z, q, fmaxf(x, 0.f));
}

// load / store one element of a BOUNDARY tensor of runtime dtype (gyre_dtype: 0 f32, 1 bf16, 2 f16) - whatever the storage flavour
// of the build: bf16 here is always real bf16, f16 always IEEE half
__device__ __forceinline__ float load_as_f32(const void* p, int dtype, size_t i) {
    if (dtype == 0) return ((const float*)p)[i];
    if (dtype == 1) return __uint_as_float(((uint32_t)((const uint16_t*)p)[i]) << 16);
    return __half2float(((const __half*)p)[i]);
}
__device__ __forceinline__ void store_from_f32(void* p, int dtype, size_t i, float v) {
    if (dtype == 0) ((float*)p)[i] = v;
    else if (dtype == 1) ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    else ((__half*)p)[i] = __float2half(v);
}

// ---- host side error plumbing (thread-local message, int status; nothing throws) ----
void gyre_set_error(const std::string& msg);
int64_t& gyre_launch_counter();
#define GYRE_FAIL(code, msg) do { gyre_set_error(std::string(msg)); return (code); } while (0)
#define GYRE_HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    gyre_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return -5; } } while (0)
#define GYRE_LAUNCH_CHECK() do { gyre_launch_counter()++; hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    gyre_set_error(std::string("kernel launch: ") + hipGetErrorString(e_) + " at " __FILE__ ":" + std::to_string(__LINE__)); \
    return -5; } } while (0)

// ---- optional per-launch timing (HIP events on the launch stream), used by bench.py's roofline leg ----
// Kernel classes; a launch is timed only when profiling is enabled for its class.
enum GyreKernelClass {
    KC_GEMM_CONV_128 = 0, KC_GEMM_CONV_256x64, KC_GEMM_CONV_64, KC_GEMM_LIN_128, KC_GEMM_LIN_256x64, KC_GEMM_LIN_64,
    KC_G8_CONV_256x320, KC_G8_CONV_128x320, KC_G8_CONV_256x256, KC_G8_CONV_128x256,
    KC_G8_LIN_256x320, KC_G8_LIN_128x320, KC_G8_LIN_256x256, KC_G8_LIN_128x256,
    KC_G8_CONV_128x160, KC_G8_X1, KC_G8_X2, KC_G8_X3, KC_G8_LIN_128x160,
    KC_G4S_192x320, KC_G4S_192x320_LIN, KC_G4S_256x256, KC_G4S_256x256_LIN, KC_G4S_128x320, KC_G4S_128x320_LIN,
    KC_G4S_128x256, KC_G4S_128x256_LIN, KC_G4S_256x320, KC_G4S_256x320_LIN,
    KC_ATTN, KC_GN_STATS, KC_GN_APPLY, KC_LAYERNORM, KC_OTHER, KC_ATTN_BWD, KC_NORM_BWD, KC_SPLITK_REDUCE, KC_GEMM_AR, KC_GEMM_SM, KC_XATTN, KC_COUNT
};
struct GyreProfScope {  // RAII: records start/stop events around one launch when enabled
    int slot = -1;
    GyreProfScope(int kclass, hipStream_t st, double flops, double bytes);
    ~GyreProfScope();
    void stop();       // record the end event now (the destructor then does nothing)
    hipStream_t st_;
};

// Large dynamic-LDS kernels need hipFuncAttributeMaxDynamicSharedMemorySize raised once PER DEVICE (a process may drive
// several GPUs from different threads, reference manager.py:2107-2139).  Returns true when the caller still has to set
// the attribute for the current device; `done` is the launcher's function-local bitmask.
#include <atomic>
static inline bool gyre_lds_attr_needed(std::atomic<unsigned long long>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return false;
    done.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

