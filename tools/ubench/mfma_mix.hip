// Dev microbenchmark: what does a SIMD do when MFMAs, vector ALU instructions and LDS fragment reads share it?
//   hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix && ./mfma_mix
// One workgroup on one CU; W waves per SIMD (block = 256 * W threads) all run the SAME loop body:
//   NM x v_mfma_f32_32x32x16_bf16 (independent accumulators), each followed by KV independent v_fma_f32 and KR ds_read_b128
//   (lane-linear, results consumed one iteration later).
// Prints shader-clock cycles per loop iteration (= per NM MFMAs) for wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define ITER 512

template <int KV, int KR, bool MF, bool DEP>
__global__ void k(float* out, long long* cyc, float seed) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * e;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x * 0.001f;
    f32x4 r8[8] = {};
    ((float*)smem)[threadIdx.x] = seed;
    __syncthreads();
    const unsigned lp = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MF) acc[DEP ? 0 : m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[DEP ? 0 : m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < KV; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(m * KV + j) & 15]));
#pragma unroll
            for (int j = 0; j < KR; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r8[(m * KR + j) & 7]) : "v"(lp), "n"(1024 * ((0 * 4 + j) & 7)));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KR) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");       // the newest four reads stay in flight
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 8; ++i) s += r8[i][0];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int KV, int KR, bool MF, bool DEP>
void run(const char* name, int waves_per_simd, float* out, long long* cyc) {
    hipFuncSetAttribute((const void*)k<KV, KR, MF, DEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KV, KR, MF, DEP>), dim3(1), dim3(256 * waves_per_simd), 65536, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < 4 * waves_per_simd; ++i) mx = c[i] > mx ? c[i] : mx;
    printf("%-46s W=%d: wave 0 %6.1f, slowest wave %6.1f cycles per slot -> %5.1f cycles of the SIMD per slot of ONE wave\n", name, waves_per_simd,
           (double)c[0] / ITER / 4, (double)mx / ITER / 4, (double)mx / ITER / 4 / waves_per_simd);
}
// one wave per SIMD issues MFMAs only, its partner vector instructions only
template <int OP>
__global__ void kpair(float* out, long long* cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    long long t0, t1;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * e;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
        t1 = __builtin_readcyclecounter();
        out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x * 0.001f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j & 15]));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 15]));
            }
        t1 = __builtin_readcyclecounter();
        float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
        out[threadIdx.x] = s;
        if (threadIdx.x == 256) cyc[1] = t1 - t0;
    }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, true, false>("MFMA only (4 independent accumulators)", w, out, cyc);
        run<0, 0, true, true>("MFMA only (ONE accumulator, dependent)", w, out, cyc);
        run<2, 0, true, false>("MFMA + 2 v_fma", w, out, cyc);
        run<4, 0, true, false>("MFMA + 4 v_fma", w, out, cyc);
        run<8, 0, true, false>("MFMA + 8 v_fma", w, out, cyc);
        run<12, 0, true, false>("MFMA + 12 v_fma", w, out, cyc);
        run<16, 0, true, false>("MFMA + 16 v_fma", w, out, cyc);
        run<12, 0, true, true>("dependent MFMA + 12 v_fma", w, out, cyc);
        run<12, 0, false, false>("12 v_fma only", w, out, cyc);
        run<0, 1, true, false>("MFMA + 1 ds_read_b128", w, out, cyc);
        run<0, 2, true, false>("MFMA + 2 ds_read_b128", w, out, cyc);
        run<0, 1, false, false>("1 ds_read_b128 only", w, out, cyc);
        run<0, 2, false, false>("2 ds_read_b128 only", w, out, cyc);
        run<12, 1, true, false>("MFMA + 12 v_fma + 1 ds_read_b128", w, out, cyc);
        run<8, 1, true, false>("MFMA + 8 v_fma + 1 ds_read_b128", w, out, cyc);
        run<4, 1, true, false>("MFMA + 4 v_fma + 1 ds_read_b128", w, out, cyc);
        run<12, 1, false, false>("12 v_fma + 1 ds_read_b128 (no MFMA)", w, out, cyc);
    }
    for (int op = 0; op < 2; ++op) {
        if (op == 0) hipLaunchKernelGGL(kpair<0>, dim3(1), dim3(512), 0, 0, out, cyc, 1.0f);
        else hipLaunchKernelGGL(kpair<1>, dim3(1), dim3(512), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
        printf("MFMA wave + %s wave on one SIMD: %.1f cycles per MFMA, %.2f per vector instruction\n", op ? "v_exp" : "v_fma",
               (double)c[0] / ITER / 4, (double)c[1] / ITER / 32);
    }
    return 0;
}
