// Dev microbenchmark: issue cost (cycles per wave64 instruction, one wave on one SIMD) of the vector ops the attention / epilogue
// loops are made of.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define ITER 256
template <int OP>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 0.001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 5) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 6) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a[i]));
                if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
                if (OP == 9) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
__global__ void kpk(float* out, long long* cyc, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a[i]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// waves 0-3: back-to-back MFMAs on 4 independent accumulators; waves 4-7 (same SIMDs): VALU chains of OP
template <int OP, bool WITH_VALU>
__global__ void kco(float* out, long long* cyc, float seed) {
    const int wave = threadIdx.x >> 6;
    long long t0, t1;
    if (wave < 4) {
        f32x4 acc[4] = {};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    } else if (WITH_VALU) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 0.001f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
                    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
        out[threadIdx.x] = s;
        if (threadIdx.x == 256) cyc[1] = t1 - t0;
    }
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_mul_f32", "v_add_u32", "", "v_cndmask_b32", "v_mov_b32"};
    for (int waves = 1; waves <= 2; ++waves) {
        printf("-- %d wave(s) on the SIMD (block of %d threads: waves share SIMDs when > 4 waves)\n", waves, waves == 1 ? 64 : 512);
#define RUN(OP) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(k<OP>, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize(); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-20s %6.2f counter ticks per instruction (wave 0)\n", names[OP], (double)c / (ITER * REP)); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(8) RUN(9)
#define RUNPK(OP, NAME) { hipLaunchKernelGGL(kpk<OP>, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(kpk<OP>, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize(); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-20s %6.2f\n", NAME, (double)c / (ITER * REP)); }
        RUNPK(0, "v_pk_fma_f32") RUNPK(1, "v_pk_mul_f32")
    }
    {
        long long c[2];
        hipLaunchKernelGGL((kco<0, false>), dim3(1), dim3(512), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize();
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); printf("MFMA 16x16x32 alone (one wave per SIMD): %.2f ticks per MFMA\n", (double)c[0] / (ITER * REP));
        hipLaunchKernelGGL((kco<0, true>), dim3(1), dim3(512), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize();
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); printf("MFMA wave + v_fma wave on the same SIMD: %.2f ticks per MFMA, %.2f per v_fma\n", (double)c[0] / (ITER * REP), (double)c[1] / (ITER * REP));
        hipLaunchKernelGGL((kco<1, true>), dim3(1), dim3(512), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize();
        hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost); printf("MFMA wave + v_exp wave on the same SIMD: %.2f ticks per MFMA, %.2f per v_exp\n", (double)c[0] / (ITER * REP), (double)c[1] / (ITER * REP));
    }
    for (int th = 1024; th <= 1024; th += 512) {   // 4 waves per SIMD: throughput
        printf("-- %d threads (4 waves per SIMD): ticks per instruction per wave; SIMD throughput = that / 4\n", th);
#define RUN4(OP) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(th), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(k<OP>, dim3(1), dim3(th), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize(); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-20s %6.2f\n", names[OP], (double)c / (ITER * REP)); }
        RUN4(0) RUN4(1) RUN4(2) RUN4(3) RUN4(4) RUN4(5)
        { hipLaunchKernelGGL(kpk<0>, dim3(1), dim3(th), 0, 0, out, cyc, 1.0f); hipDeviceSynchronize(); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-20s %6.2f\n", "v_pk_fma_f32", (double)c / (ITER * REP)); }
    }
    // counter frequency: time a long kernel with events
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("fma kernel: %.3f us per launch wall, %lld ticks inside -> counter ~%.0f MHz if the launch were all kernel\n", ms * 1e3 / 200, c, c / (ms * 1e3 / 200));
    return 0;
}
