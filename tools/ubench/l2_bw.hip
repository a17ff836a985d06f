// Dev microbenchmark (round 5): how many bytes per second does ONE CU get out of L2 / the Infinity Cache / HBM, by load path, waves per CU
// and requests in flight per wave?  Settles DESIGN.md 4c's "the CU's load path moves ~37 GB/s whatever issues the requests".
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/l2_bw.hip -o /tmp/l2_bw && /tmp/l2_bw > profiles/r05_l2_bw.txt
// Every wave keeps INFL 1-KiB requests (64 lanes x 16 B, one wave instruction) in flight in steady state:
//   path 0  DMA : global_load_lds_dwordx4 into an LDS ring, counted s_waitcnt vmcnt(INFL-1) after every request
//   path 1  REG : global_load_dwordx4 -> VGPR -> ds_write_b128 (INFL register quads rotate; hipcc counts the waits)
//   path 2  REGX: global_load_dwordx4 -> VGPR, consumed by a v_xor (no LDS write): the bare vector-memory path
//   path 3  DMA with nt (aux = 2) on the request
// Residency of the source:
//   L2  : the workgroups of one XCD (blockIdx % 8) sweep the same 2 MiB window again and again (4 MiB L2 per XCD)
//   MALL: all workgroups sweep one 96 MiB buffer from staggered offsets (256 MiB Infinity Cache, far beyond the 32 MiB of L2)
//   HBM : every workgroup streams its own slice of a 6 GiB buffer once
// Output: one line per configuration - GB/s per CU and TB/s for the chip - at 256 CUs x WG workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int PATH, int INFL>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t window, size_t wg_stride, int mode_xcd, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // window base of this workgroup and the offset it starts at
    const size_t wbase = mode_xcd ? (size_t)(blockIdx.x & 7) * window : 0;
    size_t off = ((size_t)(blockIdx.x >> (mode_xcd ? 3 : 0)) * wg_stride + (size_t)wave * 1024) % window;
    const char* base = src + wbase + lane * 16;
    char* ring = smem + wave * (INFL * 1024);
    unsigned acc = 0;
    if constexpr (PATH == 0 || PATH == 3) {
#pragma unroll
        for (int i = 0; i < INFL; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)(ring + i * 1024), 16, 0, PATH == 3 ? 2 : 0);
            off += 4096; if (off >= window) off -= window;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < INFL; ++i) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
                __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)(ring + i * 1024), 16, 0, PATH == 3 ? 2 : 0);
                off += 4096; if (off >= window) off -= window;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = ((unsigned*)ring)[lane];
    } else {
        uint4 r[INFL];
#pragma unroll
        for (int i = 0; i < INFL; ++i) {
            r[i] = *(const uint4*)(base + off);
            off += 4096; if (off >= window) off -= window;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < INFL; ++i) {
                if constexpr (PATH == 1) *(uint4*)(ring + i * 1024 + lane * 16) = r[i];
                else acc ^= r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
                r[i] = *(const uint4*)(base + off);
                off += 4096; if (off >= window) off -= window;
            }
        }
#pragma unroll
        for (int i = 0; i < INFL; ++i) acc ^= r[i].x;
        if constexpr (PATH == 1) { __syncthreads(); acc ^= ((unsigned*)ring)[lane]; }
    }
    if (acc == 0x12345677u) sink[0] = acc;
}

// GEMM-operand pattern: a request = 8 matrix rows x 128 B (row stride `rs` bytes), a wave walks along the row (k steps) and then takes
// the next 8-row block of its workgroup's 32-row panel ... : what the tile kernels' LDS-DMA requests look like to L2.
template <int INFL, int R>
__global__ __launch_bounds__(256) void ks(const char* __restrict__ src, size_t window, unsigned rs, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wbase = (size_t)(blockIdx.x & 7) * window;
    constexpr int LPR = 64 / R, BPR = 1024 / R;                 // lanes and bytes per row of one request
    const unsigned rows = (unsigned)(window / rs), ksteps = rs / BPR;
    unsigned row0 = (((blockIdx.x >> 3) * 4 + wave) * R) % rows, k = 0;
    const char* base = src + wbase + (size_t)(lane / LPR) * rs + (lane % LPR) * 16;
    char* ring = smem + wave * (INFL * 1024);
    auto next = [&]() { if (++k == ksteps) { k = 0; row0 += R * 4 * 32; if (row0 >= rows) row0 -= rows; } };
#pragma unroll
    for (int i = 0; i < INFL; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)row0 * rs + k * BPR), (lds_void*)(ring + i * 1024), 16, 0, 0);
        next();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < INFL; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)row0 * rs + k * BPR), (lds_void*)(ring + i * 1024), 16, 0, 0);
            next();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (((unsigned*)ring)[lane] == 0x12345677u) sink[0] = 1;
}
template <int INFL, int R = 8>
static double run_strided(const char* src, unsigned rs, int wg_per_cu, unsigned* sink) {
    const int grid = 256 * wg_per_cu;
    const size_t window = (2ull << 20) / rs * rs;
    const int iters = (int)((8ull << 20) / (1024 * (size_t)INFL));
    const size_t lds = 4 * INFL * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((ks<INFL, R>), dim3(grid), dim3(256), lds, 0, src, window, rs, iters, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)grid * 4 * ((double)iters + 1) * INFL * 1024 / (best * 1e-3) / 1e9 / 256;
}

struct Res { double gbs_cu, tbs; };
template <int PATH, int INFL>
static Res run(const char* src, size_t window, size_t wg_stride, int mode_xcd, int wg_per_cu, size_t bytes_per_wave, unsigned* sink) {
    const int grid = 256 * wg_per_cu;
    const int iters = (int)(bytes_per_wave / (1024 * (size_t)INFL));
    const size_t lds = 4 * INFL * 1024;
    hipFuncSetAttribute((const void*)k<PATH, INFL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<PATH, INFL>), dim3(grid), dim3(256), lds, 0, src, window, wg_stride, mode_xcd, iters, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)grid * 4 * ((double)iters + 1) * INFL * 1024;
    Res r; r.tbs = bytes / (best * 1e-3) / 1e12; r.gbs_cu = bytes / (best * 1e-3) / 1e9 / 256;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

template <int PATH>
static void sweep(const char* pname, const char* rname, const char* src, size_t window, size_t wg_stride_of(int), int mode_xcd, size_t bytes_per_wave, unsigned* sink) {
    for (int wg : {1, 2, 4}) {
        printf("%-5s %-4s waves/CU=%2d :", pname, rname, 4 * wg);
        Res r;
        r = run<PATH, 2>(src, window, wg_stride_of(wg), mode_xcd, wg, bytes_per_wave, sink);  printf("  infl2 %6.1f GB/s/CU (%5.2f TB/s)", r.gbs_cu, r.tbs);
        r = run<PATH, 4>(src, window, wg_stride_of(wg), mode_xcd, wg, bytes_per_wave, sink);  printf("  infl4 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
        r = run<PATH, 8>(src, window, wg_stride_of(wg), mode_xcd, wg, bytes_per_wave, sink);  printf("  infl8 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
        if (wg <= 2 || PATH == 2) { r = run<PATH, 16>(src, window, wg_stride_of(wg), mode_xcd, wg, bytes_per_wave, sink); printf("  infl16 %6.1f (%5.2f)", r.gbs_cu, r.tbs); }
        printf("\n"); fflush(stdout);
    }
}

static size_t g_total;
static size_t stride_l2(int) { return 65536 + 4096; }                 // workgroups of an XCD start 68 KB apart inside the 2 MiB window
static size_t stride_mall(int) { return (96ull << 20) / 1024 + 4096; }
static size_t stride_hbm(int wg) { return g_total / (256 * (size_t)wg) / 4096 * 4096; }

int main() {
    const size_t total = 6ull << 30;
    g_total = total;
    char* buf = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, total);
    unsigned* sink; hipMalloc(&sink, 4);
    hipDeviceSynchronize();
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("# l2_bw: 1 KiB wave requests, INFL in flight per wave, 256-thread workgroups, grid = 256 CUs x WG; max clock %d MHz\n", clk / 1000);
    printf("# per-CU numbers assume all 256 CUs busy (grid is a multiple of 256; blockIdx %% 8 = XCD)\n");
    const size_t per_wave_l2 = 8ull << 20, per_wave_mall = 4ull << 20;
    const bool only_strided = getenv("ONLY_STRIDED") != nullptr;
    if (!only_strided) {
    sweep<0>("DMA", "L2", buf, 2ull << 20, stride_l2, 1, per_wave_l2, sink);
    sweep<3>("DMAnt", "L2", buf, 2ull << 20, stride_l2, 1, per_wave_l2, sink);
    sweep<1>("REG", "L2", buf, 2ull << 20, stride_l2, 1, per_wave_l2, sink);
    sweep<2>("REGX", "L2", buf, 2ull << 20, stride_l2, 1, per_wave_l2, sink);
    sweep<0>("DMA", "MALL", buf, 96ull << 20, stride_mall, 0, per_wave_mall, sink);
    sweep<2>("REGX", "MALL", buf, 96ull << 20, stride_mall, 0, per_wave_mall, sink);
    }
    printf("# strided: a request = 8 rows x 128 B at row stride rs (the GEMM operand pattern), L2-resident 2 MiB per XCD, DMA path\n");
    for (unsigned rs : {128u, 640u, 1280u, 2560u, 5120u, 1536u, 4096u, 8192u, 2688u}) {
        printf("DMA   L2   stride %5u B:", rs);
        for (int wg : {1, 2}) {
            printf("  waves/CU=%d: infl2 %6.1f  infl4 %6.1f  infl8 %6.1f GB/s/CU |", 4 * wg, run_strided<2>(buf, rs, wg, sink), run_strided<4>(buf, rs, wg, sink), run_strided<8>(buf, rs, wg, sink));
        }
        printf("\n"); fflush(stdout);
    }
    printf("# strided, fewer rows per request: R rows x (1024 / R) contiguous bytes\n");
    for (unsigned rs : {2560u, 5120u, 11520u, 23040u}) {
        printf("DMA   L2   stride %5u B, 4 waves/CU, infl2 / infl4 / infl8:  R=8 %6.1f %6.1f %6.1f | R=4 %6.1f %6.1f %6.1f | R=2 %6.1f %6.1f %6.1f | R=1 %6.1f %6.1f %6.1f GB/s/CU\n", rs,
               run_strided<2, 8>(buf, rs, 1, sink), run_strided<4, 8>(buf, rs, 1, sink), run_strided<8, 8>(buf, rs, 1, sink),
               run_strided<2, 4>(buf, rs, 1, sink), run_strided<4, 4>(buf, rs, 1, sink), run_strided<8, 4>(buf, rs, 1, sink),
               run_strided<2, 2>(buf, rs, 1, sink), run_strided<4, 2>(buf, rs, 1, sink), run_strided<8, 2>(buf, rs, 1, sink),
               run_strided<2, 1>(buf, rs, 1, sink), run_strided<4, 1>(buf, rs, 1, sink), run_strided<8, 1>(buf, rs, 1, sink));
        fflush(stdout);
    }
    // HBM: every wave streams bytes_per_wave = its slice (6 GiB / waves)
    for (int pass = 0; pass < (only_strided ? 0 : 1); ++pass) {
        for (int wg : {1, 2, 4}) {
            const size_t per_wave = total / (256 * (size_t)wg) / 4 / 65536 * 65536 - 65536;
            printf("%-5s %-4s waves/CU=%2d :", "DMA", "HBM", 4 * wg);
            Res r = run<0, 4>(buf, total, stride_hbm(wg), 0, wg, per_wave, sink); printf("  infl4 %6.1f GB/s/CU (%5.2f TB/s)", r.gbs_cu, r.tbs);
            r = run<0, 8>(buf, total, stride_hbm(wg), 0, wg, per_wave, sink); printf("  infl8 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
            r = run<0, 16>(buf, total, stride_hbm(wg), 0, wg, per_wave, sink); printf("  infl16 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
            r = run<3, 8>(buf, total, stride_hbm(wg), 0, wg, per_wave, sink); printf("  nt infl8 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
            r = run<2, 8>(buf, total, stride_hbm(wg), 0, wg, per_wave, sink); printf("  REGX infl8 %6.1f (%5.2f)", r.gbs_cu, r.tbs);
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
