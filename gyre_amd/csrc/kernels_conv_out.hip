// The output convolution: 3x3, stride 1, zero padding 1, C channels -> a handful (UNet conv_out: 320 -> 4 latent channels; VAE
// decoder conv_out: 128 -> 3 colours), written straight to the caller's NCHW buffer.
//
// As a GEMM this problem is M pixels x N = 3 .. 4 x K = 9 C: the tile kernels gave it to the register-staged 64x64 tile, whose
// workgroup walks K = 2880 in 45 dependent gather steps for 1/16 of a useful tile - 52 us at batch 16 and still 40 us at batch 2
// (rocprofv3 timeline, round 5), i.e. the launch is one serial chain whatever the batch.  Here a workgroup owns an 8 x 8 pixel
// tile: the 10 x 10 halo of a 64-channel chunk is staged in LDS once (each input pixel is read once per tile instead of once per
// tap), all weights (N x 9 x C bf16, 23 KB for the UNet) sit in LDS for the life of the workgroup, and a wave multiplies its 16
// pixels against them with v_mfma_f32_16x16x32_bf16 - weights as the FIRST operand (16 rows, N of them real), pixels second, so
// the lanes of the first quarter-wave hold the N outputs of their pixel and consecutive lanes write consecutive pixels of an
// NCHW plane.  The next chunk's halo is requested before the current one is multiplied.  Summation order per output: channel
// chunks of 64 ascending, taps row-major inside a chunk, two 32-channel MFMA steps per tap - fixed, so the result does not depend
// on the batch or on the tile a pixel falls into (reference property tests/batch_independance.py:15-27).
//
// Replaces the cuDNN convolution behind `self.conv_out` of the third-party UNet / VAE decoder the reference calls at
// gyre/pipeline/unet/core.py:274 and unified_pipeline.py:1531.
#include "kernels.h"

namespace {
constexpr int CO_T = 8;                    // tile edge (output pixels)
constexpr int CO_HP = CO_T + 2;            // halo edge
constexpr int CO_PIXB = 144;               // bytes per halo pixel in LDS: 64 channels + 16 (bank spread between neighbouring pixels)
constexpr int CO_HALO_BYTES = CO_HP * CO_HP * CO_PIXB;
}  // namespace

__global__ __launch_bounds__(256) void k_conv_out(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                  const float* __restrict__ bias, void* __restrict__ out, int out_dtype,
                                                  int H, int Wd, int C, int O, int w_row_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                       // [10][10][144 B]
    char* wl = smem + CO_HALO_BYTES;         // [O][w_row_bytes]: 9 x C bf16 per output channel (+ 16 bytes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int x0 = blockIdx.x * CO_T, y0 = blockIdx.y * CO_T, b = blockIdx.z;
    const bf16_t* xb = x + (size_t)b * H * Wd * C;

    // weights -> LDS (16-byte pieces; a row of W is 9 * C bf16, C a multiple of 64)
    const int wpieces = 9 * C / 8;
    for (int i = tid; i < O * wpieces; i += 256) {
        const int o = i / wpieces, pc = i - o * wpieces;
        *(uint4*)(wl + o * w_row_bytes + pc * 16) = *(const uint4*)(W + (size_t)o * 9 * C + pc * 8);
    }
    // halo requests of a chunk: 100 pixels x 8 pieces = 800 = 3.125 per thread
    uint4 hv[4];
    auto request = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i < CO_HP * CO_HP * 8) {
                const int pix = i >> 3, pc = i & 7;
                const int py = pix / CO_HP, px = pix - py * CO_HP;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)Wd)
                    v = *(const uint4*)(xb + ((size_t)gy * Wd + gx) * C + c0 + pc * 8);
            }
            hv[j] = v;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 256;
            if (i < CO_HP * CO_HP * 8) *(uint4*)(halo + (i >> 3) * CO_PIXB + (i & 7) * 16) = hv[j];
        }
    };
    // this lane's pixel of the wave's 16 (two tile rows of eight)
    const int ty = 2 * wave + (fr >> 3), tx = fr & 7;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const bf16x8_t zero8 = __builtin_bit_cast(bf16x8_t, make_uint4(0, 0, 0, 0));
    request(0);
    for (int c0 = 0; c0 < C; c0 += 64) {
        __syncthreads();                    // the previous chunk's fragments have been read
        stage();
        __syncthreads();
        if (c0 + 64 < C) request(c0 + 64);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const char* xp = halo + ((ty + ky) * CO_HP + tx + kx) * CO_PIXB + fq * 16;
            const char* wp = wl + fr * w_row_bytes + (tap * C + c0) * 2 + fq * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8_t xf = *(const bf16x8_t*)(xp + ks * 64);
                const bf16x8_t wf = fr < O ? *(const bf16x8_t*)(wp + ks * 64) : zero8;
                acc = GYRE_MFMA_16x16x32(wf, xf, acc, 0, 0, 0);
            }
        }
    }
    // lane (fr, fq) holds outputs o = 4 fq + e of pixel fr
    const int gy = y0 + ty, gx = x0 + tx;
    if (gy < H && gx < Wd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = 4 * fq + e;
            if (o < O) store_from_f32(out, out_dtype, (((size_t)b * O + o) * H + gy) * Wd + gx, acc[e] + (bias ? bias[o] : 0.f));
        }
    }
}

bool conv_out_supports(int C, int O) {
    return C >= 64 && C % 64 == 0 && O >= 1 && O <= 16 && (size_t)CO_HALO_BYTES + (size_t)O * (9 * C * 2 + 16) <= 64 * 1024;
}

int launch_conv_out(hipStream_t st, const bf16_t* x, int B, int H, int Wd, int C, const bf16_t* W, const float* bias, int O,
                    void* out, int out_dtype) {
    if (!conv_out_supports(C, O)) GYRE_FAIL(-6, "conv_out: C must be a multiple of 64 and the weights must fit LDS (conv_out_supports)");
    if (B < 1 || H < 1 || Wd < 1) GYRE_FAIL(-1, "conv_out: empty image");
    const int w_row_bytes = 9 * C * 2 + 16;
    const size_t lds = (size_t)CO_HALO_BYTES + (size_t)O * w_row_bytes;
    const double px = (double)B * H * Wd;
    GyreProfScope prof_(KC_G8_X2, st, 2.0 * px * 9.0 * C * O, px * C * 2.0 + px * O * 4.0 + 9.0 * C * O * 2.0);
    hipLaunchKernelGGL(k_conv_out, dim3((Wd + CO_T - 1) / CO_T, (H + CO_T - 1) / CO_T, B), dim3(256), lds, st, x, W, bias, out, out_dtype,
                       H, Wd, C, O, w_row_bytes);
    GYRE_LAUNCH_CHECK();
    return 0;
}
