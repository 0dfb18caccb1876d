// "Thin" 1x1 convolutions: one side of the contraction has a handful of channels.  These are the first and last layers of
// the PixelDiscriminator (arch/discriminators.py:66-80: 1x1 conv(in -> 64) + LeakyReLU ... 1x1 conv(128 -> 1)) at FULL image
// resolution: 0.5 .. 2 M pixels, arithmetic intensity 0.5 .. 20 FLOP/B - pure HBM streams.  On the implicit-GEMM tiles they
// ran at 0.8 .. 2.6 TFLOP/s, 5x off the bandwidth roofline (a 128-wide tile for 1 output channel, a 32-deep k-tile for 3
// input channels); here every thread streams 16 bytes of the wide side and the thin side lives in registers / LDS.
//
//   forward,  K <= 4   (128 -> 1)  : y[m][k]  = act(sum_c x[m][c] * w[k][c] + b[k])        read x once
//   forward,  C <= 32  (3|20|21 -> 64) : same, the weights in LDS as [c][K]                   write y once
//   data gradient, K <= 4 (d/dx of 128 -> 1): dx[m][c] = sum_k dy[m][k] * wt[c][k]            write dx once
// (the weight gradients of the same layers are thin_wgrad_kernel in conv_wgrad.hip.)
// Arithmetic: fp32 FMA chains.  `precision == 1` (bf16 modes) rounds both operands to bfloat16 first, exactly what the
// bf16 contraction of the matrix-core path does, so the two paths stay interchangeable.
#include "common.h"
#include "sscg_internal.h"

namespace {

template <bool ROUND> __device__ __forceinline__ float rnd(float v) { return ROUND ? (float)(__bf16)v : v; }

inline int blocks_for(size_t work, int cap = 16384) {
    size_t b = (work + 255) / 256;
    if (b > (size_t)cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

// ---- forward, K <= 4: LPP lanes per pixel, 8 channels each; the pixel's K dot products are combined by shuffles
template <typename TX, typename TW, typename TY, int K, bool ROUND>
__global__ __launch_bounds__(256) void thin_fwd_smallk_kernel(const TX* __restrict__ x, const TW* __restrict__ w, const float* __restrict__ bias,
                                                               TY* __restrict__ y, size_t M, int C, int act, float slope) {
    const int lpp = C / 8;                         // lanes per pixel (power of two, <= 64)
    const int sub = threadIdx.x % lpp;
    const int ppb = 256 / lpp;                     // pixels per block and pass
    float wr[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ld8<TW>(w + (size_t)k * C + sub * 8, wr[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[k][e] = rnd<ROUND>(wr[k][e]);
    }
    for (size_t m = (size_t)blockIdx.x * ppb + threadIdx.x / lpp; m < M; m += (size_t)gridDim.x * ppb) {
        float xv[8];
        ld8<TX>(x + m * C + sub * 8, xv);
        float s[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(rnd<ROUND>(xv[e]), wr[k][e], a);
            s[k] = a;
        }
        for (int o = lpp >> 1; o > 0; o >>= 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] += __shfl_xor(s[k], o, 64);
        }
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) st1<TY>(y + m * K + k, sscg_act(s[k] + (bias ? bias[k] : 0.f), act, slope));
        }
    }
}

// ---- forward, C <= 32: a thread owns 8 output channels of PPT pixels; weights in LDS as [c][K] (two 16-byte reads per c)
template <typename TX, typename TY, bool ROUND>
__global__ __launch_bounds__(256) void thin_fwd_smallc_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               TY* __restrict__ y, size_t M, int C, int K, int act, float slope) {
    constexpr int PPT = 4;
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    float* wl = reinterpret_cast<float*>(sm_raw);                    // [C][K]
    for (int i = threadIdx.x; i < C * K; i += 256) {
        const int k = i / C, c = i - k * C;                          // w is [K][C]
        wl[c * K + k] = rnd<ROUND>(w[i]);
    }
    __syncthreads();
    const int kg = K / 8;                                            // channel groups per pixel
    const int g = threadIdx.x % kg;
    const int ppb = (256 / kg) * PPT;                                // pixels per block and pass
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias ? bias[g * 8 + e] : 0.f;
    for (size_t m0 = (size_t)blockIdx.x * ppb + (threadIdx.x / kg) * PPT; m0 < M; m0 += (size_t)gridDim.x * ppb) {
        float acc[PPT][8];
#pragma unroll
        for (int p = 0; p < PPT; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = 0.f;
        for (int c = 0; c < C; ++c) {
            float wv[8];
            ld4<float>(wl + c * K + g * 8, wv);
            ld4<float>(wl + c * K + g * 8 + 4, wv + 4);
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                const size_t m = m0 + p < M ? m0 + p : M - 1;
                const float xv = rnd<ROUND>(ld1<TX>(x + m * C + c));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[p][e] = fmaf(xv, wv[e], acc[p][e]);
            }
        }
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            if (m0 + p >= M) break;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = sscg_act(acc[p][e] + bv[e], act, slope);
            st8<TY>(y + (m0 + p) * K + g * 8, o);
        }
    }
}

// ---- data gradient, K <= 4: a thread owns 8 channels of a pixel
template <typename TDY, typename TDX, int K, bool ROUND>
__global__ __launch_bounds__(256) void thin_dgrad_smallk_kernel(const TDY* __restrict__ dy, const float* __restrict__ wt, TDX* __restrict__ dx,
                                                                 size_t M, int C) {
    const int cg = C / 8;
    const int g = threadIdx.x % cg;
    const int ppb = 256 / cg;
    float wr[K][8];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[k][e] = rnd<ROUND>(wt[(size_t)(g * 8 + e) * K + k]);      // wt is [C][K]
    for (size_t m = (size_t)blockIdx.x * ppb + threadIdx.x / cg; m < M; m += (size_t)gridDim.x * ppb) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float gk = rnd<ROUND>(ld1<TDY>(dy + m * K + k));
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(gk, wr[k][e], o[e]);
        }
        st8<TDX>(dx + m * C + g * 8, o);
    }
}

bool pointwise(const sscg_conv_desc* d) { return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->dil == 1; }

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

bool sscg_thin1x1_fwd_applies(const sscg_conv_desc* d) {
    if (!pointwise(d)) return false;
    if (d->K <= 4 && d->C % 8 == 0 && d->C <= 512 && pow2(d->C / 8) && d->x_dtype == d->w_dtype) return true;               // smallk
    if (d->C <= 32 && d->K % 8 == 0 && d->K <= 256 && pow2(d->K / 8) && d->w_dtype == SSCG_F32 && d->x_dtype == SSCG_F32) return true;   // smallc
    return false;
}

bool sscg_thin1x1_dgrad_applies(const sscg_conv_desc* d, const float* bias, int act) {
    return pointwise(d) && d->K <= 4 && d->C % 8 == 0 && d->C / 8 <= 256 && pow2(d->C / 8) &&
           d->w_dtype == SSCG_F32 && d->y_dtype == SSCG_F32 && bias == nullptr && act == SSCG_ACT_NONE;
}

template <typename TX, typename TW, typename TY, bool ROUND>
static int fwd_smallk(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t st) {
    const size_t M = (size_t)d->N * d->H * d->W;
    const int ppb = 256 / (d->C / 8);
    const dim3 g(blocks_for((M + ppb - 1) / ppb * 256)), b(256);
#define SSCG_SMALLK(k) case k: hipLaunchKernelGGL((thin_fwd_smallk_kernel<TX, TW, TY, k, ROUND>), g, b, 0, st, (const TX*)x, (const TW*)w, bias, (TY*)y, M, d->C, d->act, d->slope); break;
    switch (d->K) { SSCG_SMALLK(1) SSCG_SMALLK(2) SSCG_SMALLK(3) SSCG_SMALLK(4) default: return SSCG_ERR_UNSUPPORTED; }
#undef SSCG_SMALLK
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

template <typename TX, typename TY, bool ROUND>
static int fwd_smallc(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t st) {
    const size_t M = (size_t)d->N * d->H * d->W;
    const int ppb = (256 / (d->K / 8)) * 4;
    const size_t smem = (size_t)d->C * d->K * sizeof(float);
    hipLaunchKernelGGL((thin_fwd_smallc_kernel<TX, TY, ROUND>), dim3(blocks_for((M + ppb - 1) / ppb * 256, 4096)), dim3(256), smem, st, (const TX*)x,
                       (const float*)w, bias, (TY*)y, M, d->C, d->K, d->act, d->slope);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

int sscg_thin1x1_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t st) {
    const bool ybf = d->y_dtype == SSCG_BF16;
    const bool round = d->precision == 1;
    if (d->K <= 4) {
        if (d->x_dtype == SSCG_BF16)      // bf16 activations and bf16 weight operand: already rounded
            return ybf ? fwd_smallk<__bf16, __bf16, __bf16, false>(d, x, w, bias, y, st) : fwd_smallk<__bf16, __bf16, float, false>(d, x, w, bias, y, st);
        if (round) return ybf ? fwd_smallk<float, float, __bf16, true>(d, x, w, bias, y, st) : fwd_smallk<float, float, float, true>(d, x, w, bias, y, st);
        return ybf ? fwd_smallk<float, float, __bf16, false>(d, x, w, bias, y, st) : fwd_smallk<float, float, float, false>(d, x, w, bias, y, st);
    }
    if (round) return ybf ? fwd_smallc<float, __bf16, true>(d, x, w, bias, y, st) : fwd_smallc<float, float, true>(d, x, w, bias, y, st);
    return ybf ? fwd_smallc<float, __bf16, false>(d, x, w, bias, y, st) : fwd_smallc<float, float, false>(d, x, w, bias, y, st);
}

template <typename TDX, bool ROUND>
static int dgrad_smallk(const sscg_conv_desc* d, const void* dy, const void* wt, void* dx, hipStream_t st) {
    const size_t M = (size_t)d->N * d->H * d->W;
    const int ppb = 256 / (d->C / 8);
    const dim3 g(blocks_for((M + ppb - 1) / ppb * 256)), b(256);
#define SSCG_SMALLK(k) case k: hipLaunchKernelGGL((thin_dgrad_smallk_kernel<float, TDX, k, ROUND>), g, b, 0, st, (const float*)dy, (const float*)wt, (TDX*)dx, M, d->C); break;
    switch (d->K) { SSCG_SMALLK(1) SSCG_SMALLK(2) SSCG_SMALLK(3) SSCG_SMALLK(4) default: return SSCG_ERR_UNSUPPORTED; }
#undef SSCG_SMALLK
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

int sscg_thin1x1_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, void* dx, hipStream_t st) {
    const bool round = d->precision == 1;
    if (d->x_dtype == SSCG_BF16) return round ? dgrad_smallk<__bf16, true>(d, dy, wt, dx, st) : dgrad_smallk<__bf16, false>(d, dy, wt, dx, st);
    return round ? dgrad_smallk<float, true>(d, dy, wt, dx, st) : dgrad_smallk<float, false>(d, dy, wt, dx, st);
}
