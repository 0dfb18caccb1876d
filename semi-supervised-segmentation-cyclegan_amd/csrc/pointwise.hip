// HBM-bound pointwise / pooling / resize / layout kernels (NHWC fp32).  Reference call sites are
// listed per entry point in include/sscg.h.
#include "common.h"
#include "sscg_internal.h"

namespace {

inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        st1<T>(y + i, sscg_act(ld1<T>(x + i), act, slope));
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, size_t n,
                               int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float g = ld1<T>(dy + i), v = ld1<T>(y + i);
        if (act == SSCG_ACT_RELU) g = v > 0.f ? g : 0.f;
        else if (act == SSCG_ACT_LRELU) g = v > 0.f ? g : g * slope;
        else if (act == SSCG_ACT_TANH) g = g * (1.f - v * v);
        st1<T>(dx + i, g);
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        st1<T>(y + i, ld1<T>(a + i) + ld1<T>(b + i));
}

// 16 bytes per lane (4 fp32 / 8 bf16), two independent vectors in flight per thread
template <typename T, int V>
__global__ __launch_bounds__(256) void addv_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t nv) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    auto one = [&](size_t k, float o[V]) {
        float u[V], w[V];
        if constexpr (V == 8) { ld8<T>(a + k * V, u); ld8<T>(b + k * V, w); } else { ld4<T>(a + k * V, u); ld4<T>(b + k * V, w); }
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = u[e] + w[e];
    };
    auto put = [&](size_t k, const float o[V]) {
        if constexpr (V == 8) st8<T>(y + k * V, o); else st4<T>(y + k * V, o);
    };
    for (; i + stride < nv; i += 2 * stride) {
        float o0[V], o1[V];
        one(i, o0);
        one(i + stride, o1);
        put(i, o0);
        put(i + stride, o1);
    }
    if (i < nv) {
        float o0[V];
        one(i, o0);
        put(i, o0);
    }
}

__global__ void fill_kernel(float* __restrict__ x, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = v;
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st1<D>(dst + i, ld1<S>(src + i));
}

__device__ __forceinline__ uint32_t mix64to32(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, float p, float scale,
                               uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t r = mix64to32(seed * 0x2545f4914f6cdd1dull + (uint64_t)i);
        float u = (float)(r >> 8) * (1.0f / 16777216.0f);
        st1<T>(y + i, u >= p ? ld1<T>(x + i) * scale : 0.f);
    }
}

// y = x + sigma * x * n,  n ~ N(0,1): two hashes per element -> Box-Muller (utils.GaussianNoise, relative noise)
__global__ void gauss_noise_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float sigma, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t r1 = mix64to32(seed * 0x2545f4914f6cdd1dull + 2 * (uint64_t)i);
        const uint32_t r2 = mix64to32(seed * 0x2545f4914f6cdd1dull + 2 * (uint64_t)i + 1);
        const float u1 = ((float)(r1 >> 8) + 1.0f) * (1.0f / 16777216.0f);    // (0, 1]
        const float u2 = (float)(r2 >> 8) * (1.0f / 16777216.0f);             // [0, 1)
        const float z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
        const float v = x[i];
        y[i] = v + sigma * v * z;
    }
}

// one thread per (n, oy, ox, c)
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N,
                                   int H, int W, int C, int P, int Q) {
    size_t total = (size_t)N * P * Q * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ox = (int)(t % Q); t /= Q;
        int oy = (int)(t % P);
        int n = (int)(t / P);
        float best = -INFINITY;
        int bi = 0;
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v = ld1<T>(x + ((size_t)(n * H + iy) * W + ix) * C + c);
                if (first || v > best) { best = v; bi = ky * 3 + kx; first = false; }
            }
        }
        st1<T>(y + i, best);
        idx[i] = (uint8_t)bi;
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx,
                                   int N, int H, int W, int C, int P, int Q) {
    size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ix = (int)(t % W); t /= W;
        int iy = (int)(t % H);
        int n = (int)(t / H);
        float s = 0.f;
        int oy_lo = iy / 2, oy_hi = (iy + 1) / 2;  // oy*2-1 <= iy <= oy*2+1
        int ox_lo = ix / 2, ox_hi = (ix + 1) / 2;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= P) continue;
            int ky = iy - (oy * 2 - 1);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= Q) continue;
                int kx = ix - (ox * 2 - 1);
                size_t o = ((size_t)(n * P + oy) * Q + ox) * C + c;
                if (idx[o] == ky * 3 + kx) s += ld1<T>(dy + o);
            }
        }
        st1<T>(dx + i, s);
    }
}

// nn.MaxPool2d(kernel_size=2, stride=2) (floor mode; torchvision VGG16 features[4,9,16], utils.py:147-164): one thread per output
// element, first maximum wins (the window index 2 * ky + kx goes to idx for the backward)
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C,
                                    int P, int Q) {
    size_t total = (size_t)N * P * Q * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ox = (int)(t % Q); t /= Q;
        int oy = (int)(t % P);
        int n = (int)(t / P);
        const T* base = x + ((size_t)(n * H + oy * 2) * W + ox * 2) * C + c;
        float best = ld1<T>(base);
        int bi = 0;
        float v = ld1<T>(base + C);
        if (v > best) { best = v; bi = 1; }
        v = ld1<T>(base + (size_t)W * C);
        if (v > best) { best = v; bi = 2; }
        v = ld1<T>(base + (size_t)W * C + C);
        if (v > best) { best = v; bi = 3; }
        st1<T>(y + i, best);
        idx[i] = (uint8_t)bi;
    }
}

// windows do not overlap: every input element belongs to at most one output (rows / columns past 2P, 2Q get zero)
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N, int H, int W,
                                    int C, int P, int Q) {
    size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ix = (int)(t % W); t /= W;
        int iy = (int)(t % H);
        int n = (int)(t / H);
        int oy = iy >> 1, ox = ix >> 1;
        float g = 0.f;
        if (oy < P && ox < Q) {
            size_t o = ((size_t)(n * P + oy) * Q + ox) * C + c;
            if (idx[o] == (uint8_t)((iy & 1) * 2 + (ix & 1))) g = ld1<T>(dy + o);
        }
        st1<T>(dx + i, g);
    }
}

// bilinear, align_corners=True (torch upsample_bilinear2d arithmetic: src = scale * dst in fp32)
__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                    int OH, int OW, float sh, float sw) {
    size_t total = (size_t)N * OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ox = (int)(t % OW); t /= OW;
        int oy = (int)(t % OH);
        int n = (int)(t / OH);
        float fy = sh * oy, fx = sw * ox;
        int y0 = (int)fy, x0 = (int)fx;
        int yp = y0 < H - 1 ? 1 : 0, xp = x0 < W - 1 ? 1 : 0;
        float ly = fy - y0, lx = fx - x0;
        float hy = 1.f - ly, hx = 1.f - lx;
        const float* b = x + (size_t)n * H * W * C + c;
        float v00 = b[((size_t)y0 * W + x0) * C], v01 = b[((size_t)y0 * W + x0 + xp) * C];
        float v10 = b[((size_t)(y0 + yp) * W + x0) * C], v11 = b[((size_t)(y0 + yp) * W + x0 + xp) * C];
        y[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

// gather form of the adjoint: one thread per (n, iy, ix, c) sums the output pixels that touch it
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C,
                                    int OH, int OW, float sh, float sw, float inv_sh, float inv_sw) {
    size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ix = (int)(t % W); t /= W;
        int iy = (int)(t % H);
        int n = (int)(t / H);
        int oy_lo = (int)floorf((iy - 1) * inv_sh) - 1, oy_hi = (int)ceilf((iy + 1) * inv_sh) + 1;
        int ox_lo = (int)floorf((ix - 1) * inv_sw) - 1, ox_hi = (int)ceilf((ix + 1) * inv_sw) + 1;
        oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
        oy_hi = min(oy_hi, OH - 1); ox_hi = min(ox_hi, OW - 1);
        float s = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            float fy = sh * oy;
            int y0 = (int)fy;
            int yp = y0 < H - 1 ? 1 : 0;
            float ly = fy - y0;
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y0 + yp == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float fx = sw * ox;
                int x0 = (int)fx;
                int xp = x0 < W - 1 ? 1 : 0;
                float lx = fx - x0;
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x0 + xp == ix) wx += lx;
                if (wx == 0.f) continue;
                s += wy * wx * dy[((size_t)(n * OH + oy) * OW + ox) * C + c];
            }
        }
        dx[i] = s;
    }
}

template <typename T>
__global__ void reflect_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int pad) {
    int OH = H + 2 * pad, OW = W + 2 * pad;
    size_t total = (size_t)N * OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ox = (int)(t % OW); t /= OW;
        int oy = (int)(t % OH);
        int n = (int)(t / OH);
        int sy = oy - pad, sx = ox - pad;
        sy = sy < 0 ? -sy : sy; sx = sx < 0 ? -sx : sx;
        sy = sy >= H ? 2 * (H - 1) - sy : sy;
        sx = sx >= W ? 2 * (W - 1) - sx : sx;
        y[i] = x[((size_t)(n * H + sy) * W + sx) * C + c];
    }
}

// adjoint of reflect_pad in gather form: every source pixel sums the (up to 3 x 3) padded positions that mirror onto it
template <typename T>
__global__ void reflect_pad_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int pad) {
    int OH = H + 2 * pad, OW = W + 2 * pad;
    size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c = (int)(i % C);
        size_t t = i / C;
        int ix = (int)(t % W); t /= W;
        int iy = (int)(t % H);
        int n = (int)(t / H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = iy + pad;
        if (iy >= 1 && iy <= pad) ys[ny++] = pad - iy;
        if (iy <= H - 2 && iy >= H - 1 - pad) ys[ny++] = 2 * (H - 1) - iy + pad;
        xs[nx++] = ix + pad;
        if (ix >= 1 && ix <= pad) xs[nx++] = pad - ix;
        if (ix <= W - 2 && ix >= W - 1 - pad) xs[nx++] = 2 * (W - 1) - ix + pad;
        float s = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) s += ld1<T>(dy + ((size_t)(n * OH + ys[a]) * OW + xs[b]) * C + c);
        st1<T>(dx + i, s);
    }
}

// per image: in [R][Cc] -> out [Cc][R]   (32x32 LDS tiles)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    __shared__ float t[32][33];
    const float* src = in + (size_t)blockIdx.z * R * Cc;
    float* dst = out + (size_t)blockIdx.z * R * Cc;
    int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int rr = r0 + r, cc = c0 + tx;
        t[r][tx] = (rr < R && cc < Cc) ? src[(size_t)rr * Cc + cc] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int cc = c0 + r, rr = r0 + tx;
        if (rr < R && cc < Cc) dst[(size_t)cc * R + rr] = t[tx][r];
    }
}

}  // namespace

extern "C" int sscg_abi_version(void) { return SSCG_ABI_VERSION; }

int g_sscg_dry_run = 0;
extern "C" int sscg_set_dry_run(int on) {
    const int was = g_sscg_dry_run;
    g_sscg_dry_run = on ? 1 : 0;
    return was;
}

#define SSCG_DT_OK(dt) ((dt) == SSCG_F32 || (dt) == SSCG_BF16)
#define BF(p) reinterpret_cast<const __bf16*>(p)
#define BFW(p) reinterpret_cast<__bf16*>(p)
#define FP(p) reinterpret_cast<const float*>(p)
#define FPW(p) reinterpret_cast<float*>(p)

extern "C" int sscg_act_fwd(const void* x, void* y, int dtype, int64_t n, int act, float slope, void* stream) {
    if (!x || !y || n <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(act_fwd_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, BF(x), BFW(y), (size_t)n, act, slope);
    else
        hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, FP(x), FPW(y), (size_t)n, act, slope);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_act_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n, int act, float slope, void* stream) {
    if (!dy || !y || !dx || n <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, BF(dy), BF(y), BFW(dx), (size_t)n, act, slope);
    else
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, FP(dy), FP(y), FPW(dx), (size_t)n, act, slope);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream) {
    if (!a || !b || !y || n <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((size_t)a | (size_t)b | (size_t)y) & 15) == 0;
    if (dtype == SSCG_BF16) {
        if (n % 8 == 0 && al) {
            const size_t nv = (size_t)n / 8;
            hipLaunchKernelGGL((addv_kernel<__bf16, 8>), dim3(ew_blocks((nv + 1) / 2)), dim3(256), 0, st, BF(a), BF(b), BFW(y), nv);
        } else {
            hipLaunchKernelGGL(add_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, st, BF(a), BF(b), BFW(y), (size_t)n);
        }
    } else if (n % 4 == 0 && al) {
        const size_t nv = (size_t)n / 4;
        hipLaunchKernelGGL((addv_kernel<float, 4>), dim3(ew_blocks((nv + 1) / 2)), dim3(256), 0, st, FP(a), FP(b), FPW(y), nv);
    } else {
        hipLaunchKernelGGL(add_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, FP(a), FP(b), FPW(y), (size_t)n);
    }
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    if (!src || !dst || n <= 0 || !SSCG_DT_OK(src_dtype) || !SSCG_DT_OK(dst_dtype)) return SSCG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(ew_blocks(n)), b(256);
    if (src_dtype == SSCG_F32 && dst_dtype == SSCG_BF16) hipLaunchKernelGGL((cast_kernel<float, __bf16>), g, b, 0, st, FP(src), BFW(dst), (size_t)n);
    else if (src_dtype == SSCG_BF16 && dst_dtype == SSCG_F32) hipLaunchKernelGGL((cast_kernel<__bf16, float>), g, b, 0, st, BF(src), FPW(dst), (size_t)n);
    else if (src_dtype == SSCG_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, FP(src), FPW(dst), (size_t)n);
    else hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), g, b, 0, st, BF(src), BFW(dst), (size_t)n);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// dst[r][c] = c < Cs ? src[r][c] : 0 for c < Cd: channel padding (Cd > Cs) or channel slicing (Cd < Cs) of an NHWC tensor
__global__ __launch_bounds__(256) void resize_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int Cs, int Cd) {
    const int gd = (Cd + 3) >> 2;                       // 4-channel groups of a destination row
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * gd) return;
    const size_t r = i / gd;
    const int c0 = (int)(i - r * gd) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (c0 + e < Cs && c0 + e < Cd) ? src[r * Cs + c0 + e] : 0.f;
    if ((Cd & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst + r * Cd + c0) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < Cd) dst[r * Cd + c0 + e] = v[e];
    }
}

extern "C" int sscg_resize_channels(const float* src, float* dst, int64_t rows, int Cs, int Cd, void* stream) {
    if (!src || !dst || rows <= 0 || Cs <= 0 || Cd <= 0) return SSCG_ERR_BAD_ARG;
    const size_t n = (size_t)rows * ((Cd + 3) >> 2);
    hipLaunchKernelGGL(resize_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (size_t)rows, Cs, Cd);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_fill(float* x, int64_t n, float v, void* stream) {
    if (!x || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, v);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_dropout(const void* x, void* y, int dtype, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !y || n <= 0 || p < 0.f || p >= 1.f || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(dropout_kernel<__bf16>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, BF(x), BFW(y), (size_t)n, p, 1.f / (1.f - p), seed);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, FP(x), FPW(y), (size_t)n, p, 1.f / (1.f - p), seed);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_gauss_noise(const float* x, float* y, int64_t n, float sigma, uint64_t seed, void* stream) {
    if (!x || !y || n <= 0 || sigma < 0.f) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(gauss_noise_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n, sigma, seed);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int N, int H, int W, int C, int P, int Q,
                                     void* stream) {
    if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    if ((P - 1) * 2 - 1 >= H || (Q - 1) * 2 - 1 >= W) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * P * Q * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(x), BFW(y), idx, N, H, W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(x), FPW(y), idx, N, H, W, C, P, Q);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int N, int H, int W, int C, int P,
                                     int Q, void* stream) {
    if (!dy || !idx || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * H * W * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(dy), idx, BFW(dx), N, H, W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(dy), idx, FPW(dx), N, H, W, C, P, Q);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_maxpool2x2_fwd(const void* x, void* y, uint8_t* idx, int dtype, int N, int H, int W, int C, void* stream) {
    const int P = H / 2, Q = W / 2;
    if (!x || !y || !idx || N <= 0 || C <= 0 || P <= 0 || Q <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * P * Q * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(maxpool2_fwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(x), BFW(y), idx, N, H, W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(x), FPW(y), idx, N, H, W, C, P, Q);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_maxpool2x2_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype, int N, int H, int W, int C, void* stream) {
    const int P = H / 2, Q = W / 2;
    if (!dy || !idx || !dx || N <= 0 || C <= 0 || P <= 0 || Q <= 0 || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * H * W * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(maxpool2_bwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(dy), idx, BFW(dx), N, H, W, C, P, Q);
    else
        hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(dy), idx, FPW(dx), N, H, W, C, P, Q);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_upsample_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW,
                                          void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SSCG_ERR_BAD_ARG;
    float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    size_t total = (size_t)N * OH * OW * C;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, OH, OW, sh, sw);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_upsample_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW,
                                          void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return SSCG_ERR_BAD_ARG;
    float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    // sh == 0 (single source row) => every output row touches row 0: scan the full range
    float inv_sh = sh > 0.f ? 1.f / sh : (float)OH;
    float inv_sw = sw > 0.f ? 1.f / sw : (float)OW;
    size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C, OH, OW,
                       sh, sw, inv_sh, inv_sw);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_reflect_pad(const void* x, void* y, int dtype, int N, int H, int W, int C, int pad, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0 || pad >= H || pad >= W || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * (H + 2 * pad) * (W + 2 * pad) * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(reflect_pad_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(x), BFW(y), N, H, W, C, pad);
    else
        hipLaunchKernelGGL(reflect_pad_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(x), FPW(y), N, H, W, C, pad);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_reflect_pad_bwd(const void* dy, void* dx, int dtype, int N, int H, int W, int C, int pad, void* stream) {
    if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || pad < 0 || pad >= H || pad >= W || !SSCG_DT_OK(dtype)) return SSCG_ERR_BAD_ARG;
    size_t total = (size_t)N * H * W * C;
    if (dtype == SSCG_BF16)
        hipLaunchKernelGGL(reflect_pad_bwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, BF(dy), BFW(dx), N, H, W, C, pad);
    else
        hipLaunchKernelGGL(reflect_pad_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, FP(dy), FPW(dx), N, H, W, C, pad);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SSCG_ERR_BAD_ARG;
    int R = C, Cc = H * W;  // [C][HW] -> [HW][C]
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, R, Cc);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SSCG_ERR_BAD_ARG;
    int R = H * W, Cc = C;  // [HW][C] -> [C][HW]
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(Cc, 32), cdiv(R, 32), N), dim3(256), 0, (hipStream_t)stream, x, y, R, Cc);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// ---- input pipeline: uint8 pixels / label ids -> the tensors the step consumes
__global__ __launch_bounds__(256) void image_u8_to_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t n,
                                                               int C, const float* __restrict__ mean,
                                                               const float* __restrict__ stdev) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float t = __fdiv_rn((float)src[i], 255.0f);               // ToTensor: .div(255)
        dst[i] = __fdiv_rn(__fsub_rn(t, mean[c]), stdev[c]);             // Normalize: .sub_(mean).div_(std)
    }
}

__global__ __launch_bounds__(256) void label_lut_kernel(const uint8_t* __restrict__ src, int64_t* __restrict__ dst, size_t n,
                                                         const int64_t* __restrict__ lut) {
    __shared__ int64_t t[256];
    t[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = t[src[i]];
}

extern "C" int sscg_image_u8_to_f32(const uint8_t* src, float* dst, int64_t rows, int C, const float* mean, const float* stdev,
                                    void* stream) {
    if (!src || !dst || !mean || !stdev || rows <= 0 || C <= 0) return SSCG_ERR_BAD_ARG;
    const size_t n = (size_t)rows * C;
    hipLaunchKernelGGL(image_u8_to_f32_kernel, dim3(ew_blocks((int64_t)n)), dim3(256), 0, (hipStream_t)stream, src, dst, n, C, mean,
                       stdev);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_label_lut(const uint8_t* src, int64_t* dst, int64_t n, const int64_t* lut256, void* stream) {
    if (!src || !dst || !lut256 || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(label_lut_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (size_t)n, lut256);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}
