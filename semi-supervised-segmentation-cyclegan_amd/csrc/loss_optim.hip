// Class-axis ops, loss/grad kernels and the fused Adam step.  Reference call sites per entry point
// are listed in include/sscg.h.  Losses are mean reductions accumulated in fp64 in two fixed-order
// stages (deterministic), written as one fp32 device scalar.
#include "common.h"
#include "sscg_internal.h"

namespace {

constexpr int LOSS_BLOCKS = 1024;
constexpr int MAXC = 64;  // class axis is 4 / 20 / 21 in the reference (model.py:205-210)

inline int ew_blocks(size_t n, int cap = 8192) {
    size_t b = (n + 255) / 256;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ void block_sum_to(double v, double* out) {
    __shared__ double sm[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void finish_loss_kernel(const double* __restrict__ part, int nparts, double inv_n, float* __restrict__ loss) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += sm[i];
        *loss = (float)(t * inv_n);
    }
}

// ---------------------------------------------------------------- softmax / argmax / one-hot
__global__ void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t rows, int C) {
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* xr = x + r * C;
        float v[MAXC];
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) { v[c] = xr[c]; m = fmaxf(m, v[c]); }
        float s = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
        float inv = 1.f / s;
        float* yr = y + r * C;
        for (int c = 0; c < C; ++c) yr[c] = v[c] * inv;
    }
}

__global__ void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                   size_t rows, int C) {
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* yr = y + r * C;
        const float* gr = dy + r * C;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += yr[c] * gr[c];
        float* dr = dx + r * C;
        for (int c = 0; c < C; ++c) dr[c] = yr[c] * (gr[c] - dot);
    }
}

__global__ void argmax_onehot_kernel(const float* __restrict__ x, float* __restrict__ oh, int64_t* __restrict__ index,
                                     size_t rows, int C) {
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* xr = x + r * C;
        float best = xr[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            float v = xr[c];
            if (v > best) { best = v; bi = c; }  // strict: first maximum wins (torch.max(dim) tie rule)
        }
        if (oh) {
            float* o = oh + r * C;
            for (int c = 0; c < C; ++c) o[c] = (c == bi) ? 1.f : 0.f;
        }
        if (index) index[r] = bi;
    }
}

__global__ void label_onehot_kernel(const int64_t* __restrict__ lab, float* __restrict__ oh, size_t rows, int C) {
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const int64_t l64 = lab[r];
        const int l = (l64 < 0 || l64 >= C) ? -1 : (int)l64;     // out-of-range id: an all-zero row, as scatter_ would refuse it
        float* o = oh + r * C;
        for (int c = 0; c < C; ++c) o[c] = (c == l) ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------- cross entropy
// Pixels whose label lies outside [0, C) take no part in the loss (nn.CrossEntropyLoss's ignore_index semantics,
// extended to every out-of-range id: the 255 "void" of an un-relabelled VOC map, a raw Cityscapes id, -100): they add
// nothing to the sum, are not counted in the mean and get a zero gradient - never an out-of-bounds read.
__global__ void ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, size_t rows, int C,
                              double* __restrict__ part, int nparts) {
    double acc = 0.0, cnt = 0.0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const int64_t l = lab[r];
        if (l < 0 || l >= C) continue;
        const float* xr = x + r * C;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, xr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(xr[c] - m);
        acc += (double)(logf(s) + m - xr[l]);
        cnt += 1.0;
    }
    block_sum_to(acc, part + blockIdx.x);
    __syncthreads();
    block_sum_to(cnt, part + nparts + blockIdx.x);
}

// loss = sum / count; count (the number of pixels with a valid label) also goes to `valid` for the backward pass
__global__ void finish_ce_kernel(const double* __restrict__ part, int nparts, float* __restrict__ loss, float* __restrict__ valid) {
    __shared__ double sm[512];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { s += part[i]; c += part[nparts + i]; }
    sm[threadIdx.x] = s;
    sm[256 + threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, n = 0.0;
        for (int i = 0; i < 256; ++i) { t += sm[i]; n += sm[256 + i]; }
        *loss = n > 0.0 ? (float)(t / n) : __builtin_nanf("");      // no pixel with a label in [0, C): NaN, as nn.CrossEntropyLoss gives
        if (valid) *valid = (float)n;
    }
}

__global__ void ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, size_t rows, int C,
                              const float* __restrict__ gscale, float w, const float* __restrict__ valid,
                              float* __restrict__ dx) {
    const float n = valid ? *valid : (float)rows;
    const float g = (gscale ? *gscale : 1.f) * (n > 0.f ? w / n : 0.f);
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const float* xr = x + r * C;
        const int64_t l64 = lab[r];
        if (l64 < 0 || l64 >= C) {
            float* dr = dx + r * C;
            for (int c = 0; c < C; ++c) dr[c] = 0.f;
            continue;
        }
        float v[MAXC];
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) { v[c] = xr[c]; m = fmaxf(m, v[c]); }
        float s = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
        float inv = 1.f / s;
        const int l = (int)l64;
        float* dr = dx + r * C;
        for (int c = 0; c < C; ++c) dr[c] = (v[c] * inv - (c == l ? 1.f : 0.f)) * g;
    }
}

// ---------------------------------------------------------------- MSE vs constant, L1
__global__ void mse_const_fwd_kernel(const float* __restrict__ x, size_t n, float target, double* __restrict__ part) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float d = x[i] - target;
        acc += (double)(d * d);
    }
    block_sum_to(acc, part + blockIdx.x);
}

__global__ void mse_const_bwd_kernel(const float* __restrict__ x, size_t n, float target, const float* __restrict__ gscale,
                                     float w, float* __restrict__ dx) {
    const float g = (gscale ? *gscale : 1.f) * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = (x[i] - target) * g;
}

__global__ void l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* __restrict__ part) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += (double)fabsf(a[i] - b[i]);
    block_sum_to(acc, part + blockIdx.x);
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                              const float* __restrict__ gscale, float w, float* __restrict__ da) {
    const float g = (gscale ? *gscale : 1.f) * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float d = a[i] - b[i];
        da[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    }
}

// nn.MSELoss()(a, b) between two tensors (utils.perceptual_loss, utils.py:205-206); gradient to a (and to b when db != NULL)
__global__ void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, double* __restrict__ part) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double d = (double)a[i] - (double)b[i];
        acc += d * d;
    }
    block_sum_to(acc, part + blockIdx.x);
}

__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, const float* __restrict__ gscale,
                               float w, float* __restrict__ da, float* __restrict__ db) {
    const float g = (gscale ? *gscale : 1.f) * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = (a[i] - b[i]) * g;
        da[i] = d;
        if (db) db[i] = -d;
    }
}

struct WsumArgs {
    const float* t[16];
    float w[16];
    int n;
};

__global__ void weighted_sum_kernel(WsumArgs a, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < a.n; ++i) s += a.w[i] * (*a.t[i]);
        *out = s;
    }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam single-tensor arithmetic)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            __bf16* __restrict__ p16, int split, size_t n, float step_size, float omb1, float beta2, float omb2, float eps,
                            float inv_bc2_sqrt, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * omb1;          // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * beta2 + gi * gi * omb2;    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        const float pn = p[i] - step_size * (mi / denom);
        p[i] = pn;
        if (p16) {
            if (split) {      // the three bf16 planes of the split contraction (conv_split.hip), n elements apart
                const sscg_bf3 t = sscg_split3(pn);
                p16[i] = t.h; p16[n + i] = t.m; p16[2 * n + i] = t.l;
            } else {
                p16[i] = (__bf16)pn;       // bf16 operand copy of the fp32 master weight (RNE)
            }
        }
        m[i] = mi;
        v[i] = vi;
    }
}

}  // namespace

extern "C" int sscg_softmax_fwd(const float* x, float* y, int64_t rows, int C, void* stream) {
    if (!x || !y || rows <= 0 || C <= 0 || C > MAXC) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)rows, C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_softmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int C, void* stream) {
    if (!dy || !y || !dx || rows <= 0 || C <= 0 || C > MAXC) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (size_t)rows, C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_argmax_onehot(const float* x, float* onehot, int64_t* index, int64_t rows, int C, void* stream) {
    if (!x || (!onehot && !index) || rows <= 0 || C <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(argmax_onehot_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, onehot, index,
                       (size_t)rows, C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_label_onehot(const int64_t* labels, float* onehot, int64_t rows, int C, void* stream) {
    if (!labels || !onehot || rows <= 0 || C <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(label_onehot_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, labels, onehot,
                       (size_t)rows, C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// Confusion matrix of the evaluation pass: per-block LDS histogram (int32), then integer atomics into the global
// int64 matrix - integer sums are order independent, so this is bit-exact against np.bincount.
__global__ __launch_bounds__(256) void confusion_hist_kernel(const int64_t* __restrict__ lt, const int64_t* __restrict__ lp,
                                                              int64_t n, int C, unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int bins[];
    const int nb = C * C;
    for (int i = threadIdx.x; i < nb; i += 256) bins[i] = 0u;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = lt[i];
        const int64_t q = lp[i];
        if (t >= 0 && t < C && q >= 0 && q < C) atomicAdd(&bins[(int)t * C + (int)q], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (bins[i]) atomicAdd(&hist[i], (unsigned long long)bins[i]);
}

extern "C" int sscg_confusion_hist(const int64_t* label_true, const int64_t* label_pred, int64_t n, int C, int64_t* hist,
                                   void* stream) {
    if (!label_true || !label_pred || !hist || n < 0 || C <= 0 || C > 64) return SSCG_ERR_BAD_ARG;
    if (n == 0) return SSCG_OK;
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);   // ~16 pixels per thread: few global atomics
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(confusion_hist_kernel, dim3((unsigned)blocks), dim3(256), (size_t)C * C * sizeof(unsigned int),
                       (hipStream_t)stream, label_true, label_pred, n, C, reinterpret_cast<unsigned long long*>(hist));
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" size_t sscg_loss_workspace(int64_t n) {
    (void)n;
    return (size_t)2 * LOSS_BLOCKS * sizeof(double);   // cross entropy keeps (sum, count) partials
}

extern "C" int sscg_ce_fwd(const float* logits, const int64_t* labels, int64_t rows, int C, float* loss, float* valid,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!logits || !labels || !loss || rows <= 0 || C <= 0 || C > MAXC) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < sscg_loss_workspace(rows)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nb = ew_blocks(rows, LOSS_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nb), dim3(256), 0, st, logits, labels, (size_t)rows, C, part, nb);
    hipLaunchKernelGGL(finish_ce_kernel, dim3(1), dim3(256), 0, st, part, nb, loss, valid);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_ce_bwd(const float* logits, const int64_t* labels, int64_t rows, int C, const float* gscale, float w,
                           const float* valid, float* dx, void* stream) {
    if (!logits || !labels || !dx || rows <= 0 || C <= 0 || C > MAXC) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, logits, labels, (size_t)rows,
                       C, gscale, w, valid, dx);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_mse_const_fwd(const float* x, int64_t n, float target, float* loss, void* ws, size_t ws_bytes,
                                  void* stream) {
    if (!x || !loss || n <= 0) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < sscg_loss_workspace(n)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nb = ew_blocks(n, LOSS_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(mse_const_fwd_kernel, dim3(nb), dim3(256), 0, st, x, (size_t)n, target, part);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, loss);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_mse_const_bwd(const float* x, int64_t n, float target, const float* gscale, float w, float* dx,
                                  void* stream) {
    if (!x || !dx || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(mse_const_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, target,
                       gscale, 2.f * w / (float)n, dx);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_l1_fwd(const float* a, const float* b, int64_t n, float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !loss || n <= 0) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < sscg_loss_workspace(n)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int nb = ew_blocks(n, LOSS_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(l1_fwd_kernel, dim3(nb), dim3(256), 0, st, a, b, (size_t)n, part);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, loss);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_l1_bwd(const float* a, const float* b, int64_t n, const float* gscale, float w, float* da, void* stream) {
    if (!a || !b || !da || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, (size_t)n, gscale,
                       w / (float)n, da);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_mse_fwd(const float* a, const float* b, int64_t n, float* loss, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !loss || n <= 0) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < sscg_loss_workspace(n)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = ew_blocks(n, LOSS_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(mse_fwd_kernel, dim3(nb), dim3(256), 0, st, a, b, (size_t)n, part);
    hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)n, loss);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_mse_bwd(const float* a, const float* b, int64_t n, const float* gscale, float w, float* da, float* db, void* stream) {
    if (!a || !b || !da || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, (size_t)n, gscale,
                       2.f * w / (float)n, da, db);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_weighted_sum(const float* const* terms, const float* w, int n, float* out, void* stream) {
    if (!terms || !w || !out || n <= 0 || n > 16) return SSCG_ERR_BAD_ARG;
    WsumArgs a;
    for (int i = 0; i < 16; ++i) { a.t[i] = i < n ? terms[i] : nullptr; a.w[i] = i < n ? w[i] : 0.f; }
    a.n = n;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype, int64_t n,
                              double lr, double beta1, double beta2, double eps, int step, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return SSCG_ERR_BAD_ARG;
    if (shadow && shadow_dtype != SSCG_BF16 && shadow_dtype != SSCG_BF16X3) return SSCG_ERR_BAD_ARG;
    // hyper-parameters arrive as doubles (python floats): 1 - beta must not be formed in fp32
    double bc1 = 1.0 - pow(beta1, (double)step);
    double bc2 = 1.0 - pow(beta2, (double)step);
    float step_size = (float)(lr / bc1);
    float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n, 16384)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, reinterpret_cast<__bf16*>(shadow), shadow_dtype == SSCG_BF16X3 ? 1 : 0, (size_t)n, step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       inv_bc2_sqrt, grad_scale);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}
