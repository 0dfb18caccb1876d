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

// ---------------------------------------------------------------- upsample -> {softmax, cross entropy} from the low-resolution logits
// model.py:390-392 (interp = bilinear, align_corners=True, to the crop size), :398 / :455 (CrossEntropyLoss on the resized logits),
// :401-402 (softmax of the resized logits).  The resized [N][OH][OW][C] logits are never written: one block per SOURCE pixel walks
// the output pixels whose bilinear stencil touches it (gather form: deterministic, no atomics), interpolates their logits from the
// low-resolution map (L2-resident: N x 33 x 33 x 21 fp32 = 0.7 MB), and
//   * the block that owns an output pixel (its stencil's top-left corner) adds its loss term / writes its softmax row,
//   * every block sums weight * d(loss)/d(resized logit) of its output pixels: the gradient with respect to the low-resolution
//     logits, which for the cross entropy depends on nothing but logits and labels - so the FORWARD pass already leaves it
//     (unscaled; the backward is an elementwise scale by g / valid).
struct HeadGeom { int N, H, W, C, OH, OW; float sh, sw, inv_sh, inv_sw; };

template <int CT>
__device__ __forceinline__ void head_logits(const float* __restrict__ xn, const HeadGeom& g, int oy, int ox, int C, float* v,
                                            int* y0o, int* x0o) {
    // the arithmetic of upsample_fwd_kernel (pointwise.hip)
    const float fy = g.sh * oy, fx = g.sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int yp = y0 < g.H - 1 ? 1 : 0, xp = x0 < g.W - 1 ? 1 : 0;
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* r00 = xn + ((size_t)y0 * g.W + x0) * C;
    const float* r01 = r00 + (size_t)xp * C;
    const float* r10 = r00 + (size_t)yp * g.W * C;
    const float* r11 = r10 + (size_t)xp * C;
#pragma unroll
    for (int c = 0; c < (CT ? CT : MAXC); ++c) {
        if (CT == 0 && c >= C) break;
        v[c] = hy * (hx * r00[c] + lx * r01[c]) + ly * (hx * r10[c] + lx * r11[c]);
    }
    *y0o = y0; *x0o = x0;
}

// weight of source row / column `i` in the stencil of output row / column `o` (upsample_bwd_kernel's)
__device__ __forceinline__ float head_weight(float scale, int o, int i, int n_src) {
    const float f = scale * o;
    const int i0 = (int)f;
    const int ip = i0 < n_src - 1 ? 1 : 0;
    const float l = f - i0;
    float w = 0.f;
    if (i0 == i) w += 1.f - l;
    if (i0 + ip == i) w += l;
    return w;
}

// MODE 0: forward (labels and / or softmax output); MODE 1: backward of the softmax output (dy_soft), plus the scaled
// cross-entropy gradient the forward left
template <int CT, int MODE>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, float* __restrict__ y_soft,
                                                   const float* __restrict__ dy_soft, float* __restrict__ dlo, const float* __restrict__ dl_ce,
                                                   const float* __restrict__ g_ce, const float* __restrict__ valid,
                                                   double* __restrict__ part, int nparts, HeadGeom g) {
    __shared__ float red[4][MAXC];
    const int C = CT ? CT : g.C;
    const int b = blockIdx.x;
    const int ix = b % g.W, iy = (b / g.W) % g.H, n = b / (g.W * g.H);
    int oy_lo = (int)floorf((iy - 1) * g.inv_sh) - 1, oy_hi = (int)ceilf((iy + 1) * g.inv_sh) + 1;
    int ox_lo = (int)floorf((ix - 1) * g.inv_sw) - 1, ox_hi = (int)ceilf((ix + 1) * g.inv_sw) + 1;
    oy_lo = max(oy_lo, 0); ox_lo = max(ox_lo, 0);
    oy_hi = min(oy_hi, g.OH - 1); ox_hi = min(ox_hi, g.OW - 1);
    const int nx = ox_hi - ox_lo + 1, cand = (oy_hi - oy_lo + 1) * nx;
    const float* xn = x + (size_t)n * g.H * g.W * C;
    float acc[CT ? CT : MAXC];
#pragma unroll
    for (int c = 0; c < (CT ? CT : MAXC); ++c) acc[c] = 0.f;
    double loss = 0.0, cnt = 0.0;
    for (int t = threadIdx.x; t < cand; t += 256) {
        const int oy = oy_lo + t / nx, ox = ox_lo + t % nx;
        const float wy = head_weight(g.sh, oy, iy, g.H);
        if (wy == 0.f) continue;
        const float wx = head_weight(g.sw, ox, ix, g.W);
        if (wx == 0.f) continue;
        const float w = wy * wx;
        float v[CT ? CT : MAXC];
        int y0, x0;
        head_logits<CT>(xn, g, oy, ox, C, v, &y0, &x0);
        const bool owner = y0 == iy && x0 == ix;
        const size_t o = ((size_t)n * g.OH + oy) * g.OW + ox;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < (CT ? CT : MAXC); ++c) { if (CT == 0 && c >= C) break; m = fmaxf(m, v[c]); }
        int l = -1;
        float vl = 0.f;
        if (MODE == 0 && lab) {
            const int64_t l64 = lab[o];
            l = (l64 < 0 || l64 >= C) ? -1 : (int)l64;
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < (CT ? CT : MAXC); ++c) {
            if (CT == 0 && c >= C) break;
            if (c == l) vl = v[c];
            v[c] = expf(v[c] - m);
            s += v[c];
        }
        const float inv = 1.f / s;
        if (MODE == 0) {
            if (owner && l >= 0) { loss += (double)(logf(s) + m - vl); cnt += 1.0; }
            if (owner && y_soft) {
                float* yr = y_soft + o * C;
#pragma unroll
                for (int c = 0; c < (CT ? CT : MAXC); ++c) { if (CT == 0 && c >= C) break; yr[c] = v[c] * inv; }
            }
            if (l >= 0) {
#pragma unroll
                for (int c = 0; c < (CT ? CT : MAXC); ++c) { if (CT == 0 && c >= C) break; acc[c] += w * (v[c] * inv - (c == l ? 1.f : 0.f)); }
            }
        } else {
            const float* gr = dy_soft + o * C;
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < (CT ? CT : MAXC); ++c) { if (CT == 0 && c >= C) break; v[c] *= inv; dot += v[c] * gr[c]; }
#pragma unroll
            for (int c = 0; c < (CT ? CT : MAXC); ++c) { if (CT == 0 && c >= C) break; acc[c] += w * (v[c] * (gr[c] - dot)); }
        }
    }
    const bool want_sum = MODE == 1 || (lab && dlo);
    if (want_sum) {
#pragma unroll
        for (int c = 0; c < (CT ? CT : MAXC); ++c) {
            if (CT == 0 && c >= C) break;
            const float r = wave_sum(acc[c]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = r;
        }
        __syncthreads();
        if ((int)threadIdx.x < C) {
            const int c = threadIdx.x;
            float r = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            if (MODE == 1 && dl_ce) {
                const float nv = valid ? *valid : 0.f;
                r += dl_ce[(size_t)b * C + c] * ((g_ce ? *g_ce : 1.f) * (nv > 0.f ? 1.f / nv : 0.f));
            }
            dlo[(size_t)b * C + c] = r;
        }
    }
    if (MODE == 0 && lab && part) {
        __syncthreads();
        block_sum_to(loss, part + b);
        __syncthreads();
        block_sum_to(cnt, part + nparts + b);
    }
}

// backward of the cross entropy alone: dx = dl * g / valid
__global__ void head_scale_kernel(const float* __restrict__ dl, const float* __restrict__ g_ce, const float* __restrict__ valid,
                                  float* __restrict__ dx, size_t n) {
    const float nv = valid ? *valid : 0.f;
    const float k = (g_ce ? *g_ce : 1.f) * (nv > 0.f ? 1.f / nv : 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dx[i] = dl[i] * k;
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

static bool head_geom(HeadGeom* g, int N, int H, int W, int C, int OH, int OW) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > MAXC || OH <= 0 || OW <= 0) return false;
    if ((size_t)N * H * W >= ((size_t)1 << 31)) return false;
    g->N = N; g->H = H; g->W = W; g->C = C; g->OH = OH; g->OW = OW;
    g->sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    g->sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    g->inv_sh = g->sh > 0.f ? 1.f / g->sh : (float)OH;
    g->inv_sw = g->sw > 0.f ? 1.f / g->sw : (float)OW;
    return true;
}

template <int MODE>
static void launch_head(const HeadGeom& g, hipStream_t st, const float* x, const int64_t* lab, float* y_soft, const float* dy_soft,
                        float* dlo, const float* dl_ce, const float* g_ce, const float* valid, double* part, int nparts) {
    const dim3 grid(nparts), blk(256);
    if (g.C == 21) hipLaunchKernelGGL((head_kernel<21, MODE>), grid, blk, 0, st, x, lab, y_soft, dy_soft, dlo, dl_ce, g_ce, valid, part, nparts, g);
    else if (g.C == 20) hipLaunchKernelGGL((head_kernel<20, MODE>), grid, blk, 0, st, x, lab, y_soft, dy_soft, dlo, dl_ce, g_ce, valid, part, nparts, g);
    else if (g.C == 4) hipLaunchKernelGGL((head_kernel<4, MODE>), grid, blk, 0, st, x, lab, y_soft, dy_soft, dlo, dl_ce, g_ce, valid, part, nparts, g);
    else hipLaunchKernelGGL((head_kernel<0, MODE>), grid, blk, 0, st, x, lab, y_soft, dy_soft, dlo, dl_ce, g_ce, valid, part, nparts, g);
}

extern "C" size_t sscg_upsample_head_workspace(int N, int H, int W) {
    return (size_t)2 * (size_t)(N > 0 ? N : 0) * (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0) * sizeof(double);
}

extern "C" int sscg_upsample_head_fwd(const float* x, const int64_t* labels, float* y_soft, float* loss, float* valid, float* dlogits,
                                      int N, int H, int W, int C, int OH, int OW, void* ws, size_t ws_bytes, void* stream) {
    HeadGeom g;
    if (!x || !head_geom(&g, N, H, W, C, OH, OW) || (!labels && !y_soft)) return SSCG_ERR_BAD_ARG;
    if (labels && (!loss || !valid || !dlogits)) return SSCG_ERR_BAD_ARG;
    if (labels && (!ws || ws_bytes < sscg_upsample_head_workspace(N, H, W))) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nparts = N * H * W;
    double* part = labels ? reinterpret_cast<double*>(ws) : nullptr;
    launch_head<0>(g, st, x, labels, y_soft, nullptr, labels ? dlogits : nullptr, nullptr, nullptr, nullptr, part, nparts);
    if (labels) hipLaunchKernelGGL(finish_ce_kernel, dim3(1), dim3(256), 0, st, part, nparts, loss, valid);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_upsample_head_bwd(const float* x, const float* dy_soft, const float* dlogits, const float* g_ce, const float* valid,
                                      float* dx, int N, int H, int W, int C, int OH, int OW, void* stream) {
    HeadGeom g;
    if (!x || !dx || !head_geom(&g, N, H, W, C, OH, OW) || (!dy_soft && !dlogits)) return SSCG_ERR_BAD_ARG;
    if (dlogits && !valid) return SSCG_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nparts = N * H * W;
    if (dy_soft)
        launch_head<1>(g, st, x, nullptr, nullptr, dy_soft, dx, dlogits, g_ce, valid, nullptr, nparts);
    else {
        const size_t n = (size_t)nparts * C;
        hipLaunchKernelGGL(head_scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, dlogits, g_ce, valid, dx, n);
    }
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
