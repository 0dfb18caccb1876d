// InstanceNorm / BatchNorm (training statistics) + activation + residual, forward and backward, and
// the column-sum used for bias gradients.  All HBM-bound: float4 over the channel axis, fp64
// accumulation of per-channel sums (two-stage, fixed order => deterministic).
//
// Reference semantics: nn.InstanceNorm2d(affine=False, track_running_stats=False) (arch/ops.py:11),
// nn.BatchNorm2d in train mode with frozen affine (arch/generators.py:326-337,390,417) or trainable
// affine (arch/ops.py:9), followed by nn.ReLU / nn.LeakyReLU(0.2) (arch/ops.py:44,50,57) and, in the
// DeepLab Bottleneck, `out += residual; relu` (arch/generators.py:362-363).
//
// The tensor is viewed as [G][L][C] (see sscg.h).
#include "common.h"
#include "sscg_internal.h"
#include <cstdlib>

namespace {

enum { RM_SUM = 0, RM_STATS = 1, RM_BWD = 2, RM_BWD_HEAD = 3 };

struct RedParams {
    const void* __restrict__ x;      // SUM/STATS: input; BWD: x (pre-norm)           (fp32 or bf16: template T)
    const void* __restrict__ dy;     // BWD
    const void* __restrict__ y;      // BWD: forward output (activation mask)
    const float* __restrict__ mean;  // BWD [G][C]
    const float* __restrict__ rstd;  // BWD [G][C]
    const float* __restrict__ gamma; // BWD, y == NULL: the activation mask is recomputed as gamma * xhat + beta > 0
    const float* __restrict__ beta;
    double* __restrict__ part;       // [G][chunks][C][2]
    const float* __restrict__ head_w;    // BWD_HEAD: dy[r][c] = head_dout[r] * head_w[c] (the 1x1 single-channel conv behind the activation)
    const float* __restrict__ head_dout; // BWD_HEAD [G * L]
    double* __restrict__ head_part;      // BWD_HEAD [G][chunks][C][2]: (sum act(y) * dout, sum dout)
    long L;
    int C;
    int chunks;
    long rows_per_chunk;
    int act;
    float slope;
};

__device__ __forceinline__ float act_grad(float dy, float y, int act, float slope) {
    if (act == SSCG_ACT_RELU) return y > 0.f ? dy : 0.f;
    if (act == SSCG_ACT_LRELU) return y > 0.f ? dy : dy * slope;
    if (act == SSCG_ACT_TANH) return dy * (1.f - y * y);
    return dy;
}

template <typename T, int VEC> __device__ __forceinline__ void ldv(const T* p, float v[VEC]) {
    if constexpr (VEC == 8) ld8<T>(p, v);
    else if constexpr (VEC == 4) ld4<T>(p, v);
    else v[0] = ld1<T>(p);
}
template <typename T, int VEC> __device__ __forceinline__ void stv(T* p, const float v[VEC]) {
    if constexpr (VEC == 8) st8<T>(p, v);
    else if constexpr (VEC == 4) st4<T>(p, v);
    else st1<T>(p, v[0]);
}

// block = 256 threads = RW row-lanes x CW column groups (CW power of two), VEC channels per group; a block owns the
// channel slab blockIdx.z and the row chunk blockIdx.x of group blockIdx.y.  Rows are taken U at a time so that U
// (3U for the backward sums) independent 16-byte loads are in flight per thread: these tensors are a few MB, the
// kernel lives for a handful of memory latencies and nothing else hides them.
template <int MODE, typename T, int VEC>
__global__ __launch_bounds__(256) void col_reduce_kernel(RedParams p, int CW) {
    constexpr int U = VEC == 8 ? 2 : 4;
    const T* px = reinterpret_cast<const T*>(p.x);
    const T* pdy = reinterpret_cast<const T*>(p.dy);
    const T* py = reinterpret_cast<const T*>(p.y);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sm = reinterpret_cast<double*>(smem_raw);  // [256][VEC][2]
    const int tid = threadIdx.x;
    const int RW = 256 / CW;
    const int col = tid % CW;
    const int rl = tid / CW;
    const int chunk = blockIdx.x;
    const int g = blockIdx.y;
    const int c = (blockIdx.z * CW + col) * VEC;
    const bool cok = c < p.C;

    constexpr bool BWD = MODE == RM_BWD || MODE == RM_BWD_HEAD;
    constexpr bool HEAD = MODE == RM_BWD_HEAD;
    double s0[VEC], s1[VEC], s2[HEAD ? VEC : 1], sd = 0.0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s0[e] = 0.0; s1[e] = 0.0; }
    if constexpr (HEAD) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s2[e] = 0.0;
    }

    float mu[VEC], rs[VEC], ga[VEC], be[VEC], hw[VEC];
    const bool remask = BWD && p.act != SSCG_ACT_NONE && p.y == nullptr;   // no residual joined the forward: y = act(gamma * xhat + beta)
    if (BWD && cok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            mu[e] = p.mean[(size_t)g * p.C + c + e]; rs[e] = p.rstd[(size_t)g * p.C + c + e];
            ga[e] = ((remask || HEAD) && p.gamma) ? p.gamma[c + e] : 1.f;
            be[e] = ((remask || HEAD) && p.beta) ? p.beta[c + e] : 0.f;
            hw[e] = HEAD ? p.head_w[c + e] : 0.f;
        }
    }

    if (cok) {
        const long r_begin = (long)chunk * p.rows_per_chunk;
        const long r_end = min(p.L, r_begin + p.rows_per_chunk);
        const size_t base = (size_t)g * p.L * p.C + c;
        const bool has_y = BWD && p.act != SSCG_ACT_NONE && !remask;  // y may be NULL without an activation / with the recomputed mask
        for (long r0 = r_begin + rl; r0 < r_end; r0 += (long)RW * U) {
            float xv[U][VEC], dv[U][VEC], yv[U][VEC], dout[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long r = r0 + (long)u * RW;
                ok[u] = r < r_end;
                const size_t o = base + (size_t)(ok[u] ? r : r_begin) * p.C;
                ldv<T, VEC>(px + o, xv[u]);
                if constexpr (HEAD) {
                    dout[u] = p.head_dout[(size_t)g * p.L + (ok[u] ? r : r_begin)];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { dv[u][e] = dout[u] * hw[e]; yv[u][e] = 0.f; }
                } else if (MODE == RM_BWD) {
                    ldv<T, VEC>(pdy + o, dv[u]);
                    if (has_y) ldv<T, VEC>(py + o, yv[u]);
                    else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) yv[u][e] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (MODE == RM_SUM) {
                        s0[e] += (double)xv[u][e];
                    } else if (MODE == RM_STATS) {
                        double d = (double)xv[u][e];
                        s0[e] += d;
                        s1[e] += d * d;
                    } else {
                        float xh = (xv[u][e] - mu[e]) * rs[e];
                        float ym = yv[u][e];
                        if (remask || HEAD) ym = xh * ga[e] + be[e];      // the forward's own expression (norm_apply_kernel): same sign
                        float gg = act_grad(dv[u][e], ym, p.act, p.slope);
                        s0[e] += (double)gg;
                        s1[e] += (double)gg * (double)xh;
                        if constexpr (HEAD) s2[e] += (double)sscg_act(ym, p.act, p.slope) * (double)dout[u];
                    }
                }
                if constexpr (HEAD) sd += (double)dout[u];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sm[(tid * VEC + e) * 2] = s0[e]; sm[(tid * VEC + e) * 2 + 1] = s1[e]; }
    __syncthreads();
    if (rl == 0 && cok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < RW; ++r) {
                a += sm[((r * CW + col) * VEC + e) * 2];
                b += sm[((r * CW + col) * VEC + e) * 2 + 1];
            }
            if (c + e < p.C) {
                size_t o = (((size_t)g * p.chunks + chunk) * p.C + c + e) * 2;
                p.part[o] = a;
                p.part[o + 1] = b;
            }
        }
    }
    if constexpr (HEAD) {       // second round through the same LDS: the head's weight-gradient sums
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) { sm[(tid * VEC + e) * 2] = s2[e]; sm[(tid * VEC + e) * 2 + 1] = sd; }
        __syncthreads();
        if (rl == 0 && cok) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double a = 0.0, b = 0.0;
                for (int r = 0; r < RW; ++r) {
                    a += sm[((r * CW + col) * VEC + e) * 2];
                    b += sm[((r * CW + col) * VEC + e) * 2 + 1];
                }
                if (c + e < p.C) {
                    size_t o = (((size_t)g * p.chunks + chunk) * p.C + c + e) * 2;
                    p.head_part[o] = a;
                    p.head_part[o + 1] = b;
                }
            }
        }
    }
}

struct RedPlan {
    int vec, CW, slabs, chunks;
    long rows_per_chunk;
};

int vec_for(int C, int dtype) { return (dtype == SSCG_BF16 && C % 8 == 0) ? 8 : ((C % 4 == 0) ? 4 : 1); }

RedPlan plan_reduce(int G, long L, int C, int dtype = SSCG_F32) {
    RedPlan pl;
    pl.vec = vec_for(C, dtype);
    const int groups = C / pl.vec;
    // 256-byte row segments per block: narrow slabs => many row lanes => few chunks => a short second stage
    const int cw_max = pl.vec >= 4 ? 16 : 64;
    int cw = 1;
    while (cw < groups && cw < cw_max) cw <<= 1;
    pl.CW = cw;
    pl.slabs = cdiv(groups, cw);
    const int rw = 256 / cw;
    long budget = 2048 / ((long)G * pl.slabs);   // ~2048 workgroups at most
    if (budget < 1) budget = 1;
    if (budget > 128) budget = 128;
    long by_rows = cdiv(L, (long)rw * 8);        // at least ~8 rows per thread
    if (by_rows < 1) by_rows = 1;
    long chunks = by_rows < budget ? by_rows : budget;
    pl.rows_per_chunk = cdiv(L, chunks);
    pl.chunks = cdiv(L, pl.rows_per_chunk);
    return pl;
}

// the workspace bound covers both element types (the chunk count does not depend on the vector width)
size_t part_bytes(int G, long L, int C) {
    const int a = plan_reduce(G, L, C, SSCG_F32).chunks, b = plan_reduce(G, L, C, SSCG_BF16).chunks;
    return (size_t)G * (a > b ? a : b) * C * 2 * sizeof(double);
}

template <int MODE>
int launch_reduce(RedParams p, int G, int dtype, hipStream_t st) {
    RedPlan pl = plan_reduce(G, p.L, p.C, dtype);
    p.chunks = pl.chunks;
    p.rows_per_chunk = pl.rows_per_chunk;
    dim3 grid(pl.chunks, G, pl.slabs);
    size_t smem = (size_t)256 * pl.vec * 2 * sizeof(double);
    if (dtype == SSCG_BF16) {
        if (pl.vec == 8) hipLaunchKernelGGL((col_reduce_kernel<MODE, __bf16, 8>), grid, dim3(256), smem, st, p, pl.CW);
        else if (pl.vec == 4) hipLaunchKernelGGL((col_reduce_kernel<MODE, __bf16, 4>), grid, dim3(256), smem, st, p, pl.CW);
        else hipLaunchKernelGGL((col_reduce_kernel<MODE, __bf16, 1>), grid, dim3(256), smem, st, p, pl.CW);
    } else {
        if (pl.vec == 4) hipLaunchKernelGGL((col_reduce_kernel<MODE, float, 4>), grid, dim3(256), smem, st, p, pl.CW);
        else hipLaunchKernelGGL((col_reduce_kernel<MODE, float, 1>), grid, dim3(256), smem, st, p, pl.CW);
    }
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// Second stage of the column reductions.  A block owns 4 channels, one wave each: the 64 lanes stride the chunk
// axis (<= a handful of dependent loads per lane) and are combined by a fixed-order butterfly (deterministic).
constexpr int FIN_CH = 4;
__device__ __forceinline__ void chunk_sum16(const double* __restrict__ part, int g, int chunks, int C, int c, bool cok,
                                            double* /*sm*/, double& s0, double& s1) {
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (cok) {
        for (int k = lane; k < chunks; k += 64) {
            size_t o = (((size_t)g * chunks + k) * C + c) * 2;
            a += part[o];
            b += part[o + 1];
        }
    }
    s0 = wave_sum(a);
    s1 = wave_sum(b);
}

__global__ __launch_bounds__(256) void finalize_sum_kernel(const double* __restrict__ part, float* __restrict__ out, int C,
                                                            int chunks, float beta) {
    __shared__ double sm[512];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    const bool cok = c < C;
    double s, unused;
    chunk_sum16(part, 0, chunks, C, c, cok, sm, s, unused);
    if ((threadIdx.x & 63) == 0 && cok) out[c] = (beta != 0.f ? beta * out[c] : 0.f) + (float)s;
}

// head (norm -> act -> 1x1 conv to one channel): dw[c] = sum over groups and chunks of act(y) * dout; dbias = sum dout
__global__ __launch_bounds__(256) void finalize_head_kernel(const double* __restrict__ part, float* __restrict__ dw,
                                                             float* __restrict__ dbias, int C, int chunks_total, float beta) {
    __shared__ double sm[512];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    const bool cok = c < C;
    double s, sd;
    chunk_sum16(part, 0, chunks_total, C, c, cok, sm, s, sd);
    if ((threadIdx.x & 63) == 0 && cok) {
        if (dw) dw[c] = (beta != 0.f ? beta * dw[c] : 0.f) + (float)s;
        if (c == 0 && dbias) dbias[0] = (beta != 0.f ? beta * dbias[0] : 0.f) + (float)sd;
    }
}

__global__ __launch_bounds__(256) void finalize_stats_kernel(const double* __restrict__ part, float* __restrict__ mean,
                                                              float* __restrict__ rstd, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, int G, int C, int chunks, long L,
                                                              float eps, float momentum) {
    __shared__ double sm[512];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    const bool cok = c < C;
    // With running statistics and G > 1 ("grouped" batch norm: G batches normalised separately in one launch) the
    // groups are walked in order by one wave per channel, so the EMA sees them exactly as G successive forwards would.
    const int g0 = rmean ? 0 : blockIdx.y;
    const int g1 = rmean ? G : blockIdx.y + 1;
    for (int g = g0; g < g1; ++g) {
        double s, ss;
        chunk_sum16(part, g, chunks, C, c, cok, sm, s, ss);
        if ((threadIdx.x & 63) == 0 && cok) {
            const int i = g * C + c;
            double m = s / (double)L;
            double var = ss / (double)L - m * m;
            if (var < 0.0) var = 0.0;
            mean[i] = (float)m;
            rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
            if (rmean) {
                double unb = L > 1 ? var * (double)L / (double)(L - 1) : var;
                rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
                rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
            }
        }
    }
}

// c1 = sum_g / L, c2 = sum_g_xhat / L  per (g, c);  dgamma/dbeta += sums (over g)
// rec_bm > 0: `part` was written by a data gradient's epilogue (sscg_conv2d_dgrad_bsums): group g owns the records of the tile rows
// that overlap it, rec_wm per tile row, `chunks` apart
__global__ __launch_bounds__(256) void finalize_bwd_kernel(const double* __restrict__ part, float* __restrict__ coef,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int G, int C,
                                                            int chunks, long L, int overwrite, int rec_bm = 0, int rec_wm = 0) {
    __shared__ double sm[512];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    const bool cok = c < C;
    const bool lead = (threadIdx.x & 63) == 0 && cok;
    double tg = 0.0, tb = 0.0;
    for (int g = 0; g < G; ++g) {
        double s, sx;
        int nvalid = chunks;
        if (rec_bm > 0) nvalid = (int)((((long)(g + 1) * L + rec_bm - 1) / rec_bm) - ((long)g * L) / rec_bm) * rec_wm;
        chunk_sum16(part + (size_t)g * (chunks - nvalid) * C * 2, g, nvalid, C, c, cok, sm, s, sx);
        if (lead) {
            coef[((size_t)g * C + c) * 2] = (float)(s / (double)L);
            coef[((size_t)g * C + c) * 2 + 1] = (float)(sx / (double)L);
            tb += s;
            tg += sx;
        }
    }
    if (lead && dgamma) dgamma[c] = (overwrite ? 0.f : dgamma[c]) + (float)tg;
    if (lead && dbeta) dbeta[c] = (overwrite ? 0.f : dbeta[c]) + (float)tb;
}

struct ApplyParams {
    const void* __restrict__ x;
    const float* __restrict__ mean;
    const float* __restrict__ rstd;
    const float* __restrict__ gamma;
    const float* __restrict__ beta;
    const void* __restrict__ res;
    void* __restrict__ y;
    long L;
    int C;
    int act;
    float slope;
    uint32_t total;  // elements / VEC  (< 2^31)
    FastDiv div_cg;  // by C/VEC
    FastDiv div_l;   // by L
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_apply_kernel(ApplyParams p) {
    const int CG = p.C / VEC;
    const T* px = reinterpret_cast<const T*>(p.x);
    const T* pr = reinterpret_cast<const T*>(p.res);
    T* py = reinterpret_cast<T*>(p.y);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.total; i += gridDim.x * 256) {
        const int row = fd_div((int)i, p.div_cg);
        const int c = ((int)i - row * CG) * VEC;
        const int g = fd_div(row, p.div_l);
        const size_t o = (size_t)i * VEC;
        float xv[VEC], rv[VEC];
        ldv<T, VEC>(px + o, xv);
        if (p.res) ldv<T, VEC>(pr + o, rv);
        float out[VEC];
        // per-channel parameters as 16-byte loads (the kernel is VMEM-issue bound with one scalar load per parameter)
        float mu[VEC], rs[VEC], ga[VEC], be[VEC];
        const size_t s = (size_t)g * p.C + c;
        ldv<float, VEC>(p.mean + s, mu);
        ldv<float, VEC>(p.rstd + s, rs);
        if (p.gamma) {
            ldv<float, VEC>(p.gamma + c, ga);
            ldv<float, VEC>(p.beta + c, be);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = (xv[e] - mu[e]) * rs[e];
            if (p.gamma) v = v * ga[e] + be[e];
            if (p.res) v += rv[e];
            out[e] = sscg_act(v, p.act, p.slope);
        }
        stv<T, VEC>(py + o, out);
    }
}

// The same pass with a thread bound to ONE channel group (round 6).  The kernel above walks the tensor as a flat list of vectors: every
// iteration divides its index twice, fetches four parameter vectors beside its one data vector and has that single 16-byte load in
// flight - 32 KB per CU, 3.3-5.7 TB/s measured (profiles/r05_norm_bench.txt).  Here a block owns `cw` consecutive channel groups
// (cw = C / VEC up to 256: a whole row or a 4 KB slab of it) x a chunk of one group's rows: the per-channel parameters are loaded
// once per thread, a thread walks its column with U rows in flight (consecutive lanes still touch consecutive 16 bytes, consecutive
// row lanes consecutive rows).  Same per-element expression, bit-identical results.  Grid (chunks, C / VEC / cw, G).
template <typename T, int VEC, int U>
__global__ __launch_bounds__(256) void norm_apply_slab_kernel(ApplyParams p, int cw_shift, int rows_per_chunk) {
    const int tid = threadIdx.x;
    const int col = tid & ((1 << cw_shift) - 1), rl = tid >> cw_shift, rw = 256 >> cw_shift;
    const int c = (((int)blockIdx.y << cw_shift) + col) * VEC;
    const int g = blockIdx.z;
    float mu[VEC], rs[VEC], ga[VEC], be[VEC];
    const size_t s = (size_t)g * p.C + c;
    ldv<float, VEC>(p.mean + s, mu);
    ldv<float, VEC>(p.rstd + s, rs);
    if (p.gamma) {
        ldv<float, VEC>(p.gamma + c, ga);
        ldv<float, VEC>(p.beta + c, be);
    }
    const size_t base = (size_t)g * p.L * p.C + c;
    const T* px = reinterpret_cast<const T*>(p.x) + base;
    const T* pr = p.res ? reinterpret_cast<const T*>(p.res) + base : nullptr;
    T* py = reinterpret_cast<T*>(p.y) + base;
    const long r_begin = (long)blockIdx.x * rows_per_chunk;
    const long r_end = r_begin + rows_per_chunk < p.L ? r_begin + rows_per_chunk : p.L;
    for (long r = r_begin + rl; r < r_end; r += (long)rw * U) {
        float xv[U][VEC], rv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + (long)u * rw;
            if (rr < r_end) {
                ldv<T, VEC>(px + (size_t)rr * p.C, xv[u]);
                if (pr) ldv<T, VEC>(pr + (size_t)rr * p.C, rv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + (long)u * rw;
            if (rr < r_end) {
                float out[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float v = (xv[u][e] - mu[e]) * rs[e];
                    if (p.gamma) v = v * ga[e] + be[e];
                    if (pr) v += rv[u][e];
                    out[e] = sscg_act(v, p.act, p.slope);
                }
                stv<T, VEC>(py + (size_t)rr * p.C, out);
            }
        }
    }
}

// geometry of the slab kernels: cw = min(C / VEC, 256) channel groups per block (a power of two), rw = 256 / cw row lanes, chunks of
// rows such that the launch has ~4096 blocks at most and a thread at least one full round of U rows
struct SlabPlan { bool ok; int cw_shift, slabs, chunks, rows_per_chunk; };
static SlabPlan plan_slab(int G, long L, int C, int vec, int U) {
    SlabPlan pl = {false, 0, 1, 1, 0};
    static const bool off = getenv("SSCG_NORM_SLAB") && atoi(getenv("SSCG_NORM_SLAB")) == 0;      // A/B aid: the flat kernels
    if (off || vec < 4 || C % vec) return pl;
    const int cg = C / vec;
    if (cg & (cg - 1)) return pl;                   // channel groups per row: a power of two
    int sh = 0;
    while ((1 << sh) < cg && sh < 8) ++sh;
    pl.cw_shift = sh;
    pl.slabs = cg >> sh;
    const int rw = 256 >> sh;
    const long round = (long)rw * U;
    long chunks = (L + round - 1) / round;
    long budget = 4096 / ((long)G * pl.slabs);
    if (budget < 1) budget = 1;
    if (chunks > budget) chunks = budget;
    long rpc = (L + chunks - 1) / chunks;
    rpc = (rpc + round - 1) / round * round;
    pl.chunks = (int)((L + rpc - 1) / rpc);
    pl.rows_per_chunk = (int)rpc;
    if (G > 65535 || pl.slabs > 65535 || rpc > 0x7fffffffL) return pl;
    pl.ok = true;
    return pl;
}

struct BwdApplyParams {
    const void* __restrict__ dy;
    const void* __restrict__ x;
    const void* __restrict__ y;
    const float* __restrict__ mean;
    const float* __restrict__ rstd;
    const float* __restrict__ gamma;
    const float* __restrict__ beta;  // y == NULL with an activation: mask recomputed as gamma * xhat + beta > 0
    const float* __restrict__ coef;  // [G][C][2] or null when stats are constants
    const float* __restrict__ head_w;    // non-null: dy[r][c] = head_dout[r] * head_w[c] (never read from memory)
    const float* __restrict__ head_dout;
    void* __restrict__ dx;
    void* __restrict__ dres;
    long L;
    int C;
    int act;
    float slope;
    uint32_t total;
    FastDiv div_cg;
    FastDiv div_l;
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(BwdApplyParams p) {
    const int CG = p.C / VEC;
    const T* pdy = reinterpret_cast<const T*>(p.dy);
    const T* px = reinterpret_cast<const T*>(p.x);
    const T* py = reinterpret_cast<const T*>(p.y);
    T* pdx = reinterpret_cast<T*>(p.dx);
    T* pdres = reinterpret_cast<T*>(p.dres);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < p.total; i += gridDim.x * 256) {
        const int row = fd_div((int)i, p.div_cg);
        const int c = ((int)i - row * CG) * VEC;
        const int g = fd_div(row, p.div_l);
        const size_t o = (size_t)i * VEC;
        const bool remask = p.act != SSCG_ACT_NONE && py == nullptr;
        float dv[VEC], xv[VEC], yv[VEC];
        if (p.head_w) {
            float hw[VEC];
            ldv<float, VEC>(p.head_w + c, hw);
            const float d = p.head_dout[row];
#pragma unroll
            for (int e = 0; e < VEC; ++e) dv[e] = d * hw[e];
        } else {
            ldv<T, VEC>(pdy + o, dv);
        }
        ldv<T, VEC>(px + o, xv);
        if (p.act != SSCG_ACT_NONE && !remask) ldv<T, VEC>(py + o, yv);
        float gx[VEC], gr[VEC];
        float mu[VEC], rsv[VEC], ga[VEC], be[VEC], c1[VEC], c2[VEC];
        const size_t s = (size_t)g * p.C + c;
        ldv<float, VEC>(p.rstd + s, rsv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ga[e] = 1.f; be[e] = 0.f; mu[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; }
        if (p.gamma) ldv<float, VEC>(p.gamma + c, ga);
        if (remask && p.beta) ldv<float, VEC>(p.beta + c, be);
        if (p.coef || remask) ldv<float, VEC>(p.mean + s, mu);
        if (p.coef) {
            if constexpr (VEC == 1) {
                c1[0] = p.coef[s * 2]; c2[0] = p.coef[s * 2 + 1];
            } else {
                float kk[2 * VEC];          // (c1, c2) pairs of channels c .. c+VEC-1
#pragma unroll
                for (int q = 0; q < 2 * VEC; q += 4) ld4<float>(p.coef + s * 2 + q, kk + q);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { c1[e] = kk[2 * e]; c2[e] = kk[2 * e + 1]; }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xh = (xv[e] - mu[e]) * rsv[e];
            float ym = yv[e];
            if (remask) ym = xh * ga[e] + be[e];
            float gg = p.act != SSCG_ACT_NONE ? act_grad(dv[e], ym, p.act, p.slope) : dv[e];
            gr[e] = gg;
            float v = gg;
            if (p.coef) v = gg - c1[e] - xh * c2[e];
            gx[e] = v * rsv[e] * ga[e];
        }
        stv<T, VEC>(pdx + o, gx);
        if (p.dres) stv<T, VEC>(pdres + o, gr);
    }
}

// norm_bwd_apply_kernel with a thread bound to one channel group (see norm_apply_slab_kernel): the six per-channel parameter vectors
// are loaded once per thread, U rows of (dy, x, y) in flight.  Same per-element expression.
template <typename T, int VEC, int U>
__global__ __launch_bounds__(256) void norm_bwd_apply_slab_kernel(BwdApplyParams p, int cw_shift, int rows_per_chunk) {
    const int tid = threadIdx.x;
    const int col = tid & ((1 << cw_shift) - 1), rl = tid >> cw_shift, rw = 256 >> cw_shift;
    const int c = (((int)blockIdx.y << cw_shift) + col) * VEC;
    const int g = blockIdx.z;
    const bool remask = p.act != SSCG_ACT_NONE && p.y == nullptr;
    const bool has_y = p.act != SSCG_ACT_NONE && !remask;
    float mu[VEC], rsv[VEC], ga[VEC], be[VEC], c1[VEC], c2[VEC], hw[VEC];
    const size_t s = (size_t)g * p.C + c;
    ldv<float, VEC>(p.rstd + s, rsv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ga[e] = 1.f; be[e] = 0.f; mu[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; hw[e] = 0.f; }
    if (p.gamma) ldv<float, VEC>(p.gamma + c, ga);
    if (remask && p.beta) ldv<float, VEC>(p.beta + c, be);
    if (p.coef || remask) ldv<float, VEC>(p.mean + s, mu);
    if (p.coef) {
        float kk[2 * VEC];          // (c1, c2) pairs of channels c .. c+VEC-1
#pragma unroll
        for (int q = 0; q < 2 * VEC; q += 4) ld4<float>(p.coef + s * 2 + q, kk + q);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { c1[e] = kk[2 * e]; c2[e] = kk[2 * e + 1]; }
    }
    if (p.head_w) ldv<float, VEC>(p.head_w + c, hw);
    const size_t base = (size_t)g * p.L * p.C + c;
    const T* pdy = p.head_w ? nullptr : reinterpret_cast<const T*>(p.dy) + base;
    const T* px = reinterpret_cast<const T*>(p.x) + base;
    const T* py = has_y ? reinterpret_cast<const T*>(p.y) + base : nullptr;
    T* pdx = reinterpret_cast<T*>(p.dx) + base;
    T* pdres = p.dres ? reinterpret_cast<T*>(p.dres) + base : nullptr;
    const float* pdo = p.head_w ? p.head_dout + (size_t)g * p.L : nullptr;
    const long r_begin = (long)blockIdx.x * rows_per_chunk;
    const long r_end = r_begin + rows_per_chunk < p.L ? r_begin + rows_per_chunk : p.L;
    for (long r = r_begin + rl; r < r_end; r += (long)rw * U) {
        float dv[U][VEC], xv[U][VEC], yv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + (long)u * rw;
            if (rr < r_end) {
                const size_t o = (size_t)rr * p.C;
                if (p.head_w) {
                    const float d = pdo[rr];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dv[u][e] = d * hw[e];
                } else {
                    ldv<T, VEC>(pdy + o, dv[u]);
                }
                ldv<T, VEC>(px + o, xv[u]);
                if (has_y) ldv<T, VEC>(py + o, yv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long rr = r + (long)u * rw;
            if (rr < r_end) {
                const size_t o = (size_t)rr * p.C;
                float gx[VEC], gr[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xh = (xv[u][e] - mu[e]) * rsv[e];
                    float ym = yv[u][e];
                    if (remask) ym = xh * ga[e] + be[e];
                    float gg = p.act != SSCG_ACT_NONE ? act_grad(dv[u][e], ym, p.act, p.slope) : dv[u][e];
                    gr[e] = gg;
                    float v = gg;
                    if (p.coef) v = gg - c1[e] - xh * c2[e];
                    gx[e] = v * rsv[e] * ga[e];
                }
                stv<T, VEC>(pdx + o, gx);
                if (pdres) stv<T, VEC>(pdres + o, gr);
            }
        }
    }
}

// ---- PixelDiscriminator tail (arch/discriminators.py:72-75): norm -> LeakyReLU -> Conv2d(C, 1, 1x1) in one pass over the C-channel
// map: out[r] = bias + sum_c w[c] * act(gamma * xhat + beta).  A row's C / 4 lanes (a power of two <= 64) hold four channels each
// and are combined by a butterfly; the normalised map is never written (the backward recomputes it from x).
struct HeadParams {
    const void* __restrict__ x;
    const float* __restrict__ mean;
    const float* __restrict__ rstd;
    const float* __restrict__ gamma;
    const float* __restrict__ beta;
    const float* __restrict__ w;
    const float* __restrict__ bias;
    float* __restrict__ out;
    long L;
    long rows;       // G * L
    int C;
    int act;
    float slope;
};

template <typename T>
__global__ __launch_bounds__(256) void norm_head_fwd_kernel(HeadParams p) {
    const int lanes = p.C >> 2;                    // lanes per row
    const int rows_per_wave = 64 / lanes;
    const int lane = threadIdx.x & 63;
    const int sub = lane / lanes;
    const int c = (lane - sub * lanes) * 4;
    const T* px = reinterpret_cast<const T*>(p.x);
    float w[4], ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    ld4<float>(p.w + c, w);
    if (p.gamma) { ld4<float>(p.gamma + c, ga); ld4<float>(p.beta + c, be); }
    const float b0 = p.bias ? p.bias[0] : 0.f;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    constexpr int U = 4;                           // rows in flight per lane
    for (long r0 = wave * rows_per_wave * U; r0 < p.rows; r0 += nwaves * rows_per_wave * U) {
        float xv[U][4], mu[U][4], rs[U][4];
        long r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = r0 + (long)u * rows_per_wave + sub;
            const long rr = r[u] < p.rows ? r[u] : p.rows - 1;
            const long g = rr / p.L;
            ld4<T>(px + (size_t)rr * p.C + c, xv[u]);
            ld4<float>(p.mean + (size_t)g * p.C + c, mu[u]);
            ld4<float>(p.rstd + (size_t)g * p.C + c, rs[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = (xv[u][e] - mu[u][e]) * rs[u][e];
                if (p.gamma) v = v * ga[e] + be[e];
                acc += w[e] * sscg_act(v, p.act, p.slope);
            }
            for (int m = 1; m < lanes; m <<= 1) acc += __shfl_xor(acc, m, 64);
            if (c == 0 && r[u] < p.rows) p.out[r[u]] = acc + b0;
        }
    }
}

__global__ void rstd_from_var_kernel(const float* __restrict__ var, float* __restrict__ rstd, int n, float eps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rstd[i] = (float)(1.0 / sqrt((double)var[i] + (double)eps));
}

inline int ew_grid(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}


// ---- second stage of the statistics a convolution epilogue left behind (conv_igemm.hip / conv_bf16.hip):
// records [rec][2 slots][C][2] doubles, rec = tile_m * RPT + wave_row; slot 0 = rows of the tile's first group
// (group of row tile_m * BMT), slot 1 = rows of the next group; plus `xrec` records [xrec][C][2] of the rows that went
// through split-K (all in group `xgroup`).  Record lanes are combined through LDS in a fixed order (deterministic).
struct ConvStatParams {
    const double* __restrict__ rec;
    const double* __restrict__ xrecs;
    float* __restrict__ mean;
    float* __restrict__ rstd;
    float* __restrict__ rmean;
    float* __restrict__ rvar;
    int valid_tiles;     // tile rows [0, valid_tiles) wrote records
    int BMT, RPT;        // rows per tile, records per tile
    int xrec, xgroup;
    int G, C;
    long L;
    float eps, momentum;
};

// block = 1024 threads = 16 channels x 64 record lanes (a 16-channel row segment of a record is 256 contiguous bytes);
// grid (C / 16, G or 1).  This kernel is a serial link (conv -> statistics -> normalise) in front of EVERY normalisation layer:
// with 16 record lanes a thread walked ~34 dependent-latency loads (29 us per launch, 16 ms per Cityscapes step); 64 lanes and
// four loads in flight per thread bring it to a handful of memory latencies.  CPB = 4 (256 record lanes, grid C / 4): few channels and
// many records - the 64-channel BatchNorm layers of DeepLab's stem and layer1 are 4 blocks of CPB = 16 walking 4096-8192 records,
// 47 us per launch 35 times per step (profiles/r04_experiments.txt item 14).
template <int CPB>
__global__ __launch_bounds__(1024) void finalize_conv_stats_kernel(ConvStatParams p) {
    constexpr int FCS_LANES = 1024 / CPB;
    __shared__ double sm[FCS_LANES][CPB][2];
    const int cl = threadIdx.x % CPB;
    const int w = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    const bool cok = c < p.C;
    const int g0 = p.rmean ? 0 : blockIdx.y;
    const int g1 = p.rmean ? p.G : blockIdx.y + 1;
    for (int g = g0; g < g1; ++g) {
        const long lo = (long)g * p.L, hi = lo + p.L;
        const int t_first = (int)((lo + p.BMT - 1) / p.BMT);
        int t_last = (int)((hi + p.BMT - 1) / p.BMT) - 1;
        if (t_last > p.valid_tiles - 1) t_last = p.valid_tiles - 1;
        double s = 0.0, q = 0.0;
        if (cok) {
            const int n0 = (t_last - t_first + 1) * p.RPT;          // slot-0 records of the tiles that start inside the group
            const double* base = p.rec + (((size_t)t_first * p.RPT * 2 + 0) * p.C + c) * 2;
            const size_t rstride = (size_t)2 * p.C * 2;             // doubles between consecutive records of one slot
            int k = w;
            for (; k + 3 * FCS_LANES < n0; k += 4 * FCS_LANES) {    // four independent loads in flight
                const double2 e0 = *reinterpret_cast<const double2*>(base + (size_t)k * rstride);
                const double2 e1 = *reinterpret_cast<const double2*>(base + (size_t)(k + FCS_LANES) * rstride);
                const double2 e2 = *reinterpret_cast<const double2*>(base + (size_t)(k + 2 * FCS_LANES) * rstride);
                const double2 e3 = *reinterpret_cast<const double2*>(base + (size_t)(k + 3 * FCS_LANES) * rstride);
                s += (e0.x + e1.x) + (e2.x + e3.x);
                q += (e0.y + e1.y) + (e2.y + e3.y);
            }
            for (; k < n0; k += FCS_LANES) {
                const double2 e = *reinterpret_cast<const double2*>(base + (size_t)k * rstride);
                s += e.x; q += e.y;
            }
            if (g > 0 && t_first >= 1 && t_first - 1 < p.valid_tiles) {  // slot 1 of the tile that straddles the lower boundary
                for (int k1 = w; k1 < p.RPT; k1 += FCS_LANES) {
                    const size_t r = (size_t)(t_first - 1) * p.RPT + k1;
                    const double2 e = *reinterpret_cast<const double2*>(p.rec + ((r * 2 + 1) * p.C + c) * 2);
                    s += e.x; q += e.y;
                }
            }
            if (p.xrec > 0 && p.xgroup == g) {
                for (int k2 = w; k2 < p.xrec; k2 += FCS_LANES) {
                    const double2 e = *reinterpret_cast<const double2*>(p.xrecs + ((size_t)k2 * p.C + c) * 2);
                    s += e.x; q += e.y;
                }
            }
        }
        sm[w][cl][0] = s; sm[w][cl][1] = q;
        __syncthreads();
#pragma unroll
        for (int st = FCS_LANES / 2; st >= 1; st >>= 1) {           // fixed-shape tree: deterministic
            if (w < st) { sm[w][cl][0] += sm[w + st][cl][0]; sm[w][cl][1] += sm[w + st][cl][1]; }
            __syncthreads();
        }
        if (w == 0 && cok) {
            const double a = sm[0][cl][0], b = sm[0][cl][1];
            const int i = g * p.C + c;
            const double m = a / (double)p.L;
            double var = b / (double)p.L - m * m;
            if (var < 0.0) var = 0.0;
            p.mean[i] = (float)m;
            p.rstd[i] = (float)(1.0 / sqrt(var + (double)p.eps));
            if (p.rmean) {
                const double unb = p.L > 1 ? var * (double)p.L / (double)(p.L - 1) : var;
                p.rmean[c] = (float)((1.0 - (double)p.momentum) * (double)p.rmean[c] + (double)p.momentum * m);
                p.rvar[c] = (float)((1.0 - (double)p.momentum) * (double)p.rvar[c] + (double)p.momentum * unb);
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t sscg_colsum_workspace(int64_t rows, int cols) { return part_bytes(1, rows, cols); }

extern "C" int sscg_colsum(const void* x, int dtype, float* out, int64_t rows, int cols, float beta, void* ws, size_t ws_bytes,
                           void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < part_bytes(1, rows, cols)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RedParams p = {};
    p.x = x; p.part = reinterpret_cast<double*>(ws); p.L = rows; p.C = cols;
    int rc = launch_reduce<RM_SUM>(p, 1, dtype, st);
    if (rc) return rc;
    RedPlan pl = plan_reduce(1, rows, cols, dtype);
    hipLaunchKernelGGL(finalize_sum_kernel, dim3(cdiv(cols, FIN_CH)), dim3(256), 0, st, p.part, out, cols, pl.chunks, beta);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" size_t sscg_norm_stats_workspace(int G, int64_t L, int C) { return part_bytes(G, L, C); }

extern "C" int sscg_norm_stats(const void* x, int dtype, int G, int64_t L, int C, float eps, float* mean, float* rstd,
                               float* running_mean, float* running_var, float momentum, void* ws, size_t ws_bytes,
                               void* stream) {
    if (!x || !mean || !rstd || G <= 0 || L <= 0 || C <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if ((running_mean != nullptr) != (running_var != nullptr)) return SSCG_ERR_BAD_ARG;
    if (!ws || ws_bytes < part_bytes(G, L, C)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RedParams p = {};
    p.x = x; p.part = reinterpret_cast<double*>(ws); p.L = L; p.C = C;
    int rc = launch_reduce<RM_STATS>(p, G, dtype, st);
    if (rc) return rc;
    RedPlan pl = plan_reduce(G, L, C, dtype);
    hipLaunchKernelGGL(finalize_stats_kernel, dim3(cdiv(C, FIN_CH), running_mean ? 1 : G), dim3(256), 0, st, p.part, mean, rstd,
                       running_mean, running_var, G, C, pl.chunks, (long)L, eps, momentum);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

int sscg_finalize_conv_stats(const double* stats, int valid_tiles, int rows_per_tile, int records_per_tile, const double* xrecs,
                             int xrec, int xgroup, int G, long L, int C, float eps, float* mean, float* rstd, float* running_mean,
                             float* running_var, float momentum, hipStream_t st) {
    ConvStatParams p = {};
    p.rec = stats; p.xrecs = xrecs; p.mean = mean; p.rstd = rstd; p.rmean = running_mean; p.rvar = running_var;
    p.valid_tiles = valid_tiles; p.BMT = rows_per_tile; p.RPT = records_per_tile; p.xrec = xrec; p.xgroup = xgroup;
    p.G = G; p.C = C; p.L = L; p.eps = eps; p.momentum = momentum;
    const int gy = running_mean ? 1 : G;
    const long recs = (L / rows_per_tile + 1) * records_per_tile;       // records per channel and group
#ifndef FCS_NARROW
#define FCS_NARROW 1      // (0: kernel-ablation build)
#endif
    if (FCS_NARROW && cdiv(C, 16) * gy < 64 && recs >= 512)
        hipLaunchKernelGGL(finalize_conv_stats_kernel<4>, dim3(cdiv(C, 4), gy), dim3(1024), 0, st, p);
    else
        hipLaunchKernelGGL(finalize_conv_stats_kernel<16>, dim3(cdiv(C, 16), gy), dim3(1024), 0, st, p);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

template <typename T>
static void launch_apply(const ApplyParams& p, int vec, hipStream_t st, int G = 0) {
    // rows in flight per thread (measured, profiles/r06_experiments.txt item 5): 8-channel bf16 groups 2 (130 -> 86 registers: config 3
    // -1 ms against 4), 4-channel groups 4; tuning aid SSCG_NORM_SLAB_U8A (0 = the flat kernel; 4-channel groups at 2 rows: no difference in the step)
    static const int u8 = getenv("SSCG_NORM_SLAB_U8A") ? atoi(getenv("SSCG_NORM_SLAB_U8A")) : 2;
    const int U = vec == 8 ? u8 : 4;
    const SlabPlan sp = (G > 0 && U > 0) ? plan_slab(G, p.L, p.C, vec, U) : SlabPlan{false, 0, 1, 1, 0};
    if (sp.ok) {
        const dim3 grid(sp.chunks, sp.slabs, G);
        if (vec == 8 && U == 2) hipLaunchKernelGGL((norm_apply_slab_kernel<T, 8, 2>), grid, dim3(256), 0, st, p, sp.cw_shift, sp.rows_per_chunk);
        else if (vec == 8) hipLaunchKernelGGL((norm_apply_slab_kernel<T, 8, 4>), grid, dim3(256), 0, st, p, sp.cw_shift, sp.rows_per_chunk);
        else hipLaunchKernelGGL((norm_apply_slab_kernel<T, 4, 4>), grid, dim3(256), 0, st, p, sp.cw_shift, sp.rows_per_chunk);
        return;
    }
    if (vec == 8) hipLaunchKernelGGL((norm_apply_kernel<T, 8>), dim3(ew_grid(p.total)), dim3(256), 0, st, p);
    else if (vec == 4) hipLaunchKernelGGL((norm_apply_kernel<T, 4>), dim3(ew_grid(p.total)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((norm_apply_kernel<T, 1>), dim3(ew_grid(p.total)), dim3(256), 0, st, p);
}

extern "C" int sscg_norm_apply(const void* x, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const void* residual, void* y, int dtype, int G, int64_t L, int C, int act,
                               float slope, void* stream) {
    if (!x || !mean || !rstd || !y || G <= 0 || L <= 0 || C <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if ((gamma != nullptr) != (beta != nullptr)) return SSCG_ERR_BAD_ARG;
    ApplyParams p = {};
    p.x = x; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.res = residual; p.y = y;
    p.L = L; p.C = C; p.act = act; p.slope = slope;
    const int vec = vec_for(C, dtype);
    if ((size_t)G * L * C / vec >= ((size_t)1 << 31)) return SSCG_ERR_UNSUPPORTED;
    p.total = (uint32_t)((size_t)G * L * C / vec);
    p.div_cg = make_fastdiv(C / vec);
    p.div_l = make_fastdiv((int)L);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SSCG_BF16) launch_apply<__bf16>(p, vec, st, G);
    else launch_apply<float>(p, vec, st, G);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_rstd_from_var(const float* var, float* rstd, int n, float eps, void* stream) {
    if (!var || !rstd || n <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(rstd_from_var_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, var, rstd, n, eps);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" size_t sscg_norm_bwd_workspace(int G, int64_t L, int C) {
    return part_bytes(G, L, C) + (size_t)G * C * 2 * sizeof(float);
}

template <typename T>
static void launch_bwd_apply(const BwdApplyParams& q, int vec, hipStream_t st, int G = 0) {
    // rows in flight (three loads each): 8-channel bf16 groups 1 (99 registers; with 4 rows - 210 registers - the kernel is slower than
    // the flat one), 4-channel groups 4; tuning aid SSCG_NORM_SLAB_U8B (0 = the flat kernel)
    static const int u8 = getenv("SSCG_NORM_SLAB_U8B") ? atoi(getenv("SSCG_NORM_SLAB_U8B")) : 1;
    const int U = vec == 8 ? u8 : 4;
    const SlabPlan sp = (G > 0 && U > 0) ? plan_slab(G, q.L, q.C, vec, U) : SlabPlan{false, 0, 1, 1, 0};
    if (sp.ok) {
        const dim3 grid(sp.chunks, sp.slabs, G);
        if (vec == 8 && U == 1) hipLaunchKernelGGL((norm_bwd_apply_slab_kernel<T, 8, 1>), grid, dim3(256), 0, st, q, sp.cw_shift, sp.rows_per_chunk);
        else if (vec == 8 && U == 2) hipLaunchKernelGGL((norm_bwd_apply_slab_kernel<T, 8, 2>), grid, dim3(256), 0, st, q, sp.cw_shift, sp.rows_per_chunk);
        else if (vec == 8) return launch_bwd_apply<T>(q, vec, st);      // (4 rows of 8 channels: 210 registers, slower than the flat kernel - not built)
        else hipLaunchKernelGGL((norm_bwd_apply_slab_kernel<T, 4, 4>), grid, dim3(256), 0, st, q, sp.cw_shift, sp.rows_per_chunk);
        return;
    }
    if (vec == 8) hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 8>), dim3(ew_grid(q.total)), dim3(256), 0, st, q);
    else if (vec == 4) hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 4>), dim3(ew_grid(q.total)), dim3(256), 0, st, q);
    else hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1>), dim3(ew_grid(q.total)), dim3(256), 0, st, q);
}

extern "C" int sscg_norm_bwd(const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, int dtype, int G,
                             int64_t L, int C, int act, float slope, int stats_grad, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !mean || !rstd || !dx || G <= 0 || L <= 0 || C <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    // y == NULL with ReLU / LeakyReLU: the mask is recomputed from x (only valid when no residual joined the forward);
    // tanh needs the forward value itself
    if (act == SSCG_ACT_TANH && !y) return SSCG_ERR_BAD_ARG;
    if (act != SSCG_ACT_NONE && !y && dres) return SSCG_ERR_BAD_ARG;      // a residual joined the forward: the mask needs y
    hipStream_t st = (hipStream_t)stream;
    const int overwrite = (stats_grad >> 1) & 1;      // bit 1: dgamma / dbeta are written, not accumulated
    stats_grad &= 1;
    const bool need_red = stats_grad || dgamma || dbeta;
    float* coef = nullptr;
    if (need_red) {
        if (!ws || ws_bytes < sscg_norm_bwd_workspace(G, L, C)) return SSCG_ERR_WORKSPACE;
        RedParams p = {};
        p.x = x; p.dy = dy; p.y = y; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta;
        p.part = reinterpret_cast<double*>(ws); p.L = L; p.C = C; p.act = act; p.slope = slope;
        int rc = launch_reduce<RM_BWD>(p, G, dtype, st);
        if (rc) return rc;
        RedPlan pl = plan_reduce(G, L, C, dtype);
        coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + part_bytes(G, L, C));
        hipLaunchKernelGGL(finalize_bwd_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, p.part, coef, dgamma, dbeta, G, C,
                           pl.chunks, (long)L, overwrite);
        SSCG_LAUNCH_CHECK();
    }
    BwdApplyParams q = {};
    q.dy = dy; q.x = x; q.y = y; q.mean = mean; q.rstd = rstd; q.gamma = gamma; q.beta = beta;
    q.coef = stats_grad ? coef : nullptr;
    q.dx = dx; q.dres = dres; q.L = L; q.C = C; q.act = act; q.slope = slope;
    const int vec = vec_for(C, dtype);
    if ((size_t)G * L * C / vec >= ((size_t)1 << 31)) return SSCG_ERR_UNSUPPORTED;
    q.total = (uint32_t)((size_t)G * L * C / vec);
    q.div_cg = make_fastdiv(C / vec);
    q.div_l = make_fastdiv((int)L);
    if (dtype == SSCG_BF16) launch_bwd_apply<__bf16>(q, vec, st, G);
    else launch_bwd_apply<float>(q, vec, st, G);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// Backward of y = act(norm(x) [+ residual]) (batch statistics) whose per-channel sums were already taken by the data gradient that
// produced dy (sscg_conv2d_dgrad_bsums with descriptor d): finalize + apply only - the reduction pass over (dy, x) is gone.
// y / dres: the unit's forward output (mask source) and the residual's gradient, for a unit a residual joined (sums taken with nz).
extern "C" int sscg_norm_bwd_from_sums(const sscg_conv_desc* d, const void* sums, const void* dy, const void* x, const void* y,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta, void* dx, void* dres,
                                       float* dgamma, float* dbeta, int dtype, int G, int64_t L, int C, int act, float slope, int flags,
                                       void* ws, size_t ws_bytes, void* stream) {
    if (!d || !sums || !dy || !x || !mean || !rstd || !dx || G <= 0 || L <= 0 || C <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if (act != SSCG_ACT_NONE && act != SSCG_ACT_RELU && act != SSCG_ACT_LRELU) return SSCG_ERR_UNSUPPORTED;
    if (act != SSCG_ACT_NONE && !y && dres) return SSCG_ERR_BAD_ARG;      // a residual joined the forward: the mask needs y
    int bm, wm, chunks;
    if (!sscg_bsums_records(d, G, L, &bm, &wm, &chunks) || d->C != C) return SSCG_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)G * C * 2 * sizeof(float)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* coef = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(finalize_bwd_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, reinterpret_cast<const double*>(sums), coef, dgamma,
                       dbeta, G, C, chunks, (long)L, (flags >> 1) & 1, bm, wm);
    SSCG_LAUNCH_CHECK();
    BwdApplyParams q = {};
    q.dy = dy; q.x = x; q.y = y; q.mean = mean; q.rstd = rstd; q.gamma = gamma; q.beta = beta;
    q.coef = coef;
    q.dx = dx; q.dres = dres; q.L = L; q.C = C; q.act = act; q.slope = slope;
    const int vec = vec_for(C, dtype);
    if ((size_t)G * L * C / vec >= ((size_t)1 << 31)) return SSCG_ERR_UNSUPPORTED;
    q.total = (uint32_t)((size_t)G * L * C / vec);
    q.div_cg = make_fastdiv(C / vec);
    q.div_l = make_fastdiv((int)L);
    if (dtype == SSCG_BF16) launch_bwd_apply<__bf16>(q, vec, st, G);
    else launch_bwd_apply<float>(q, vec, st, G);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

int sscg_finalize_bwd(const double* sums, float* coef, float* dgamma, float* dbeta, int G, int C, int chunks, long L, int overwrite, int rec_bm,
                      int rec_wm, hipStream_t st) {
    hipLaunchKernelGGL(finalize_bwd_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, sums, coef, dgamma, dbeta, G, C, chunks, L, overwrite,
                       rec_bm, rec_wm);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// ---- PixelDiscriminator tail: norm -> activation -> 1x1 conv to ONE channel (arch/discriminators.py:72-75)
static bool head_ok(int C) { return C >= 16 && C <= 256 && (C & (C - 1)) == 0; }

extern "C" int sscg_norm_head_applies(int C) { return head_ok(C) ? 1 : 0; }

extern "C" int sscg_norm_head_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* w, const float* bias, float* out, int G, int64_t L, int C, int act, float slope,
                                  void* stream) {
    if (!x || !mean || !rstd || !w || !out || G <= 0 || L <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if ((gamma != nullptr) != (beta != nullptr)) return SSCG_ERR_BAD_ARG;
    if (!head_ok(C)) return SSCG_ERR_UNSUPPORTED;
    HeadParams p = {};
    p.x = x; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta; p.w = w; p.bias = bias; p.out = out;
    p.L = L; p.rows = (long)G * L; p.C = C; p.act = act; p.slope = slope;
    const long rows_per_block = 4L * (256 / C) * 4;       // 4 waves x rows per wave x rows in flight
    long blocks = cdiv(p.rows, rows_per_block);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SSCG_BF16) hipLaunchKernelGGL(norm_head_fwd_kernel<__bf16>, dim3((int)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(norm_head_fwd_kernel<float>, dim3((int)blocks), dim3(256), 0, st, p);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" size_t sscg_norm_head_bwd_workspace(int G, int64_t L, int C) {
    return 2 * part_bytes(G, L, C) + (size_t)G * C * 2 * sizeof(float);
}

// flags: bit 0 = the statistics are functions of x (training-mode norm), bit 1 = dgamma / dbeta written (else accumulated),
// bit 2 = dw / dbias written (else accumulated)
extern "C" int sscg_norm_head_bwd(const float* dout, const float* w, const void* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, void* dx, float* dw, float* dbias, float* dgamma,
                                  float* dbeta, int dtype, int G, int64_t L, int C, int act, float slope, int flags, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (!dout || !w || !x || !mean || !rstd || !dx || G <= 0 || L <= 0 || (dtype != SSCG_F32 && dtype != SSCG_BF16)) return SSCG_ERR_BAD_ARG;
    if (act == SSCG_ACT_TANH) return SSCG_ERR_UNSUPPORTED;      // the mask / value is recomputed from x
    if (!head_ok(C)) return SSCG_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < sscg_norm_head_bwd_workspace(G, L, C)) return SSCG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int stats_grad = flags & 1, overwrite = (flags >> 1) & 1, overwrite_w = (flags >> 2) & 1;
    RedParams p = {};
    p.x = x; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.beta = beta;
    p.part = reinterpret_cast<double*>(ws); p.L = L; p.C = C; p.act = act; p.slope = slope;
    p.head_w = w; p.head_dout = dout;
    p.head_part = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + part_bytes(G, L, C));
    int rc = launch_reduce<RM_BWD_HEAD>(p, G, dtype, st);
    if (rc) return rc;
    RedPlan pl = plan_reduce(G, L, C, dtype);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 2 * part_bytes(G, L, C));
    hipLaunchKernelGGL(finalize_bwd_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, p.part, coef, dgamma, dbeta, G, C, pl.chunks,
                       (long)L, overwrite);
    SSCG_LAUNCH_CHECK();
    if (dw || dbias) {
        hipLaunchKernelGGL(finalize_head_kernel, dim3(cdiv(C, FIN_CH)), dim3(256), 0, st, p.head_part, dw, dbias, C, G * pl.chunks,
                           overwrite_w ? 0.f : 1.f);
        SSCG_LAUNCH_CHECK();
    }
    BwdApplyParams q = {};
    q.x = x; q.mean = mean; q.rstd = rstd; q.gamma = gamma; q.beta = beta;
    q.coef = stats_grad ? coef : nullptr;
    q.head_w = w; q.head_dout = dout;
    q.dx = dx; q.L = L; q.C = C; q.act = act; q.slope = slope;
    const int vec = vec_for(C, dtype);
    if ((size_t)G * L * C / vec >= ((size_t)1 << 31)) return SSCG_ERR_UNSUPPORTED;
    q.total = (uint32_t)((size_t)G * L * C / vec);
    q.div_cg = make_fastdiv(C / vec);
    q.div_l = make_fastdiv((int)L);
    if (dtype == SSCG_BF16) launch_bwd_apply<__bf16>(q, vec, st, G);
    else launch_bwd_apply<float>(q, vec, st, G);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}
