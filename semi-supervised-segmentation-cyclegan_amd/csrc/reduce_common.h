// Second stage of a tail-split / split-K convolution launch, shared by conv_igemm.hip and conv_bf16.hip (device code is
// not linked across translation units: this header is instantiated in each).
#pragma once
#include "common.h"

// y[r][c] = act(sum_s part[s][r][c] + bias[c]) over the `rows` rows that went through split-K (fixed order =>
// deterministic), AND the column statistics of the pre-activation values for the fused normalisation statistics:
// one record [Ng][2] (sum, sum of squares; fp64) per block of `rows_per_block` rows.  V channels per thread.
template <int V>
__global__ __launch_bounds__(256) void split_reduce_stats_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                  void* __restrict__ y, int out_bf16, int rows, int Ng, int splits,
                                                                  int act, float slope, double* __restrict__ recs, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    double* sm = reinterpret_cast<double*>(sm_raw);          // [256][V][2]
    const int cgs = Ng / V;                                  // channel groups of the tensor
    const int cg_blk = cgs < 256 ? cgs : 256;                // ... owned by one block
    const int lanes = 256 / cg_blk;                          // row lanes of the block
    const int cl = threadIdx.x % cg_blk;
    const int rl = threadIdx.x / cg_blk;
    const int cg = blockIdx.y * cg_blk + cl;
    const bool on = cg < cgs && rl < lanes;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    const size_t n = (size_t)rows * Ng;
    double s[V], q[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s[e] = 0.0; q[e] = 0.0; }
    if (on) {
        float bv[V];
#pragma unroll
        for (int e = 0; e < V; ++e) bv[e] = bias ? bias[cg * V + e] : 0.f;
        for (int r = r0 + rl; r < r1; r += lanes) {
            const size_t i = (size_t)r * Ng + (size_t)cg * V;
            float a[V];
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] = 0.f;
#pragma unroll 8
            for (int k = 0; k < splits; ++k) {
                float t[V];
                if constexpr (V == 4) ld4<float>(part + (size_t)k * n + i, t); else t[0] = part[(size_t)k * n + i];
#pragma unroll
                for (int e = 0; e < V; ++e) a[e] += t[e];
            }
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float pre = a[e] + bv[e];
                const double d = (double)pre;
                s[e] += d;
                q[e] += d * d;
                o[e] = sscg_act(pre, act, slope);
            }
            if (out_bf16) {
                __bf16* yb = reinterpret_cast<__bf16*>(y) + i;
                if constexpr (V == 4) st4<__bf16>(yb, o); else st1<__bf16>(yb, o[0]);
            } else {
                float* yf = reinterpret_cast<float*>(y) + i;
                if constexpr (V == 4) st4<float>(yf, o); else st1<float>(yf, o[0]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) { sm[(threadIdx.x * V + e) * 2] = s[e]; sm[(threadIdx.x * V + e) * 2 + 1] = q[e]; }
    __syncthreads();
    if (on && rl == 0) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            double a = 0.0, b = 0.0;
            for (int l = 0; l < lanes; ++l) {
                a += sm[((l * cg_blk + cl) * V + e) * 2];
                b += sm[((l * cg_blk + cl) * V + e) * 2 + 1];
            }
            double* rec = recs + ((size_t)blockIdx.x * Ng + (size_t)cg * V + e) * 2;
            rec[0] = a;
            rec[1] = b;
        }
    }
}

// Rows per block (= per statistics record) of the split reduction: one row per row-lane, at least 4.  The split rows are
// a few hundred to a thousand; the reduction sits on the critical path of its layer (conv -> statistics -> normalise), so
// it is spread over >= 100 workgroups rather than kept compact.
static inline int split_stats_rpb(int Ng) {
    const int cgs = Ng % 4 == 0 ? Ng / 4 : Ng;
    const int lanes = 256 / (cgs < 256 ? cgs : 256);
    return lanes > 4 ? lanes : 4;
}

static inline int split_stats_records(long rows, int Ng) {
    const int rpb = split_stats_rpb(Ng);
    return (int)((rows + rpb - 1) / rpb);
}

static inline int launch_split_reduce_stats(const float* part, const float* bias, void* y, int out_bf16, long rows, int Ng, int splits,
                                            int act, float slope, double* recs, hipStream_t st) {
    const bool v4 = Ng % 4 == 0;      // (y and part are 16-byte aligned: allocator granularity, and m_tail0 * Ng * 4 bytes with Ng % 4 == 0)
    const int cgs = v4 ? Ng / 4 : Ng;
    const int cg_blk = cgs < 256 ? cgs : 256;
    const int rpb = split_stats_rpb(Ng);
    dim3 grid(split_stats_records(rows, Ng), (cgs + cg_blk - 1) / cg_blk);
    const size_t smem = (size_t)256 * (v4 ? 4 : 1) * 2 * sizeof(double);
    if (v4)
        hipLaunchKernelGGL(split_reduce_stats_kernel<4>, grid, dim3(256), smem, st, part, bias, y, out_bf16, (int)rows, Ng, splits, act,
                           slope, recs, rpb);
    else
        hipLaunchKernelGGL(split_reduce_stats_kernel<1>, grid, dim3(256), smem, st, part, bias, y, out_bf16, (int)rows, Ng, splits, act,
                           slope, recs, rpb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
