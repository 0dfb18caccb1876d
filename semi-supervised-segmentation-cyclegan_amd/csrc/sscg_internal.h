// Internal glue: public ABI types + host functions shared between translation units (device code is not linked
// across files; only host-side launchers cross them).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sscg.h"

struct sscg_bsums;
// conv_bf16.hip: bf16 MFMA kernels behind sscg_conv2d_{fwd,dgrad,wgrad} (dispatch lives in conv_igemm.hip / conv_wgrad.hip)
bool sscg_conv16_fwd_applies(const sscg_conv_desc* d);
bool sscg_conv16_dgrad_applies(const sscg_conv_desc* d);
bool sscg_conv16_stats_geometry(const sscg_conv_desc* d, long L, int* bm, int* wm, int* tiles_n, int* splits, int* full_tiles, int* m_tail0);
size_t sscg_conv16_fwd_workspace(const sscg_conv_desc* d, long stat_L);
size_t sscg_conv16_dgrad_workspace(const sscg_conv_desc* d);
int sscg_conv16_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* stats, long stat_L,
                    double* xstats, void* ws, size_t ws_bytes, hipStream_t st);
int sscg_conv16_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias, void* dx, int act, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, const sscg_bsums* bs = nullptr, const void* addend = nullptr);
bool sscg_conv16_dgrad_add_applies(const sscg_conv_desc* d);
bool sscg_conv16_bsums_geometry(const sscg_conv_desc* d, int G, long L, int* bm, int* wm, int* chunks);
bool sscg_wgrad16_applies(const sscg_conv_desc* d);
size_t sscg_wgrad16_workspace(const sscg_conv_desc* d);
int sscg_wgrad16(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, size_t ws_bytes, hipStream_t st);
// conv_split.hip: fp32-accurate "split" contraction on the bf16 matrix cores (fp32 activations, w_dtype == SSCG_BF16X3)
bool sscg_convs_fwd_applies(const sscg_conv_desc* d);
bool sscg_convs_dgrad_applies(const sscg_conv_desc* d);
bool sscg_convs_stats_geometry(const sscg_conv_desc* d, long L, int* bm, int* wm, int* tiles_n, int* splits, int* full_tiles, int* m_tail0);
size_t sscg_convs_fwd_workspace(const sscg_conv_desc* d, long stat_L);
size_t sscg_convs_dgrad_workspace(const sscg_conv_desc* d);
int sscg_convs_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* stats, long stat_L,
                   double* xstats, void* ws, size_t ws_bytes, hipStream_t st);
// the forward with the 1x1 conv + LeakyReLU in front of it (cin -> 64 channels) formed in the prologue (PixelDiscriminator's front half)
bool sscg_convs_front_applies(const sscg_conv_desc* d, int cin);
int sscg_convs_fwd_front(const sscg_conv_desc* d, const void* xf, int cin, const float* w1, const float* b1, float slope1, void* h1,
                         const void* w, const float* bias, void* y, double* stats, long stat_L, hipStream_t st);
// the backward sums of the normalisation layer whose output a data gradient differentiates, taken in that launch's epilogue
struct sscg_bsums {
    const void* nx;          // the layer's input [G * L][C]
    const void* nz;          // the layer's OUTPUT (mask source of a unit a residual joined), or null: the mask is recomputed from nx
    const float* mean;       // [G][C]
    const float* rstd;
    const float* gamma;      // [C] or null
    const float* beta;
    void* sums;              // [G][chunks][C][2] doubles (sscg_convs_bsums_geometry)
    int G;
    long L;
    int act;
    float slope;
};
bool sscg_convs_bsums_geometry(const sscg_conv_desc* d, int G, long L, int* bm, int* wm, int* chunks);
int sscg_convs_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias, void* dx, int act, float slope,
                     void* ws, size_t ws_bytes, hipStream_t st, const sscg_bsums* bs = nullptr, const void* addend = nullptr);
// conv_igemm.hip: record geometry of sscg_conv2d_dgrad_bsums for this descriptor (false = the fusion does not apply)
bool sscg_bsums_records(const sscg_conv_desc* d, int G, int64_t L, int* bm, int* wm, int* chunks);
int sscg_krsc_to_crsk_split(const float* w, void* wt, int K, int RS, int C, hipStream_t st);
// weight gradient by the split contraction on pre-split scratch planes (precision = 2, >= 128 x 128 outputs)
bool sscg_wgrads_applies(const sscg_conv_desc* d);
size_t sscg_wgrads_workspace(const sscg_conv_desc* d);
int sscg_wgrads(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, size_t ws_bytes, hipStream_t st);
// conv_thin.hip: HBM-streaming kernels for 1x1 convolutions with a handful of channels on one side (PixelDiscriminator ends)
bool sscg_thin1x1_fwd_applies(const sscg_conv_desc* d);
int sscg_thin1x1_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, hipStream_t st);
bool sscg_thin1x1_dgrad_applies(const sscg_conv_desc* d, const float* bias, int act);
int sscg_thin1x1_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, void* dx, hipStream_t st);
// conv_wgrad.hip: dw = beta * dw + sum_s ws[s] (fixed order)
int sscg_wgrad_reduce(const float* ws, float* dw, size_t n, int splits, float beta, hipStream_t st);
// norm.hip: the backward sums a data gradient's epilogue left behind -> coefficients (c1, c2) per (g, c), dgamma / dbeta
int sscg_finalize_bwd(const double* sums, float* coef, float* dgamma, float* dbeta, int G, int C, int chunks, long L, int overwrite, int rec_bm,
                      int rec_wm, hipStream_t st);
// norm.hip: the statistics a conv epilogue left behind -> mean / rstd (+ running statistics); the rows that went through
// split-K arrive as `xrec` extra records written by the split reduction (reduce_common.h)
int sscg_finalize_conv_stats(const double* stats, int valid_tiles, int rows_per_tile, int records_per_tile, const double* xrecs,
                             int xrec, int xgroup, int G, long L, int C, float eps, float* mean, float* rstd, float* running_mean,
                             float* running_var, float momentum, hipStream_t st);
