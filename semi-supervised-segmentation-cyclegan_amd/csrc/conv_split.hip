// fp32 convolutions on the BF16 matrix cores with fp32 accuracy ("split" contraction; sscg_conv_desc.w_dtype == SSCG_BF16X3).
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate.  Here every fp32 operand is written as three bfloat16
// pieces x = h + m + l (round-to-nearest each time; the residuals x - h and x - h - m are exact in fp32, so the sum is x to
// 2^-24) and a product a * b is the six piece products a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0, each EXACT (8-bit x 8-bit
// mantissas) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (smallest terms first); the three terms left out are below
// 2^-23 |a b|.  Six MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 16 k: 2.67x fewer matrix-core cycles, at or
// below the rounding error of the exact-fp32 kernel (conv_igemm.hip) against fp64 (tests/test_kernels_gpu.py).
//
// Same products as conv_igemm.hip (reference call sites: arch/ops.py:43,49,55-56,68; arch/generators.py:325-336,373,388,415;
// arch/discriminators.py:45,58,70-75 and their autograd, model.py:472,539):
//   forward / data gradient / ConvTranspose forward : D[m][n] = sum_k A[m][k] * B[n][k]
// Operands:
//   A = the activation (fp32 NHWC in HBM, gathered on the fly exactly like conv_igemm.hip's fast path): it goes to LDS as it
//       is, by LDS-DMA (128-byte rows of 32 k), and is split in registers between LDS and the matrix cores;
//   B = the weight, split ONCE per optimiser step (sscg_split3 / the Adam kernel's shadow / the transposing copy): three bf16
//       planes in HBM, three 64-byte rows per output channel and k-tile in LDS - a lane's ds_read_b128 is a complete MFMA operand.
// A wave owns the 32 x (TN * 32) or (TM * 32) x (TN * 32) corner of the block tile; each A fragment is split by the waves of
// one tile row only (WN waves), ~5.5 VALU operations per element, hidden under the 6 * TN MFMAs it feeds.
//
// The k-loop is software-pipelined across k-tiles with ONE barrier per tile: while the MFMAs of half-tile (t, 0) run, the
// fragments of (t, 1) are read and split; then tile t+1 must have landed (vmcnt + barrier), the copy of tile t+2 is requested
// into the stage just drained, and the fragments of (t+1, 0) are read and split under the MFMAs of (t, 1).  LDS accesses and
// waits inside the loop are hand-placed (the compiler would order every visible LDS read behind the copies' vmcnt(0)).
#include "common.h"
#include "sscg_internal.h"
#include "reduce_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

#ifndef KS_ABLATE
#define KS_ABLATE 0        // tools/: timing ablations of the k-loop (1 = no copies after the prologue, 2 = no operand split, 4 = no barrier / copy wait, 8 = no epilogue); WRONG results
#endif
#ifndef KS_BUFLD
#define KS_BUFLD 1         // LDS-DMA copies as `buffer_load_dwordx4 ... offen lds`: per-lane row offset in ONE VGPR (computed once per tap), the k-tile's
#endif                     // chunk offset in an SGPR - no 64-bit VALU address arithmetic per copy (0 = global_load_lds with 64-bit addresses)
#ifndef KS_ACC2
#define KS_ACC2 1          // leading piece product and the five small ones in separate accumulators (wave tiles <= 32 x 64)
#endif
#ifndef KS_LATE_WAIT
#define KS_LATE_WAIT 0
#endif
#ifndef KS_LB4
#define KS_LB4 1           // 64x64 class: hold the kernel to 128 registers (4 workgroups per CU, what its 40 KB of LDS allow) - two accumulator sets take it to 134
#endif
constexpr int BKS = 32;                    // k per tile
typedef __bf16 bf16;
typedef uint32_t u32;

// Zeros: the source of masked LDS-DMA lanes (padding taps, rows past M).  A masked row's pointer is chosen once per tap and then
// advanced by the channel-chunk offset like any other (one 64-bit add per copy piece, no per-tile select): the page covers the
// largest channel offset of a row (Cs <= 4096 fp32 = 16 KB).
constexpr int KS_ZERO_FLOATS = 4096 + 64;
__device__ float sscg_zero_page_s[KS_ZERO_FLOATS];

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// eight fp32 (two LDS fragments) -> three bf16x8 pieces
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    sscg_split8(x, h, m, l);
}

struct KsParams {
    const float* __restrict__ src;   // A source (input for fwd, dy for dgrad), fp32 NHWC
    const bf16* __restrict__ wgt;    // plane 0 of the split weight: [3][Ng][wKtot] bf16, planes `wplane` elements apart
    long wplane;
    const float* __restrict__ bias;  // [Ng] or null
    float* __restrict__ dst;         // [M][Ng] fp32
    int M, Ng, Ktot, Cs;
    int SH, SW, OH, OW;
    int R, S;
    int stride, pad, dil, pad_x;
    int wKtot;
    int wt_ky0, wt_kx0, wt_step, wt_S;
    int o_step, o_a, o_b, o_W, o_HW;
    int pad_mode, act;
    float slope;
    int tiles_n, tiles;
    int splits, ksplit, full_tiles, m_tail0;
    float* __restrict__ part;        // [splits][M - m_tail0][Ng] when splits > 1
    double* __restrict__ stats;      // fused normalisation statistics: [tiles_m * WM][2][Ng][2] doubles, or null
    int stat_L;
    double* __restrict__ xstats;     // host side
    unsigned src_bytes, wgt_bytes;   // extents of the two buffer resources (< 2 GB: a masked row's offset 0x80000000 is out of range = zeros)
    // data gradient only: the backward sums of the normalisation layer whose OUTPUT this launch differentiates (sscg_conv2d_dgrad_bsums).
    // dst is dz; per channel n and group g:  sum gg,  sum gg * xhat  with  xhat = (nx - mean) * rstd,  gg = act'(gamma xhat + beta) dz
    const float* __restrict__ bn_x;      // [M][Ng] the layer's input (pre-normalisation), or null
    const float* __restrict__ bn_z;      // [M][Ng] the layer's OUTPUT act(norm(bn_x) + residual): the mask source of a unit a residual joined, or null
    const float* __restrict__ addend;    // [M][Ng] added to the result (the gradient another consumer of the same tensor left), or null
    const float* __restrict__ bn_mean;   // [G][Ng]
    const float* __restrict__ bn_rstd;
    const float* __restrict__ bn_gamma;  // [Ng] or null
    const float* __restrict__ bn_beta;
    double* __restrict__ bn_sums;        // [G][bn_chunks][Ng][2]
    int bn_L, bn_G, bn_chunks, bn_act;
    float bn_slope;
    // fused front conv (CIN > 0; PixelDiscriminator, arch/discriminators.py:70-71): the A operand is not read from memory - row m of it is
    // lrelu(fr_b1 + fr_w1 . fr_x[m]), the 1x1 conv (CIN -> Cs = 64 channels) + LeakyReLU in front of this one, formed in the prologue
    const float* __restrict__ fr_x;      // [M][CIN] fp32
    const float* __restrict__ fr_w1;     // [Cs][CIN]
    const float* __restrict__ fr_b1;     // [Cs] or null
    float fr_slope;
    float* __restrict__ fr_h1;           // [M][Cs] or null: the front conv's output is ALSO written (a backward pass that wants it stored)
    FastDiv div_tn, div_gl;              // by tiles_n; by stat_L / bn_L (whichever the launch uses)
    FastDiv div_hw, div_w;               // by OH * OW and by OW (launch_ks): a row's (image, y, x) without integer divisions (~30 VALU operations each)
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(bf16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// activations of this family's epilogue: none / ReLU / LeakyReLU; tanh (the ResNet generators' 3- and 21-channel heads) only in the
// heads' 32-column class - its software expansion would sit 32-64 times in every other instance's epilogue (sscg_convs_fwd_applies)
__device__ __forceinline__ float ks_act(float v, int act, float slope) {
    const float neg = act == SSCG_ACT_RELU ? 0.f : (act == SSCG_ACT_LRELU ? v * slope : v);
    return v > 0.f ? v : neg;
}

template <int MODE, int WM, int WN, int TM, int TN, int CIN = 0>
__global__ __launch_bounds__(WM * WN * 64, ((KS_LB4 && TM * TN == 1) ? 4 : 2)) void convs_kernel(KsParams p) {      // (HIP: the second figure is WAVES PER SIMD)
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NSTAGE = 2;
    constexpr int RPA = NT / 8;               // A rows per loader pass (8 lanes x 16 B per 128-byte row)
    constexpr int RPB = (NT / 4 < BN) ? NT / 4 : BN;   // B rows per loader pass (4 lanes x 16 B per 64-byte row); a tile narrower than a pass
                                              // (BN = 32): the upper waves copy the same rows to the same place again
    constexpr int PA = BM / RPA;
    constexpr int HB = BN / RPB;              // passes per weight plane
    constexpr int PB = 3 * HB;
    static_assert(BM % RPA == 0 && BN % RPB == 0, "whole loader passes");
    constexpr int A_STAGE = BM * 128;         // bytes
    constexpr int B_PLANE = BN * 64;
    constexpr int B_STAGE = 3 * B_PLANE;
    constexpr bool FRONT = CIN > 0;           // the A tiles (both k-tiles of a 64-channel reduction) are computed, not copied
    constexpr int NPIECE = (FRONT ? 0 : PA) + PB;
    static_assert(!FRONT || (MODE == MODE_FWD && (BM == 128 || BM == 64) && NT == 256), "fused front conv: forward, 64- or 128-row tiles, four waves");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [2] A images, then [2][3] B plane images

    const int tid = (int)threadIdx.x;
    int split = 0, tile;
    bool partial = false;
    if ((int)blockIdx.x < p.full_tiles) {
        tile = xcd_remap(blockIdx.x, p.full_tiles);
    } else {
        const int ntail = p.tiles - p.full_tiles;
        const int t = xcd_remap(blockIdx.x - p.full_tiles, gridDim.x - p.full_tiles);
        split = t / ntail;
        tile = p.full_tiles + (t - split * ntail);
        partial = p.splits > 1;
    }
    const int tile_m = fd_div(tile, p.div_tn);
    const int tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int wave_id = tid >> 6;

    // ---- copy side
    // OWN_A (the 32x128-wave class): a wave copies exactly the 32 A rows its own MFMAs consume (pass ps = rows 32 w + 8 ps .. + 7), so the A
    // side of the pipeline needs no workgroup barrier and its copies go out as soon as the wave itself has read the stage (below)
    constexpr bool OWN_A = TM == 1 && TN == 4 && WM == 4 && WN == 1 && CIN == 0;
    const int r0 = tid >> 3;                                   // A row inside a pass
    const int kqa = (tid & 7) ^ ((r0 >> 1) & 7);               // 16-byte k slot this lane fetches (lands at LDS slot tid & 7 of row r0)
    auto a_row = [&](int ps) -> int { return OWN_A ? (tid >> 6) * 32 + ps * 8 + ((tid & 63) >> 3) : r0 + ps * RPA; };
    auto a_kq = [&](int ps) -> int { return OWN_A ? ((tid & 7) ^ ((ps * 4 + ((tid & 63) >> 4)) & 7)) : kqa; };
    const int rb0 = (tid >> 2) % RPB;                          // B row inside a pass
    const int kqb = (tid & 3) ^ ((rb0 >> 2) & 3);

#if KS_BUFLD
    unsigned arow[PA];                                         // byte offset of the row's image inside src
#else
    const float* arow[PA];
#endif
    int ay0[PA], ax0[PA];
    bool aok[PA];
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
        const int m = m0 + a_row(ps);
        aok[ps] = m < p.M;
        const int mm = aok[ps] ? m : 0;
        const int img = fd_div(mm, p.div_hw);
        const int rem = mm - img * (p.OH * p.OW);
        const int oy = fd_div(rem, p.div_w);
        const int ox = rem - oy * p.OW;
#if KS_BUFLD
        arow[ps] = (unsigned)img * (unsigned)(p.SH * p.SW * p.Cs) * 4u;
#else
        arow[ps] = p.src + (size_t)img * p.SH * p.SW * p.Cs;
#endif
        if (MODE == MODE_FWD) {
            ay0[ps] = oy * p.stride - p.pad;
            ax0[ps] = ox * p.stride - p.pad_x;
        } else {
            ay0[ps] = oy + p.pad;
            ax0[ps] = ox + p.pad_x;
        }
    }
    // weight rows past Ng are clamped to the last row: their products land in output columns that are never stored
#if KS_BUFLD
    unsigned brow[HB];                                         // byte offset of the lane's 16-byte piece of its weight row (plane 0, k = 0)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, (short)0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, (short)0, (int)p.wgt_bytes, 0x00020000);
#else
    const bf16* brow[HB];
#endif
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        const int n = n0 + rb0 + hb * RPB;
#if KS_BUFLD
        brow[hb] = ((unsigned)(n < p.Ng ? n : p.Ng - 1) * (unsigned)p.wKtot + kqb * 8) * 2u;
#else
        brow[hb] = p.wgt + (size_t)(n < p.Ng ? n : p.Ng - 1) * p.wKtot + kqb * 8;
#endif
    }

    const bool reflect = p.pad_mode == 1;
    auto locate = [&](int ps, int tdy, int tdx, int& pix) -> bool {
        bool ok = aok[ps];
        int sy, sx;
        if (MODE == MODE_FWD) {
            sy = ay0[ps] + tdy;
            sx = ax0[ps] + tdx;
            if (reflect) {              // (uniform: only the ResNet generators' reflection-padded convolutions pay for the mirror arithmetic)
                int ry = sy < 0 ? -sy : sy;
                int rx = sx < 0 ? -sx : sx;
                sy = ry >= p.SH ? 2 * (p.SH - 1) - ry : ry;
                sx = rx >= p.SW ? 2 * (p.SW - 1) - rx : rx;
            }
        } else {
            const int ty = ay0[ps] - tdy;
            const int tx = ax0[ps] - tdx;
            if (p.stride == 1) {
                sy = ty; sx = tx;
            } else {                    // stride 2 (sscg_convs_dgrad_applies: larger strides stay on conv_igemm.hip's kernel)
                sy = ty >> 1; sx = tx >> 1;
                ok = ok && (((ty | tx) & 1) == 0);
            }
        }
        ok = ok && ((unsigned)sy < (unsigned)p.SH) && ((unsigned)sx < (unsigned)p.SW);
        pix = ok ? sy * p.SW + sx : 0;
        return ok;
    };

    const int nk_all = p.Ktot / BKS;                         // Cs % 32 == 0: a k-tile never straddles a tap
    const int kt0 = partial ? split * p.ksplit : 0;
    const int kt1 = partial ? min(nk_all, kt0 + p.ksplit) : nk_all;
    const int f_nchunk = p.Cs / BKS;
    int f_chunk, f_ky, f_kx;
    if (kt0 == 0) {                     // (every whole tile: the two integer divisions below are ~30 VALU operations)
        f_chunk = 0; f_ky = 0; f_kx = 0;
    } else {
        const int tap0 = kt0 / f_nchunk;
        f_chunk = kt0 - tap0 * f_nchunk;
        f_ky = tap0 / (p.S > 0 ? p.S : 1);
        f_kx = tap0 - f_ky * p.S;
    }
    int f_k = 0;
#if KS_BUFLD
    unsigned aptr[PA];
#else
    const float* aptr[PA];
    const float* const zero = sscg_zero_page_s;
#endif
    int dma_stage = 0;

    auto set_tap = [&]() {
        const int tdy = f_ky * p.dil, tdx = f_kx * p.dil;
        f_k = ((p.wt_ky0 + f_ky * p.wt_step) * p.wt_S + p.wt_kx0 + f_kx * p.wt_step) * p.Cs;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            int pix;
            const bool ok = locate(ps, tdy, tdx, pix);
#if KS_BUFLD
            aptr[ps] = ok ? arow[ps] + ((unsigned)pix * (unsigned)p.Cs + a_kq(ps) * 4) * 4u : 0x80000000u;      // out of range: the copy delivers zeros
#else
            aptr[ps] = ok ? arow[ps] + (size_t)pix * p.Cs + a_kq(ps) * 4 : zero + a_kq(ps) * 4;
#endif
        }
    };
    set_tap();
    f_k += f_chunk * BKS;

    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;
    const int lds_wave = __builtin_amdgcn_readfirstlane(wave_id * 1024);      // this wave's 1 KB of every loader pass
    const int lds_wave_b = __builtin_amdgcn_readfirstlane((wave_id % (RPB / 16)) * 1024);
    auto request_tile = [&]() {
        if (f_chunk == f_nchunk) {       // wave-uniform: next tap
            f_chunk = 0;
            ++f_kx;
            if (f_kx == p.S) { f_kx = 0; ++f_ky; }
            if (f_ky >= p.R) { f_ky = 0; f_kx = 0; }      // past the last tap (never requested)
            set_tap();
        }
        if ((KS_ABLATE & 1) && f_k > 2 * BKS) { dma_stage ^= 1; ++f_chunk; f_k += BKS; return; }
#if KS_BUFLD
        if constexpr (!FRONT) {
            const int so_a = __builtin_amdgcn_readfirstlane(f_chunk * (BKS * 4));
#pragma unroll
            for (int q = 0; q < PA; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave + q * (RPA * 128)),
                                                         16, (int)aptr[q], so_a, 0, 0);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int so_b = __builtin_amdgcn_readfirstlane((int)((pl * p.wplane + f_k) * 2));
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE +
                                                                                                        pl * B_PLANE + lds_wave_b + hb * (RPB * 64)),
                                                         16, (int)brow[hb], so_b, 0, 0);
        }
#else
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const float* g = aptr[q] + f_chunk * BKS;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave + q * (RPA * 128)), 16, 0, 0);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                const bf16* g = brow[hb] + (pl * p.wplane + f_k);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE + pl * B_PLANE +
                                                                                            lds_wave_b + hb * (RPB * 64)), 16, 0, 0);
            }
        }
#endif
        dma_stage ^= 1;
        ++f_chunk;
        f_k += BKS;
    };
    // OWN_A: the two operands are requested separately - A by the tap walker above (request_a), B by a walker of its own (request_b)
    int b_chunk = f_chunk, b_ky = f_ky, b_kx = f_kx, b_stage = 0;
    int b_k = ((p.wt_ky0 + b_ky * p.wt_step) * p.wt_S + p.wt_kx0 + b_kx * p.wt_step) * p.Cs + b_chunk * BKS;
    const int lds_own_a = __builtin_amdgcn_readfirstlane(wave_id * 4096);
    auto request_a = [&]() {
        if (f_chunk == f_nchunk) {
            f_chunk = 0;
            ++f_kx;
            if (f_kx == p.S) { f_kx = 0; ++f_ky; }
            if (f_ky >= p.R) { f_ky = 0; f_kx = 0; }
            set_tap();
        }
        const int so_a = __builtin_amdgcn_readfirstlane(f_chunk * (BKS * 4));
#pragma unroll
        for (int q = 0; q < PA; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_own_a + q * 1024),
                                                     16, (int)aptr[q], so_a, 0, 0);
        dma_stage ^= 1;
        ++f_chunk;
    };
    auto request_b = [&]() {
        if (b_chunk == f_nchunk) {
            b_chunk = 0;
            ++b_kx;
            if (b_kx == p.S) { b_kx = 0; ++b_ky; }
            if (b_ky >= p.R) { b_ky = 0; b_kx = 0; }
            b_k = ((p.wt_ky0 + b_ky * p.wt_step) * p.wt_S + p.wt_kx0 + b_kx * p.wt_step) * p.Cs;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int so_b = __builtin_amdgcn_readfirstlane((int)((pl * p.wplane + b_k) * 2));
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + b_stage * B_STAGE +
                                                                                                        pl * B_PLANE + lds_wave_b + hb * (RPB * 64)),
                                                         16, (int)brow[hb], so_b, 0, 0);
        }
        b_stage ^= 1;
        ++b_chunk;
        b_k += BKS;
    };
    // the first two k-tiles are requested HERE, before the matrix-core side is set up: the ~150 instructions of that set-up (fragment
    // addresses, 32-64 accumulator registers to clear) run under the copies' latency instead of in front of it
    const int nk = kt1 - kt0;
    if constexpr (OWN_A) {
        if (nk > 0) { request_a(); request_b(); }
        if (nk > 1) { request_a(); request_b(); }
    } else {
        if (nk > 0) request_tile();               // tile 0 -> stage 0
        if (nk > 1) request_tile();               // tile 1 -> stage 1
    }
    if constexpr (FRONT) {
        // (host: Ktot == 64, never split - nk == 2, both A stages are filled here, under the latency of the weight copies.)
        // Waves 2 s and 2 s + 1 form k-tile s: 128-row tiles - a thread takes all 32 channels of one of the 128 rows; 64-row tiles - 16
        // channels of one of the 64 rows.  The channels are wave-uniform, so the front conv's weights and bias arrive through scalar
        // loads; exact fp32 FMAs (CIN <= 21 terms).  The row lands in the image the copies would have left: 16-byte slot q of row r
        // at slot q ^ ((r >> 1) & 7).
        const int fwv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int fs = fwv >> 1;
        constexpr int QN = BM == 128 ? 8 : 4;
        const int fr = BM == 128 ? (fwv & 1) * 64 + (tid & 63) : (tid & 63);
        const int q0 = BM == 128 ? 0 : (fwv & 1) * 4;
        const int fm = m0 + fr;
        float xr[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) xr[c] = fm < p.M ? p.fr_x[(size_t)fm * CIN + c] : 0.f;
        float* const arow_l = reinterpret_cast<float*>(smem_raw + fs * A_STAGE + fr * 128);
        const float* const w1 = p.fr_w1 + (size_t)fs * BKS * CIN;
        const float* const b1 = p.fr_b1 ? p.fr_b1 + fs * BKS : nullptr;
        const int sw = (fr >> 1) & 7;
#pragma unroll
        for (int qq = 0; qq < QN; ++qq) {
            const int q = q0 + qq;
            f32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = q * 4 + e;
                float a = b1 ? b1[k] : 0.f;
#pragma unroll
                for (int c = 0; c < CIN; ++c) a = fmaf(xr[c], w1[k * CIN + c], a);
                h[e] = a > 0.f ? a : a * p.fr_slope;
            }
            *reinterpret_cast<f32x4*>(arow_l + ((q ^ sw) << 2)) = h;
            if (p.fr_h1 && tile_n == 0 && fm < p.M) *reinterpret_cast<f32x4*>(p.fr_h1 + (size_t)fm * (2 * BKS) + fs * BKS + q * 4) = h;
        }
        __syncthreads();
    }

    // ---- matrix-core side
    const int lane = tid & 63;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave_id / WN;
    const int wn = wave_id % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;

    // Two accumulator sets where the wave tile leaves the registers (wave tiles up to 32 x 64): the LEADING piece product a0 b0 goes
    // to `acc`, the five small ones (<= 2^-8 of it) to `acc_lo`, summed once after the k-loop.  Every MFMA rounds its accumulator
    // once; with one set a 3x3 256-channel reduction is a chain of 864 roundings at the full magnitude of the running sum - 5.9e-7
    // rms against fp64, 3.8x torch's blocked CPU convolution on the same inputs (tests/aids/local_error.py: the source of the
    // build's 1.2-1.5x forward noise on DeepLab).  Kept apart, only the 144 roundings of the a0 b0 chain are at full magnitude
    // (the others are 2^-8 of it): sqrt(6) = 2.4x less rounding noise for 16 / 32 more registers, no more MFMAs.
    constexpr bool ACC2 = KS_ACC2 && TM * TN <= 2;
    f32x16 acc[TM][TN];
    f32x16 acc_lo[ACC2 ? TM : 1][ACC2 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][j][e] = 0.f;
                if (ACC2) acc_lo[i][j][e] = 0.f;
            }

    // fragment addresses.  MFMA step s (k = 16 s .. 16 s + 15 of the tile): lane half h supplies k = 16 s + 8 h + (0..7):
    //   A (fp32): 16-byte slots 4 s + 2 h and 4 s + 2 h + 1 of the 128-byte row, stored at slot ^ ((row >> 1) & 7);
    //   B (bf16): slot 2 s + h of the 64-byte row, stored at slot ^ ((row >> 2) & 3).
    const int swa = (li >> 1) & 7, swb = (li >> 2) & 3;
    int aoff[2][2], boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        aoff[s][0] = (row_w + li) * 128 + (((4 * s + 2 * lh) ^ swa) << 4);
        aoff[s][1] = (row_w + li) * 128 + (((4 * s + 2 * lh + 1) ^ swa) << 4);
        boff[s] = NSTAGE * A_STAGE + (col_w + li) * 64 + (((2 * s + lh) ^ swb) << 4);
    }

    f32x4 ra[TM][2];            // raw fp32 fragments of the half-tile being split
    bf16x8 pa[2][TM][3];        // split A fragments (two half-tiles in flight)
    bf16x8 fb[2][TN][3];        // weight pieces

    auto read_frags = [&](int stage, int s, int set) {
        const lds_char* a = lds0 + stage * A_STAGE;
        const lds_char* b = lds0 + stage * B_STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra[i][0]) : "v"(a + aoff[s][0]), "n"(i * 32 * 128));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra[i][1]) : "v"(a + aoff[s][1]), "n"(i * 32 * 128));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][j][pl]) : "v"(b + boff[s]), "n"(pl * B_PLANE + j * 32 * 64));
    };
    auto landed = [&](int set) {      // after the lgkmcnt wait: order the consumers of this set behind it
#pragma unroll
        for (int i = 0; i < TM; ++i) { pin(ra[i][0]); pin(ra[i][1]); }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) pin(fb[set][j][pl]);
    };
    auto split_set = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (KS_ABLATE & 2) {
                pa[set][i][0] = __builtin_bit_cast(bf16x8, ra[i][0]); pa[set][i][1] = __builtin_bit_cast(bf16x8, ra[i][1]);
                pa[set][i][2] = __builtin_bit_cast(bf16x8, ra[i][0] + ra[i][1]);
                continue;
            }
            split8(ra[i][0], ra[i][1], pa[set][i][0], pa[set][i][1], pa[set][i][2]);
        }
    };
    // one piece product over the wave's accumulators (consecutive MFMAs write different accumulators)
    auto product = [&](int set, int ap, int bp) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (ACC2 && (ap | bp) != 0)
                    acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][i][ap], fb[set][j][bp], acc_lo[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[set][i][ap], fb[set][j][bp], acc[i][j], 0, 0, 0);
            }
    };
    constexpr int NM = TM * TN;                 // MFMAs per piece product
    constexpr int NV = TM * 44;                 // VALU operations of one split_set
    // products issued BEFORE the wait for the next half-tile's fragments (KS_LATE_WAIT: two where a product is at least two MFMAs -
    // the fragment reads were requested just ahead of this half-tile, one product of 64 cycles does not cover their latency)
    constexpr int NPRE = (KS_LATE_WAIT && NM >= 2) ? 2 : 1;
    constexpr int VPM = (NV + (6 - NPRE) * NM - 1) / ((6 - NPRE) * NM);
    // the six piece products of half-tile `set` (smallest terms first); with `prep`, the fragments of the next half-tile - already
    // requested - are waited for after the first NPRE products and split between the MFMAs of the others
    auto half_tile = [&](int set, bool prep) {
        __builtin_amdgcn_sched_barrier(0);
        product(set, 2, 0);
        if (NPRE == 2) product(set, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (prep) {
            wait_lgkm<0>();
            landed(set ^ 1);
            split_set(set ^ 1);
        }
        if (NPRE == 1) product(set, 1, 1);
        product(set, 0, 2);
        product(set, 1, 0);
        product(set, 0, 1);
        product(set, 0, 0);
        if (prep) {
#pragma unroll
            for (int n = 0; n < (6 - NPRE) * NM; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);      // VPM VALU
            }
            // the pieces are produced HERE, under this half-tile's MFMAs (left alone, the compiler sinks the split to their first use)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pin(pa[set ^ 1][i][pl]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    constexpr bool W128 = TM == 1 && TN == 4;      // wave tile 32 x 128: every A fragment is split ONCE per workgroup, 44 VALU operations per 24 MFMAs
    if constexpr (W128) {
        // The k-loop of the 32 x 128 wave tile (round 6).  With the generic loop's two complete sets of weight fragments (96 registers
        // for four column blocks) the tile spills; here planes 1 and 2 of the weight pieces are SINGLE-buffered and reloaded as soon
        // as their products are issued, plane 0 - the last products of a half-tile - keeps two sets.  The six piece products run in
        // plane order: a0 b2 | a1 b1, a0 b1 | a2 b0, a1 b0, a0 b0.  Issue order of the LDS reads inside a half-tile: A (2) and
        // plane 0 (4) of the NEXT half-tile at its start, plane 2 (4) behind this one's plane-2 product, plane 1 (4) behind its
        // plane-1 products; every wait below counts the reads issued after the ones it needs (LDS returns in order).
        bf16x8 b12[4][2];           // [column block][0: plane 1, 1: plane 2]
        bf16x8 b0[2][4];            // [set][column block]: plane 0
        auto rd_a = [&](int stage, int s) {
            const lds_char* a = lds0 + stage * A_STAGE;
            asm volatile("ds_read_b128 %0, %1" : "=v"(ra[0][0]) : "v"(a + aoff[s][0]));
            asm volatile("ds_read_b128 %0, %1" : "=v"(ra[0][1]) : "v"(a + aoff[s][1]));
        };
        auto rd_b0 = [&](int stage, int s, int set) {
            const lds_char* b = lds0 + stage * B_STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0[set][j]) : "v"(b + boff[s]), "n"(j * 32 * 64));
        };
        auto rd_b1 = [&](int stage, int s) {
            const lds_char* b = lds0 + stage * B_STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b12[j][0]) : "v"(b + boff[s]), "n"(B_PLANE + j * 32 * 64));
        };
        auto rd_b2 = [&](int stage, int s) {
            const lds_char* b = lds0 + stage * B_STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b12[j][1]) : "v"(b + boff[s]), "n"(2 * B_PLANE + j * 32 * 64));
        };
        auto mm = [&](const bf16x8& a, int j, const bf16x8& b) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0][j], 0, 0, 0); };
        // one half-tile: pieces pa[cur], weight planes b12 / b0[cur]; NEXT: fragments of half-tile (stage_n, s_n) are fetched and split
        auto half_w = [&](int cur, auto next_tag, int stage_n, int s_n, bool issue_a = false) {
            constexpr bool NEXT = decltype(next_tag)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (NEXT) { rd_a(stage_n, s_n); rd_b0(stage_n, s_n, cur ^ 1); }
            if (NEXT) wait_lgkm<10>(); else wait_lgkm<4>();            // plane 2 of this half-tile (behind it: plane 1 [, A, plane 0 of the next])
#pragma unroll
            for (int j = 0; j < 4; ++j) pin(b12[j][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][0], j, b12[j][1]);                     // a0 b2
            __builtin_amdgcn_sched_barrier(0);
            if (NEXT) rd_b2(stage_n, s_n);
            if (NEXT) wait_lgkm<10>(); else wait_lgkm<0>();            // plane 1 (behind it: A, plane 0, plane 2 of the next)
#pragma unroll
            for (int j = 0; j < 4; ++j) pin(b12[j][0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][1], j, b12[j][0]);                     // a1 b1
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][0], j, b12[j][0]);                     // a0 b1
            __builtin_amdgcn_sched_barrier(0);
            if (NEXT) {
                rd_b1(stage_n, s_n);
                wait_lgkm<12>();                                       // A of the next half-tile and this one's plane 0 (behind them: planes 0, 2, 1 of the next)
                pin(ra[0][0]); pin(ra[0][1]);
                // OWN_A: this wave has now read BOTH halves of its A rows of the current k-tile - the rows of the tile after next go
                // into the same stage at once (1.5 k-tiles ahead of their first read; no other wave touches these rows)
                if (OWN_A && issue_a) request_a();
                split8(ra[0][0], ra[0][1], pa[cur ^ 1][0][0], pa[cur ^ 1][0][1], pa[cur ^ 1][0][2]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) pin(b0[cur][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][2], j, b0[cur][j]);                    // a2 b0
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][1], j, b0[cur][j]);                    // a1 b0
#pragma unroll
            for (int j = 0; j < 4; ++j) mm(pa[cur][0][0], j, b0[cur][j]);                    // a0 b0
            if (NEXT) {
#pragma unroll
                for (int n = 0; n < 12; ++n) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);        // four VALU (44 in all)
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pin(pa[cur ^ 1][0][pl]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (nk > 0) {
            if (nk > 1) wait_vm<NPIECE>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            rd_a(0, 0); rd_b0(0, 0, 0); rd_b2(0, 0); rd_b1(0, 0);
            wait_lgkm<0>();
            pin(ra[0][0]); pin(ra[0][1]);
            split8(ra[0][0], ra[0][1], pa[0][0][0], pa[0][0][1], pa[0][0][2]);
            int issued = nk > 1 ? 2 : 1;
            for (int kt = 0; kt + 1 < nk; ++kt) {
                const int stage = kt & 1;
                const bool more = issued < nk;                          // tile kt + 2 exists
                half_w(0, std::true_type{}, stage, 1, more);
                wait_lgkm<0>();         // (no stall expected: the last reads were issued in front of twelve MFMAs) every fragment of tile kt is in registers
                // tile kt + 1 has landed (OWN_A: behind it only this wave's A rows of tile kt + 2 may still be in flight)
                if (OWN_A && more) wait_vm<PA>(); else wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                if (more) {                                             // -> the stage of tile kt
                    if constexpr (OWN_A) request_b(); else request_tile();
                    ++issued;
                }
                half_w(1, std::true_type{}, stage ^ 1, 0);
            }
            half_w(0, std::true_type{}, (nk - 1) & 1, 1);
            half_w(1, std::false_type{}, 0, 0);
        }
    } else
    if (nk > 0) {
        if (nk > 1) wait_vm<NPIECE>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        read_frags(0, 0, 0);
        wait_lgkm<0>();
        landed(0);
        split_set(0);
        int issued = nk > 1 ? 2 : 1;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int stage = kt & 1;
            read_frags(stage, 1, 1);
            half_tile(0, true);
            // tile kt + 1 (the only copy in flight) has landed for every wave, and every wave has finished reading tile kt
            if (!(KS_ABLATE & 4)) {
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (issued < nk) { request_tile(); ++issued; }        // -> the stage of tile kt
            read_frags(stage ^ 1, 0, 0);
            half_tile(1, true);
        }
        {
            const int stage = (nk - 1) & 1;
            read_frags(stage, 1, 1);
            half_tile(0, true);
            half_tile(1, false);
        }
    }
    if (ACC2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += acc_lo[i][j];
    }

    const KsParams& ep = p;
    const FastDiv e_gl = {ep.div_gl.mul, ep.div_gl.shift, ep.div_gl.d}, e_hw = {ep.div_hw.mul, ep.div_hw.shift, ep.div_hw.d},
                  e_w = {ep.div_w.mul, ep.div_w.shift, ep.div_w.d};
    if (KS_ABLATE & 8) {        // timing ablation: no epilogue (one store per lane keeps the accumulators alive); WRONG results
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        p.dst[(size_t)(m0 % 64) * p.Ng + (tid & 31)] = t;
        return;
    }
    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // Fused statistics of the normalisation layer that follows (conv_igemm.hip): fp64 column sums of y and y^2 over this tile's
    // rows, four rows at a time in fp32 where the tile lies inside one group and inside the tensor.
    // Tiles inside one group and inside the tensor (nearly all): four consecutive rows are summed in fp32, the 4-row sums in fp64 (as the
    // bf16 kernel does); tiles that straddle a group boundary or the tensor's end take every element to fp64.  Round 3 took EVERY
    // element to fp64 because the shortcut moves a BatchNorm statistic by ~1e-7 and the chained-loss bound then sat at 1e-3; round 4
    // measured those losses as noise of several 1e-3 in every fp32 arithmetic, the reference's included (tests/test_accuracy_gpu.py),
    // and the whole golden suite passes either way: -0.5 ... -0.9 ms per config-2 step (KS_FAST_STATS=0 restores the old epilogue).
#ifndef KS_FAST_STATS
#define KS_FAST_STATS 1
#endif
    // Results leave through LDS (whole tiles; the k-loop's images are dead): in the MFMA layout a lane owns single elements of 16 rows -
    // 32 four-byte stores per lane of a 128x64 tile, each wave-instruction touching 2 x 128 bytes, ~2000 cycles of the store path
    // against 6000 cycles of MFMAs when the reduction is 256 long.  Staged as [row][BN + 4] floats, every thread then writes 16 bytes =
    // four consecutive channels of a row: 4x fewer store instructions, full-width (conv_bf16.hip's epilogue does the same for bf16).
#ifndef KS_STAGE_OUT
#define KS_STAGE_OUT 1
#endif
    constexpr int OLD = BN + 4;
    constexpr int SREC_OFF = BM * OLD * 4;     // the row lanes' statistics of a straddling tile meet behind the staged tile, [NT / BN][BN][4] doubles
    const bool want_bsums = MODE == MODE_DGRAD && ep.bn_sums != nullptr;        // (never with split-K: the host plans these launches unsplit)
    float* const ot = reinterpret_cast<float*>(smem_raw);
    const bool want_stats = MODE == MODE_FWD && ep.stats != nullptr && !partial;      // (a data gradient never takes forward statistics)
    int gb = 0x7fffffff;
    if (want_stats) gb = (fd_div(m0, e_gl) + 1) * ep.stat_L;
    int bg = 0;
    if (want_bsums) { bg = fd_div(m0, e_gl); gb = (bg + 1) * ep.bn_L; }
    const bool slow_stats = want_stats && (!KS_FAST_STATS || m0 + BM > gb || m0 + BM > ep.M);
    const bool fast_stats = want_stats && !slow_stats;
    // Only the heads' class (32 columns) serves Ng % 4 != 0 (ks_choose routes such launches to it) and stores element by element then;
    // everywhere else the tile - a partial one of a split-K tail too - leaves through LDS as 16-byte row segments
    constexpr bool ANY_NG = BN == 32;
    const bool staged = KS_STAGE_OUT && (!ANY_NG || (ep.Ng & 3) == 0);
    // the few tiles that straddle a group boundary or the tensor's end take their statistics element by element in fp64: from the STAGED
    // tile, in a loop over its rows (store phase below) - in registers (64 unrolled fp64 updates per wave tile) only where nothing is staged
    const bool reg_slow = slow_stats && !staged;
    const bool reg_stats = fast_stats || reg_slow;
    if (staged) __syncthreads();           // every wave has read its last fragments
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + col_w + j * 32 + li;
        const bool nok = n < ep.Ng;
        const float bv = (!partial && ep.bias && nok) ? ep.bias[n] : 0.f;
        double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
        if (fast_stats) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float pre = acc[i][j][g4 * 4 + t] + bv;
                        a += pre;
                        b = fmaf(pre, pre, b);
                    }
                    s0 += (double)a; q0 += (double)b;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr (!ANY_NG && KS_STAGE_OUT) {
                    // Every class but the heads' stages the RAW accumulators (round 6): bias and activation wait for the store phase,
                    // where a thread owns four fixed channels of whole rows - the bias is one hoisted vector, the activation sits
                    // behind a uniform branch.  Here they were six VALU operations per element whether or not the launch carries a
                    // bias or an activation (DeepLab's convolutions carry neither): 190-380 of the ~900 fixed VALU operations a
                    // tile costs beside its k-loop, as many as the split of a 256-channel reduction.  Same operations in the same
                    // order per element: bit-identical.
                    ot[(row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * OLD + col_w + j * 32 + li] = acc[i][j][e];
                    continue;
                }
                const int m = m0 + row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float pre = acc[i][j][e] + bv;
                if (reg_slow && m < ep.M && nok) {
                    const double d = (double)pre;
                    if (m < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
                }
                const float v = partial ? pre : (ANY_NG ? sscg_act(pre, ep.act, ep.slope) : ks_act(pre, ep.act, ep.slope));      // (tanh: the heads' class only)
                if (staged) {           // (rows / columns past the tensor are staged too: the store phase drops them)
                    ot[(row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * OLD + col_w + j * 32 + li] = v;
                } else if (m < ep.M && nok) {
                    if (partial) {
                        ep.part[((size_t)split * (ep.M - ep.m_tail0) + (m - ep.m_tail0)) * ep.Ng + n] = v;
                    } else {
                        size_t row = (size_t)m;
                        if (MODE == MODE_DGRAD && ep.o_step != 1) {   // parity class of a strided data gradient: rows interleave into dx
                            const int img = fd_div(m, e_hw);
                            const int rem = m - img * (ep.OH * ep.OW);
                            const int oi = fd_div(rem, e_w);
                            const int oj = rem - oi * ep.OW;
                            row = (size_t)img * ep.o_HW + (size_t)(oi * ep.o_step + ep.o_a) * ep.o_W + oj * ep.o_step + ep.o_b;
                        }
                        ep.dst[row * ep.Ng + n] = v;
                    }
                }
            }
        }
        if (reg_stats) {
            s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);      // the lane halves hold different rows of a column
            s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
            if (fast_stats && staged) {
                // ONE record per tile (round 6; it was one per wave row): the WM wave rows' column sums meet in LDS behind the staged
                // tile and are added in a fixed order after the store phase's barrier - the finalize launch behind EVERY conv -> norm
                // link reads 2-4x fewer records (276 -> 69 per channel on the 8712-row maps, 538 -> 269 on config 3's)
                if (lh == 0) {
                    double* const sr = reinterpret_cast<double*>(smem_raw + SREC_OFF);
                    sr[(wm * BN + col_w + j * 32 + li) * 2] = s0; sr[(wm * BN + col_w + j * 32 + li) * 2 + 1] = q0;
                }
            } else if (lh == 0 && nok) {      // (the heads' class without a staged tile: never planned with statistics - sscg_convs_stats_geometry)
                double* rec = ep.stats + ((size_t)tile_m * 2) * ep.Ng * 2;
                rec[(size_t)n * 2] = s0; rec[(size_t)n * 2 + 1] = q0;
                rec[((size_t)ep.Ng + n) * 2] = s1; rec[((size_t)ep.Ng + n) * 2 + 1] = q1;
            }
        }
    }
    if (staged) {
        __syncthreads();
        if (fast_stats && tid < BN && n0 + tid < ep.Ng) {
            const double* const sr = reinterpret_cast<const double*>(smem_raw + SREC_OFF);
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += sr[(w * BN + tid) * 2]; b += sr[(w * BN + tid) * 2 + 1]; }
            const int nn = n0 + tid;
            double* rec = ep.stats + ((size_t)tile_m * 2) * ep.Ng * 2;
            rec[(size_t)nn * 2] = a; rec[(size_t)nn * 2 + 1] = b;
            rec[((size_t)ep.Ng + nn) * 2] = 0.0; rec[((size_t)ep.Ng + nn) * 2 + 1] = 0.0;
        }
        if (slow_stats) {               // (workgroup-uniform; the statistics launches carry no activation: the staged tile holds y + bias)
            constexpr int NLS = NT / BN;
            const int c = tid % BN, ln = tid / BN;
            double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
            const float sb = (!ANY_NG && ep.bias && n0 + c < ep.Ng) ? ep.bias[n0 + c] : 0.f;       // (the staged tile of these classes is raw)
            for (int r = ln; r < BM; r += NLS) {
                const int m = m0 + r;
                if (m >= ep.M) break;
                const double d = (double)(ot[r * OLD + c] + sb);
                if (m < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
            }
            double* const sr = reinterpret_cast<double*>(smem_raw + SREC_OFF);
            sr[(ln * BN + c) * 4] = s0; sr[(ln * BN + c) * 4 + 1] = q0; sr[(ln * BN + c) * 4 + 2] = s1; sr[(ln * BN + c) * 4 + 3] = q1;
            __syncthreads();
            if (tid < BN && n0 + tid < ep.Ng) {
                double a = 0.0, b = 0.0, cc = 0.0, d = 0.0;
#pragma unroll
                for (int l = 0; l < NLS; ++l) { a += sr[(l * BN + tid) * 4]; b += sr[(l * BN + tid) * 4 + 1]; cc += sr[(l * BN + tid) * 4 + 2]; d += sr[(l * BN + tid) * 4 + 3]; }
                const int nn = n0 + tid;
                double* rec = ep.stats + ((size_t)tile_m * 2) * ep.Ng * 2;       // the tile's one record
                rec[(size_t)nn * 2] = a; rec[(size_t)nn * 2 + 1] = b;
                rec[((size_t)ep.Ng + nn) * 2] = cc; rec[((size_t)ep.Ng + nn) * 2 + 1] = d;
            }
        }
        constexpr int TPR = BN / 4;             // threads per row (four channels = 16 bytes each)
        constexpr int RPP = NT / TPR;           // rows per pass
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
        // Backward sums of the normalisation layer in front (norm.hip col_reduce_kernel<RM_BWD>, mask recomputed from the layer's input
        // nx): taken HERE, where a thread holds four consecutive channels of whole rows - nx arrives as 16-byte row segments like the
        // result leaves - over the thread's <= BM / RPP rows in fp32 (per group: a tile meets at most one group boundary), then in
        // fp64 across the RPP row lanes.  One record per tile and group.
        f32x4 mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f, ga = 1.f, be = 0.f, sl = 0.f, ql = 0.f, sh = 0.f, qh = 0.f;
        if (want_bsums && n < ep.Ng) {
            mu0 = *reinterpret_cast<const f32x4*>(ep.bn_mean + (size_t)bg * ep.Ng + n);
            rs0 = *reinterpret_cast<const f32x4*>(ep.bn_rstd + (size_t)bg * ep.Ng + n);
            if (bg + 1 < ep.bn_G) {
                mu1 = *reinterpret_cast<const f32x4*>(ep.bn_mean + (size_t)(bg + 1) * ep.Ng + n);
                rs1 = *reinterpret_cast<const f32x4*>(ep.bn_rstd + (size_t)(bg + 1) * ep.Ng + n);
            }
            if (ep.bn_gamma) { ga = *reinterpret_cast<const f32x4*>(ep.bn_gamma + n); be = *reinterpret_cast<const f32x4*>(ep.bn_beta + n); }
        }
        auto out_row = [&](int m) -> size_t {
            if (MODE == MODE_FWD || ep.o_step == 1) return (size_t)m;
            const int img = fd_div(m, e_hw);      // parity class of a strided data gradient: rows interleave into dx
            const int rem = m - img * (ep.OH * ep.OW);
            const int oi = fd_div(rem, e_w);
            const int oj = rem - oi * ep.OW;
            return (size_t)img * ep.o_HW + (size_t)(oi * ep.o_step + ep.o_a) * ep.o_W + oj * ep.o_step + ep.o_b;
        };
        const bool joins = MODE == MODE_DGRAD && !partial && (want_bsums || ep.addend != nullptr);
        // the activation's derivative on the masked side, one scalar for the launch (none: 1, ReLU: 0, LeakyReLU: its slope)
        const float neg_scale = ep.bn_act == SSCG_ACT_RELU ? 0.f : (ep.bn_act == SSCG_ACT_LRELU ? ep.bn_slope : 1.f);
        // (a partial tile of a split-K tail goes to its slice of the workspace, rows counted from the tail's first)
        float* const obase = partial ? ep.part + ((long)split * (ep.M - ep.m_tail0) - ep.m_tail0) * (long)ep.Ng : ep.dst;
        if (n < ep.Ng && !joins) {               // (Ng % 4 == 0: a 16-byte piece is inside the row or outside it)
            // bias and activation of the raw staged tile (the heads' class applied them before staging; a partial tile carries neither)
            const bool has_b = !ANY_NG && !partial && ep.bias != nullptr;
            const bool has_a = !ANY_NG && !partial && ep.act != SSCG_ACT_NONE;
            f32x4 b4 = 0.f;
            if (has_b) b4 = *reinterpret_cast<const f32x4*>(ep.bias + n);
#pragma unroll
            for (int ps = 0; ps < BM / RPP; ++ps) {
                const int r = tid / TPR + ps * RPP;
                const int m = m0 + r;
                if (m >= ep.M) break;
                f32x4 v = *reinterpret_cast<const f32x4*>(ot + r * OLD + c4);
                if (has_b) v += b4;
                if (has_a) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ks_act(v[e], ep.act, ep.slope);
                }
                *reinterpret_cast<f32x4*>(obase + out_row(m) * ep.Ng + n) = v;
            }
        }
        if (n < ep.Ng && joins) {
            // the addend, the layer's input and the mask source of FOUR rows are requested before the first is used: taken row by
            // row the store phase waited one memory latency per row (the 128x64 data-gradient class went from 88 to 107 us)
            constexpr int PSN = BM / RPP;
            static_assert(PSN % 4 == 0, "rows per thread");
#pragma unroll 1
            for (int h = 0; h < PSN; h += 4) {          // (rolled: the body - twelve 16-byte loads in flight, four rows of sums - is the launch's largest block of code)
                f32x4 av[4], yv[4], zv[4];
                size_t rows[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + tid / TPR + (h + q) * RPP;
                    av[q] = 0.f; yv[q] = 0.f; zv[q] = 0.f; rows[q] = 0;
                    if (m < ep.M) {
                        rows[q] = out_row(m);
                        if (ep.addend) av[q] = *reinterpret_cast<const f32x4*>(ep.addend + rows[q] * ep.Ng + n);
                        if (want_bsums) {
                            yv[q] = *reinterpret_cast<const f32x4*>(ep.bn_x + (size_t)m * ep.Ng + n);
                            if (ep.bn_z) zv[q] = *reinterpret_cast<const f32x4*>(ep.bn_z + (size_t)m * ep.Ng + n);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = tid / TPR + (h + q) * RPP;
                    const int m = m0 + r;
                    if (m >= ep.M) break;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ot + r * OLD + c4) + av[q];
                    *reinterpret_cast<f32x4*>(ep.dst + rows[q] * ep.Ng + n) = v;
                    if (want_bsums) {
                        const bool lo = m < gb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xh = (yv[q][e] - (lo ? mu0[e] : mu1[e])) * (lo ? rs0[e] : rs1[e]);
                            const float ym = ep.bn_z ? zv[q][e] : xh * ga[e] + be[e];       // (sign of the output = sign of the pre-activation)
                            const float gg = ym > 0.f ? v[e] : v[e] * neg_scale;        // (ReLU: -0 for a negative masked gradient - the sums do not see the sign of a zero)
                            if (lo) { sl[e] += gg; ql[e] = fmaf(gg, xh, ql[e]); } else { sh[e] += gg; qh[e] = fmaf(gg, xh, qh[e]); }
                        }
                    }
                }
            }
        }
        if (want_bsums) {
            __syncthreads();                    // the staged tile is dead: its LDS takes the row lanes' partial sums [RPP][BN][4]
            f32x4* ps4 = reinterpret_cast<f32x4*>(smem_raw);
            if (n < ep.Ng) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ps4[(tid / TPR) * BN + c4 + e] = f32x4{sl[e], ql[e], sh[e], qh[e]};
            }
            __syncthreads();
            if (tid < BN && n0 + tid < ep.Ng) {
                double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll 4
                for (int rl = 0; rl < RPP; ++rl) {
                    const f32x4 t = ps4[rl * BN + tid];
                    a += (double)t[0]; b += (double)t[1]; c += (double)t[2]; d += (double)t[3];
                }
                const int nn = n0 + tid;
                const int k0 = tile_m - (int)(((long)bg * ep.bn_L) / BM);          // chunk of group g = tile row - first tile row of g
                double* r0 = ep.bn_sums + (((size_t)bg * ep.bn_chunks + k0) * ep.Ng + nn) * 2;
                r0[0] = a; r0[1] = b;
                if (m0 + BM > gb && bg + 1 < ep.bn_G) {          // the tile straddles into group g + 1: it is that group's first tile
                    double* r1 = ep.bn_sums + ((size_t)(bg + 1) * ep.bn_chunks * ep.Ng + nn) * 2;
                    r1[0] = c; r1[1] = d;
                }
            }
        }
    }

}

// y[i] = act(sum_s part[s][i] + bias[i % Ng])   (fixed order => deterministic); 4 floats per thread (Ng % 4 == 0)
__global__ __launch_bounds__(256) void ks_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                                         size_t n, int Ng, int splits, int act, float slope, const float* __restrict__ addend) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(part + (size_t)k * n + i);
    const int c = (int)(i % Ng);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sscg_act(s[e] + (bias ? bias[c + e] : 0.f), act, slope);
    if (addend) o += *reinterpret_cast<const f32x4*>(addend + i);
    *reinterpret_cast<f32x4*>(y + i) = o;
}

// the same, element by element (Ng % 4 != 0: heads with 21 / 20 output channels)
__global__ __launch_bounds__(256) void ks_reduce1_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                                          size_t n, int Ng, int splits, int act, float slope) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    y[i] = sscg_act(s + (bias ? bias[(int)(i % Ng)] : 0.f), act, slope);
}


#ifndef KS_STAGE_OUT
#define KS_STAGE_OUT 1
#endif
constexpr bool KS_STAGE_OUT_HOST = KS_STAGE_OUT != 0;      // the addend joins in the staged store phase

// ---- host side: tile classes and the split-K plan of the tail (same policy as conv_igemm.hip)
enum { KS_128x128 = 0, KS_64x64 = 1, KS_128x64 = 2, KS_128x32 = 3, KS_NCFG = 4 };
const int KS_BM[KS_NCFG] = {128, 64, 128, 128};
const int KS_BN[KS_NCFG] = {128, 64, 64, 32};
const int KS_WM[KS_NCFG] = {4, 2, 4, 4};  // wave rows of a tile
// (Built, measured slower and deleted - numbers in profiles/r05_experiments.txt items 3, 12 and r05_tile_classes_after_diet.txt: a 64x128
// tile, the 128x64 tile as two wave groups halving the reduction, the 64x64 tile as two waves.  The 128x128 tile of four 32x128 waves
// lost in round 3 WITH the generic k-loop (two complete sets of weight fragments: spills); with a k-loop of its own - planes 1 / 2 of
// the weight pieces single-buffered - it replaced the four 64x64 waves in round 6: 220.6 against 216.2 TFLOP/s alone on the 65536-row
// 3x3, half the split operations, profiles/r06_experiments.txt item 18.)
// A/B aids for the tile-class policy inside the step (the thresholds below were tuned on kernels timed ALONE; in the step the VALU
// pipe is the contended resource, and classes with fewer split operations per MFMA may win there although they lose alone)
static int ks_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Measured on the step's shapes (tools/convs_bench.py, profiles/r03_convs_tile_classes.txt):
//  * 128x128 (4 waves of 32x128; 64x64 waves until round 6) wins wherever it fills the chip twice over (>= 512 tiles; >= 1024 when the reduction is short):
//    197-200 TF/s on the 65536-row and 17424-row 3x3 convs;
//  * 128x64 (4 waves of 32x64: every A fragment is split by ONE wave) for long reductions on the 8712 / 17424-row maps whose
//    128x128 tiling would leave CUs idle (135-185 TF/s against 95-168);
//  * 64x64 for short reductions (1x1 convs with <= 512 source channels: prologue / epilogue bound) and few output channels.
int ks_choose(long M, int Ng, int Ktot, int tuning) {
    static const int T128_SHORT = ks_env("SSCG_KS_T128_SHORT", 512);    // min 128x128 tiles for that class on a short reduction (1024 until round 5: alone the
                                                                        // 552-tile 1x1 256 -> 1024 is 8 % faster on 64x64 tiles, in the four-lane step the larger tiles win: -0.9 ms)
    static const int K128 = ks_env("SSCG_KS_K128", 512);               // min reduction length for the 128x128 class
    static const int T128_NARROW = ks_env("SSCG_KS_T128_NARROW", 512);  // ... with one column block of 128x128 tiles (Ng <= 128)
    static const int K128_NARROW = ks_env("SSCG_KS_K128_NARROW", 256);  // min reduction length for the 128x128 class where it has >= T128_SHORT / T128_NARROW tiles
    static const int K12864 = ks_env("SSCG_KS_K12864", 1024);          // min reduction length for the 128x64 class
    static const int T12864 = ks_env("SSCG_KS_T12864", 0);             // min 128x64 tiles for that class (0: no condition)
    if (Ng & 3) return KS_128x32;         // the only class that stores element by element (rows of Ng floats are no multiple of 16 bytes)
    const int forced = (tuning & 0xff) - 1;
    if (forced >= 0 && forced < KS_NCFG && Ng >= KS_BN[forced] / 2) return forced;
    const long tm = cdiv(M, 128);
    if (Ng <= 32) return KS_128x32;       // heads: 21 / 20 output channels (the ResNet generators' 7x7 heads, the DeepLab classifiers)
    if (Ng <= 64) return tm >= 256 ? KS_128x64 : KS_64x64;          // (one 128x64 tile per CU and more; re-swept in round 5: 265-tile maps gain 4-10 %)
    const long t128 = tm * cdiv(Ng, 128);
    // (re-swept after the epilogue diet, profiles/r05_tile_classes_after_diet.txt: with a cheap epilogue the 128x128 class pays from 512
    // tiles on for reductions of 256 and more, and from 4096 tiles on - the HBM-bound 524288-row maps of the PixelDiscriminator -
    // whatever the reduction)
    if (Ng <= 128) return ((t128 >= T128_NARROW && Ktot >= K128_NARROW) || t128 >= 4096) ? KS_128x128 : KS_64x64;
    if (t128 >= (Ktot >= K128 ? 512 : T128_SHORT) && Ktot >= K128_NARROW) return KS_128x128;
    if (Ktot >= K12864 && tm * cdiv(Ng, 64) >= T12864) return KS_128x64;
    return KS_64x64;
}
struct KsSplit { int splits, ksplit, full_tiles, m_tail0; };

KsSplit ks_plan_raw(long M, int Ng, int Ktot, int tuning) {
    const int nk = Ktot / BKS;
    const int cfg = ks_choose(M, Ng, Ktot, tuning);
    const int bm = KS_BM[cfg], bn = KS_BN[cfg];
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(Ng, bn);
    const int tiles = tiles_m * tiles_n;
    KsSplit r = {1, nk, tiles, (int)M};
    const int force = (tuning >> 8) & 0xff;         // 1 = never split, n > 1 = every tile cut in n
    if (force == 1) return r;
    if (force > 1) {
        r.ksplit = cdiv(nk, force);
        r.splits = cdiv(nk, r.ksplit);
        r.full_tiles = 0; r.m_tail0 = 0;
        return r;
    }
    if (nk < 8 || tiles > 2300) return r;
    const int q = tiles / 256;
    const int full_m = (q * 256) / tiles_n;
    const int tail = tiles - full_m * tiles_n;
    if (tail <= 0 || tail > 208) return r;
    int s = 256 / tail;
    if (s > 8) s = 8;
    if (s > nk / 4) s = nk / 4;
    if (s < 2) return r;
    r.ksplit = cdiv(nk, s);
    r.splits = cdiv(nk, r.ksplit);
    r.full_tiles = full_m * tiles_n;
    r.m_tail0 = full_m * bm;
    return r;
}

// stat_L > 0: the launch also produces normalisation statistics; the rows of split tiles are summed separately as ONE extra group of
// records, so they must lie in one normalisation group (else the launch is not split).
KsSplit ks_plan(long M, int Ng, int Ktot, int tuning, long stat_L = 0) {
    KsSplit r = ks_plan_raw(M, Ng, Ktot, tuning);
    if (stat_L > 0 && r.splits > 1 && (r.full_tiles == 0 || r.m_tail0 / stat_L != (M - 1) / stat_L)) {
        const int cfg = ks_choose(M, Ng, Ktot, tuning);
        r.splits = 1; r.ksplit = Ktot / BKS;
        r.full_tiles = cdiv(M, KS_BM[cfg]) * cdiv(Ng, KS_BN[cfg]); r.m_tail0 = (int)M;
    }
    return r;
}

size_t ks_split_bytes(const KsSplit& sp, long M, int Ng) {
    return sp.splits > 1 ? (size_t)sp.splits * (M - sp.m_tail0) * Ng * sizeof(float) : 0;
}

template <int MODE, int WM, int WN, int TM, int TN, int CIN = 0>
int launch_ks(const KsParams& p0, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NT = WM * WN * 64;
    KsParams p = p0;
    p.div_hw = make_fastdiv(p.OH * p.OW);
    p.div_w = make_fastdiv(p.OW);
    p.tiles_n = cdiv(p.Ng, BN);
    p.div_tn = make_fastdiv(p.tiles_n);
    p.div_gl = make_fastdiv((MODE == MODE_DGRAD && p.bn_sums != nullptr) ? p.bn_L : (p.stat_L > 0 ? p.stat_L : 1));
    p.tiles = cdiv(p.M, BM) * p.tiles_n;
    const size_t smem = (size_t)2 * (BM * 128 + 3 * BN * 64);
    auto kern = convs_kernel<MODE, WM, WN, TM, TN, CIN>;
    SSCG_ENSURE_SMEM((kern), smem);
    if (p.splits <= 1) { p.full_tiles = p.tiles; p.m_tail0 = p.M; }
    const int grid = p.full_tiles + (p.tiles - p.full_tiles) * p.splits;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, p);
    SSCG_LAUNCH_CHECK();
    if (p.splits > 1) {
        const size_t n = (size_t)(p.M - p.m_tail0) * p.Ng;
        float* yt = p.dst + (size_t)p.m_tail0 * p.Ng;
        if (p.xstats)
            return launch_split_reduce_stats(p.part, p.bias, yt, 0, p.M - p.m_tail0, p.Ng, p.splits, p.act, p.slope, p.xstats, st);
        if (p.Ng & 3)
            hipLaunchKernelGGL(ks_reduce1_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, st, p.part, p.bias, yt, n, p.Ng, p.splits, p.act, p.slope);
        else
            hipLaunchKernelGGL(ks_reduce_kernel, dim3(cdiv((long)(n / 4), 256)), dim3(256), 0, st, p.part, p.bias, yt, n, p.Ng, p.splits, p.act,
                               p.slope, p.addend ? p.addend + (size_t)p.m_tail0 * p.Ng : nullptr);
        SSCG_LAUNCH_CHECK();
    }
    return SSCG_OK;
}

template <int MODE>
int dispatch_ks(const KsParams& p, int tuning, hipStream_t st) {
    switch (ks_choose(p.M, p.Ng, p.Ktot, tuning)) {
        case KS_128x128: return launch_ks<MODE, 4, 1, 1, 4>(p, st);       // 4 waves of 32x128: every A fragment split by ONE wave, 1.83 split operations per MFMA (round 6)
        case KS_64x64: return launch_ks<MODE, 2, 2, 1, 1>(p, st);
        case KS_128x64: return launch_ks<MODE, 4, 1, 1, 2>(p, st);         // 4 waves of 32x64: every A fragment split by one wave only
        case KS_128x32: return launch_ks<MODE, 4, 1, 1, 1>(p, st);        // 4 waves of 32x32: few-channel heads
        default: return SSCG_ERR_BAD_ARG;
    }
}

// the forward with the front conv fused (CIN input channels -> the 64 channels this conv reduces over): 128x128 or 64x64 tiles
template <int CIN>
int dispatch_ks_front(const KsParams& p, int tuning, hipStream_t st) {
    switch (ks_choose(p.M, p.Ng, p.Ktot, tuning)) {
        case KS_128x128: return launch_ks<MODE_FWD, 4, 1, 1, 4, CIN>(p, st);
        case KS_64x64: return launch_ks<MODE_FWD, 2, 2, 1, 1, CIN>(p, st);
        default: return SSCG_ERR_UNSUPPORTED;
    }
}

void ks_dense_taps(KsParams& p) {
    p.pad_x = p.pad; p.wKtot = p.Ktot;
    p.wt_ky0 = 0; p.wt_kx0 = 0; p.wt_step = 1; p.wt_S = p.S;
    p.o_step = 1; p.o_a = 0; p.o_b = 0; p.o_W = 0; p.o_HW = 0;
}

bool ks_dgrad_by_parity(const sscg_conv_desc* d) { return d->stride == 2 && d->dil == 1 && d->pad_mode == 0; }

long ks_plane(const sscg_conv_desc* d) { return d->w_plane > 0 ? (long)d->w_plane : (long)d->K * d->R * d->S * d->C; }

// ---- fp32 -> three bf16 planes
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n, size_t plane) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    if (i + 8 <= n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(x + i + 4);
        bf16x8 h, m, l;
        split8(a, b, h, m, l);
        *reinterpret_cast<bf16x8*>(y + i) = h;
        *reinterpret_cast<bf16x8*>(y + plane + i) = m;
        *reinterpret_cast<bf16x8*>(y + 2 * plane + i) = l;
    } else {
        for (size_t e = i; e < n; ++e) {
            const sscg_bf3 t = sscg_split3(x[e]);
            y[e] = t.h; y[plane + e] = t.m; y[2 * plane + e] = t.l;
        }
    }
}

// [K][RS][C] fp32 -> three planes of [C][RS][K] bf16 (the data-gradient operand of a split weight)
__global__ void krsc_to_crsk_split_kernel(const float* __restrict__ w, bf16* __restrict__ wt, int K, int RS, int C, size_t plane) {
    __shared__ float t[32][33];
    const int rs = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, c = c0 + tx;
        t[r][tx] = (k < K && c < C) ? w[((size_t)k * RS + rs) * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, k = k0 + tx;
        if (k < K && c < C) {
            const sscg_bf3 s = sscg_split3(t[tx][r]);
            const size_t o = ((size_t)c * RS + rs) * K + k;
            wt[o] = s.h; wt[plane + o] = s.m; wt[2 * plane + o] = s.l;
        }
    }
}

}  // namespace

// ---- entry points used by conv_igemm.hip's dispatch
// the copies address both operands through 32-bit buffer offsets, a masked row's offset (2 GB) must lie outside the tensor
static bool ks_extents_ok(const sscg_conv_desc* d, bool dgrad) {
    const size_t src = (dgrad ? (size_t)d->N * d->P * d->Q * d->K : (size_t)d->N * d->H * d->W * d->C) * sizeof(float);
    const size_t wgt = ((size_t)2 * ks_plane(d) + (size_t)d->K * d->R * d->S * d->C) * sizeof(bf16);
    return src < ((size_t)1 << 31) && wgt < ((size_t)1 << 31);
}

bool sscg_convs_fwd_applies(const sscg_conv_desc* d) {
    // (>= 16 output channels: the 128x32 class serves the 21 / 20-channel heads; 1- and 3-channel heads keep conv_igemm.hip's 4-column MFMA)
    static const bool heads = getenv("SSCG_KS_NO_HEADS") == nullptr;       // A/B aid: heads back on the exact kernel
    return d->x_dtype == SSCG_F32 && d->w_dtype == SSCG_BF16X3 && d->y_dtype == SSCG_F32 && d->C % BKS == 0 && d->C <= 4096 &&
           (heads ? d->K >= 16 : (d->K >= 32 && d->K % 4 == 0)) && (d->act != SSCG_ACT_TANH || d->K <= 32) && ks_extents_ok(d, false);
}

bool sscg_convs_dgrad_applies(const sscg_conv_desc* d) {
    return d->y_dtype == SSCG_F32 && d->w_dtype == SSCG_BF16X3 && d->x_dtype == SSCG_F32 && d->K % BKS == 0 && d->K <= 4096 && d->C >= 32 && d->C % 4 == 0 &&
           d->pad_mode == 0 && d->stride <= 2 && ks_extents_ok(d, true);
}

bool sscg_convs_stats_geometry(const sscg_conv_desc* d, long L, int* bm, int* wm, int* tiles_n, int* splits, int* full_tiles, int* m_tail0) {
    const long M = (long)d->N * d->P * d->Q;
    const int cfg = ks_choose(M, d->K, d->R * d->S * d->C, d->tuning);
    if (L < KS_BM[cfg] || (d->K & 3)) return false;       // (the epilogue's statistics ride on the staged tile: Ng % 4 == 0)
    *bm = KS_BM[cfg];
    *wm = 1;                                // one record per tile (round 6: the wave rows meet in LDS)
    *tiles_n = cdiv(d->K, KS_BN[cfg]);
    KsSplit sp = ks_plan(M, d->K, d->R * d->S * d->C, d->tuning, L);
    *splits = sp.splits; *full_tiles = sp.full_tiles; *m_tail0 = sp.m_tail0;
    return true;
}

size_t sscg_convs_fwd_workspace(const sscg_conv_desc* d, long stat_L) {
    const long M = (long)d->N * d->P * d->Q;
    return ks_split_bytes(ks_plan(M, d->K, d->R * d->S * d->C, d->tuning, stat_L), M, d->K);
}

size_t sscg_convs_dgrad_workspace(const sscg_conv_desc* d) {
    if (ks_dgrad_by_parity(d)) return 0;
    const long M = (long)d->N * d->H * d->W;
    return ks_split_bytes(ks_plan(M, d->C, d->R * d->S * d->K, d->tuning), M, d->C);
}

int sscg_convs_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* stats, long stat_L,
                   double* xstats, void* ws, size_t ws_bytes, hipStream_t st) {
    KsParams p = {};
    p.src = reinterpret_cast<const float*>(x); p.wgt = reinterpret_cast<const bf16*>(w); p.wplane = ks_plane(d);
    p.bias = bias; p.dst = reinterpret_cast<float*>(y);
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Cs = d->C; p.Ktot = d->R * d->S * d->C;
    p.SH = d->H; p.SW = d->W; p.OH = d->P; p.OW = d->Q;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = d->pad_mode; p.act = d->act; p.slope = d->slope;
    p.stats = stats; p.stat_L = (int)stat_L; p.xstats = xstats;
    p.src_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->C * sizeof(float));
    p.wgt_bytes = (unsigned)(((size_t)2 * p.wplane + (size_t)d->K * d->R * d->S * d->C) * sizeof(bf16));
    ks_dense_taps(p);
    KsSplit sp = ks_plan(p.M, p.Ng, p.Ktot, d->tuning, stats ? stat_L : 0);
    if (sp.splits > 1 && (!ws || ws_bytes < ks_split_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch_ks<MODE_FWD>(p, d->tuning, st);
}

// PixelDiscriminator's front half as ONE launch (arch/discriminators.py:70-73: Conv2d(cin, 64, 1x1) -> LeakyReLU -> Conv2d(64, 2 ndf, 1x1)
// [-> the norm layer's statistics]): `d` describes the SECOND conv (C = 64 source channels, 1x1, stride 1, no padding), `xf` is the
// FIRST conv's input [N*H*W][cin] fp32, (w1 [64][cin], b1 [64] or null, slope1) its parameters.  The 64-channel map never exists in
// memory: every workgroup forms its rows of it in LDS (exact fp32 FMAs) in front of the split contraction.
bool sscg_convs_front_applies(const sscg_conv_desc* d, int cin) {
    if (!(cin == 3 || cin == 4 || cin == 20 || cin == 21)) return false;
    if (!sscg_convs_fwd_applies(d) || d->C != 2 * BKS || d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->pad_mode != 0 ||
        d->K % 4 != 0 || d->act == SSCG_ACT_TANH)
        return false;
    const long M = (long)d->N * d->P * d->Q;
    const int cfg = ks_choose(M, d->K, d->C, d->tuning);
    return (cfg == KS_128x128 || cfg == KS_64x64) && ks_plan(M, d->K, d->C, d->tuning, 0).splits == 1;
}

int sscg_convs_fwd_front(const sscg_conv_desc* d, const void* xf, int cin, const float* w1, const float* b1, float slope1, void* h1,
                         const void* w, const float* bias, void* y, double* stats, long stat_L, hipStream_t st) {
    if (!sscg_convs_front_applies(d, cin)) return SSCG_ERR_UNSUPPORTED;
    KsParams p = {};
    p.fr_x = reinterpret_cast<const float*>(xf); p.fr_w1 = w1; p.fr_b1 = b1; p.fr_slope = slope1; p.fr_h1 = reinterpret_cast<float*>(h1);
    p.src = nullptr; p.wgt = reinterpret_cast<const bf16*>(w); p.wplane = ks_plane(d);
    p.bias = bias; p.dst = reinterpret_cast<float*>(y);
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Cs = d->C; p.Ktot = d->C;
    p.SH = d->H; p.SW = d->W; p.OH = d->P; p.OW = d->Q;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.dil = 1;
    p.pad_mode = 0; p.act = d->act; p.slope = d->slope;
    p.stats = stats; p.stat_L = (int)stat_L; p.xstats = nullptr;
    p.src_bytes = 0;
    p.wgt_bytes = (unsigned)(((size_t)2 * p.wplane + (size_t)d->K * d->C) * sizeof(bf16));
    ks_dense_taps(p);
    p.splits = 1; p.ksplit = p.Ktot / BKS; p.full_tiles = 0; p.m_tail0 = p.M;      // (launch_ks sets full_tiles / m_tail0 of an unsplit launch)
    p.part = nullptr;
    switch (cin) {
        case 3: return dispatch_ks_front<3>(p, d->tuning, st);
        case 4: return dispatch_ks_front<4>(p, d->tuning, st);
        case 20: return dispatch_ks_front<20>(p, d->tuning, st);
        case 21: return dispatch_ks_front<21>(p, d->tuning, st);
        default: return SSCG_ERR_UNSUPPORTED;
    }
}

// Backward sums of the normalisation layer in front, from this data gradient's epilogue: plain stride-1 / dilated data gradients of the
// split family, groups at least one tile tall (a tile then meets at most one group boundary).  The launch is planned WITHOUT the
// tail split-K (its partial tiles would need the sums in the reduction as well).
bool sscg_convs_bsums_geometry(const sscg_conv_desc* d, int G, long L, int* bm, int* wm, int* chunks) {
    if (!sscg_convs_dgrad_applies(d) || ks_dgrad_by_parity(d) || d->stride != 1) return false;
    const long M = (long)d->N * d->H * d->W;
    if (G <= 0 || L <= 0 || (long)G * L != M) return false;
    const int cfg = ks_choose(M, d->C, d->R * d->S * d->K, d->tuning);
    if (L < KS_BM[cfg]) return false;
    *bm = KS_BM[cfg];
    *wm = 1;                                // one record per tile and group (the sums are taken in the store phase, per workgroup)
    *chunks = (int)(cdiv(L, (long)KS_BM[cfg]) + 1);
    return true;
}

int sscg_convs_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias, void* dx, int act, float slope,
                     void* ws, size_t ws_bytes, hipStream_t st, const sscg_bsums* bs, const void* addend) {
    KsParams p = {};
    if (addend) {
        if (bias || act != SSCG_ACT_NONE || !KS_STAGE_OUT_HOST) return SSCG_ERR_UNSUPPORTED;
        p.addend = reinterpret_cast<const float*>(addend);
    }
    if (bs) {
        int bm, wm, chunks;
        if (bias || act != SSCG_ACT_NONE || !sscg_convs_bsums_geometry(d, bs->G, bs->L, &bm, &wm, &chunks)) return SSCG_ERR_UNSUPPORTED;
        p.bn_x = reinterpret_cast<const float*>(bs->nx); p.bn_mean = bs->mean; p.bn_rstd = bs->rstd; p.bn_gamma = bs->gamma; p.bn_beta = bs->beta;
        p.bn_sums = reinterpret_cast<double*>(bs->sums); p.bn_L = (int)bs->L; p.bn_G = bs->G; p.bn_chunks = chunks;
        p.bn_act = bs->act; p.bn_slope = bs->slope;
        p.bn_z = reinterpret_cast<const float*>(bs->nz);
    }
    p.src = reinterpret_cast<const float*>(dy); p.wgt = reinterpret_cast<const bf16*>(wt); p.wplane = ks_plane(d);
    p.bias = bias; p.dst = reinterpret_cast<float*>(dx);
    p.M = d->N * d->H * d->W; p.Ng = d->C; p.Cs = d->K; p.Ktot = d->R * d->S * d->K;
    p.SH = d->P; p.SW = d->Q; p.OH = d->H; p.OW = d->W;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = 0; p.act = act; p.slope = slope;
    p.src_bytes = (unsigned)((size_t)d->N * d->P * d->Q * d->K * sizeof(float));
    p.wgt_bytes = (unsigned)(((size_t)2 * p.wplane + (size_t)d->K * d->R * d->S * d->C) * sizeof(bf16));
    ks_dense_taps(p);
    if (ks_dgrad_by_parity(d)) {
        // stride 2: four parity classes, each a stride-1 data gradient over its sub-lattice of taps (conv_igemm.hip)
        p.splits = 1; p.ksplit = 0; p.part = nullptr;
        p.stride = 1; p.wt_step = 2; p.wt_S = d->S;
        p.o_step = 2; p.o_W = d->W; p.o_HW = d->H * d->W;
        for (int a = 0; a < 2; ++a) {
            for (int b = 0; b < 2; ++b) {
                const int Ha = (d->H - a + 1) / 2, Wb = (d->W - b + 1) / 2;
                if (Ha <= 0 || Wb <= 0) continue;
                const int ky0 = (a + d->pad) & 1, kx0 = (b + d->pad) & 1;
                KsParams q = p;
                q.R = ky0 < d->R ? (d->R - ky0 + 1) / 2 : 0;
                q.S = kx0 < d->S ? (d->S - kx0 + 1) / 2 : 0;
                if (q.R == 0 || q.S == 0) { q.R = 0; q.S = 0; }
                q.pad = (a + d->pad - ky0) / 2;
                q.pad_x = (b + d->pad - kx0) / 2;
                q.wt_ky0 = ky0; q.wt_kx0 = kx0;
                q.o_a = a; q.o_b = b;
                q.OH = Ha; q.OW = Wb;
                q.M = d->N * Ha * Wb;
                q.Ktot = q.R * q.S * q.Cs;
                int rc = dispatch_ks<MODE_DGRAD>(q, d->tuning & ~0xff00, st);
                if (rc) return rc;
            }
        }
        return SSCG_OK;
    }
    // Launches with fused sums are never split: the idle second round of a 276-tile launch costs ~16 us ALONE, but in the step the other
    // lanes' kernels fill it - a tail split whose reduction took the sums for its rows (built in round 6, bit-reproducible, 3x3 256 d2
    // 148 -> 121 us alone) left the step where it was (127.9 / 128.1 against 127.4 / 127.4 ms; profiles/r06_experiments.txt item 2).
    KsSplit sp = ks_plan(p.M, p.Ng, p.Ktot, bs ? ((d->tuning & 0xff) | 0x100) : d->tuning);
    if (sp.splits > 1 && (!ws || ws_bytes < ks_split_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch_ks<MODE_DGRAD>(p, d->tuning, st);
}

extern "C" int sscg_conv2d_split_applies(const sscg_conv_desc* d, int kind) {
    if (!d) return 0;
    sscg_conv_desc t = *d;
    t.x_dtype = SSCG_F32; t.y_dtype = SSCG_F32; t.w_dtype = SSCG_BF16X3;
    if (kind == 0) return sscg_convs_fwd_applies(&t) ? 1 : 0;
    if (kind == 1) return sscg_convs_dgrad_applies(&t) ? 1 : 0;
    return 0;
}

extern "C" int sscg_split3(const float* src, void* dst, int64_t n, int64_t plane_stride, void* stream) {
    if (!src || !dst || n <= 0 || plane_stride < n) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<bf16*>(dst),
                       (size_t)n, (size_t)plane_stride);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

int sscg_krsc_to_crsk_split(const float* w, void* wt, int K, int RS, int C, hipStream_t st) {
    dim3 grid(cdiv(C, 32), cdiv(K, 32), RS);
    hipLaunchKernelGGL(krsc_to_crsk_split_kernel, grid, dim3(256), 0, st, w, reinterpret_cast<bf16*>(wt), K, RS, C, (size_t)K * RS * C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

// =====================================================================================================================
// Weight gradient by the split contraction:  dW[k][tap][c] = sum_p dy[p][k] * x[src(p, tap)][c]   (fp32 tensors, fp32 result)
//
// Both operands are activations.  Split between LDS and the matrix cores (conv_wgrad.hip, BF16 == 2) every element is split once per
// tile that touches it - 36 tiles for a 256-channel 3x3 - and the kernel is VALU-bound at 7-15 VALU operations per MFMA.  Here the
// two tensors are split ONCE, by an element-wise pass into scratch planes (sscg_split3's rounding: 10 bytes of HBM traffic per
// element, a tenth of the kernel's time), and the contraction reads bf16 planes only: the [pixel][channel] rows go to LDS as they
// are, by LDS-DMA, and gfx950's transposing LDS read (`ds_read_b64_tr_b16`, conv_bf16.hip wgrad16t_kernel) hands every lane the
// 8 pixels of its channel.  No VALU work in the k-loop.
//   tile 128 (output channels) x 128 (tap, source channel); 4 waves of 64 x 64; k-tile = 16 pixels = ONE MFMA k-step of six piece
//   products per accumulator; three planes per operand and stage = 24 KB: three workgroups per CU.
namespace {

constexpr int WS_BKP = 16;                 // pixels per k-tile
constexpr int WS_ROWB = 256;               // bytes of one pixel row of one plane tile (128 channels bf16)
constexpr int WS_PLANE = WS_BKP * WS_ROWB; // 4 KB
constexpr int WS_OP = 3 * WS_PLANE;        // one operand, one stage
constexpr int WS_STAGE = 2 * WS_OP;

struct WgsParams {
    const bf16* __restrict__ x3;      // three planes of x  [N*H*W][C], xplane elements apart
    const bf16* __restrict__ dy3;     // three planes of dy [N*P*Q][Kc], yplane elements apart
    long xplane, yplane;
    float* __restrict__ out;          // dw (splits == 1) or workspace [splits][Kc][Ng]
    int Kc, Ng, C;
    int H, W, P, Q, S;
    int stride, pad, dil, pad_mode;
    int npix, chunk;
    int tiles_n, tiles, splits;
    float beta;
    FastDiv div_pq, div_q;
};

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4s;
__device__ __forceinline__ int ws_sw(int pixel) { return (pixel & 3) << 2; }      // chunk swizzle of a 256-byte pixel row (wgrad16t_kernel)

// The 128 x 128 result tile of the weight-gradient kernels (4 waves of 64 x 64) leaves through LDS: in the MFMA layout the tile costs
// 64 four-byte stores per lane (256 wave-instructions of 2 x 128 bytes: ~4000 cycles of the store path against 13000 cycles of MFMAs
// in a 17-k-tile workgroup of the 1x1 kernel); staged as [row][132] floats (the k-loop's LDS is dead) every thread writes 16 bytes.
// Ng % 4 == 0 (the callers require C % 8 == 0).  beta != 0: out = tile + beta * out (single-split launches accumulate in place).
__device__ __forceinline__ void ws_store_tile(const f32x16 (&acc)[2][2], float* ot, float* __restrict__ out, int m0, int n0, int Kc, int Ng,
                                              int row_w, int col_w, int li, int lh, int tid, float beta) {
    constexpr int OLD = 132;
    __syncthreads();                        // every wave has read its last fragments
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ot[(row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * OLD + col_w + j * 32 + li] = acc[i][j][e];
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    const int n = n0 + c4;
    if (n >= Ng) return;
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) {
        const int r = (tid >> 5) + ps * 8;
        const int m = m0 + r;
        if (m >= Kc) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(ot + r * OLD + c4);
        float* o = out + (size_t)m * Ng + n;
        if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
}

template <int NSTAGE>
__global__ __launch_bounds__(256, 2) void wgrads_kernel(WgsParams p) {
    constexpr int TM = 2, TN = 2, WN = 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [NSTAGE][A planes 0..2][B planes 0..2]
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;

    const int tid = threadIdx.x;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / p.tiles;
    const int tl = lin - split * p.tiles;
    const int tile_n = tl % p.tiles_n;
    const int tile_m = tl / p.tiles_n;
    const int m0 = tile_m * 128;
    const int n0 = tile_n * 128;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.npix, p_begin + p.chunk);
    const bool reflect = p.pad_mode == 1;
    const int wave_id = tid >> 6;
    const int lane = tid & 63;
    const bf16* const zero = reinterpret_cast<const bf16*>(sscg_zero_page_s);

    // ---- copy side: wave w owns pixel rows 4 w .. 4 w + 3 of every k-tile, in all three planes of both operands (one 1 KB piece
    // each); lane l writes chunk l % 16 of row l / 16, i.e. FETCHES source chunk (l % 16) ^ ws_sw(row)
    const int c_row = wave_id * 4 + (lane >> 4);
    const int c_q = (lane & 15) ^ ws_sw(c_row);
    const bool a_colok = m0 + c_q * 8 < p.Kc;
    const bf16* const a_src = p.dy3 + (a_colok ? m0 + c_q * 8 : 0);
    const int b_n = n0 + c_q * 8;
    const bool b_colok = b_n < p.Ng;
    int tdy, tdx;
    const bf16* b_src;
    {
        const int nn = b_colok ? b_n : 0;
        const int tap = nn / p.C;
        const int c = nn - tap * p.C;
        const int ky = tap / p.S;
        const int kx = tap - ky * p.S;
        tdy = ky * p.dil - p.pad;
        tdx = kx * p.dil - p.pad;
        b_src = p.x3 + c;
    }
    const bool plain = p.Ng == p.C && p.stride == 1 && p.pad == 0;      // 1x1, stride 1: source pixel = output pixel
    int dma_stage = 0;
    int f_pix = p_begin;
    const int lds_wave = __builtin_amdgcn_readfirstlane(wave_id * 1024);
    auto request_tile = [&]() {
        const int pix = f_pix + c_row;
        const bool pok = pix < p_end;
        const bf16* ga = (a_colok && pok) ? a_src + (size_t)pix * p.Kc : zero;
        const long ya = (a_colok && pok) ? p.yplane : 0;
        bool ok = b_colok && pok;
        const int pp = ok ? pix : 0;
        size_t spix;
        if (plain) {
            spix = (size_t)pp;
        } else {
            const int img = fd_div(pp, p.div_pq);
            const int rem = pp - img * (p.P * p.Q);
            const int oy = fd_div(rem, p.div_q);
            const int ox = rem - oy * p.Q;
            int sy = oy * p.stride + tdy;
            int sx = ox * p.stride + tdx;
            if (reflect) {
                sy = sy < 0 ? -sy : sy;
                sx = sx < 0 ? -sx : sx;
                sy = sy >= p.H ? 2 * (p.H - 1) - sy : sy;
                sx = sx >= p.W ? 2 * (p.W - 1) - sx : sx;
            }
            ok = ok & ((unsigned)sy < (unsigned)p.H) & ((unsigned)sx < (unsigned)p.W);
            spix = (size_t)((img * p.H + sy) * p.W + sx);
        }
        const bf16* gb = ok ? b_src + spix * p.C : zero;
        const long xb = ok ? p.xplane : 0;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + pl * ya),
                                             (__attribute__((address_space(3))) void*)(lds0 + dma_stage * WS_STAGE + pl * WS_PLANE + lds_wave), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + pl * xb),
                                             (__attribute__((address_space(3))) void*)(lds0 + dma_stage * WS_STAGE + WS_OP + pl * WS_PLANE + lds_wave), 16, 0, 0);
        }
        dma_stage = dma_stage + 1 == NSTAGE ? 0 : dma_stage + 1;
        f_pix += WS_BKP;
    };
    constexpr int NPIECE = 6;

    // ---- matrix-core side
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int i16 = lane & 15;
    const int wm = wave_id / WN;
    const int wn = wave_id % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;
    // transposing read: the lane supplies 4 consecutive channels (8 bytes) of pixel 8 lh + (i16 >> 2) and receives 4 pixels of ITS channel
    const int tr_pix = 8 * lh + (i16 >> 2);
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = row_w + i * 32 + (li & 16) + 4 * (i16 & 3);
        a_off[i] = tr_pix * WS_ROWB + (((ch >> 3) ^ ws_sw(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = col_w + j * 32 + (li & 16) + 4 * (i16 & 3);
        b_off[j] = WS_OP + tr_pix * WS_ROWB + (((ch >> 3) ^ ws_sw(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nsteps = (p_end - p_begin + WS_BKP - 1) / WS_BKP;
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (s < nsteps) { request_tile(); ++issued; }
    }
    if (nsteps >= NSTAGE - 1) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    int rd_stage = 0;
    for (int it = 0; it < nsteps; ++it) {
        const lds_char* base = lds0 + rd_stage * WS_STAGE;
        bf16x4s fa[TM][3][2], fb[TN][3][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][pl][0]) : "v"(base + a_off[i]), "n"(pl * WS_PLANE));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][pl][1]) : "v"(base + a_off[i]), "n"(pl * WS_PLANE + 4 * WS_ROWB));
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][pl][0]) : "v"(base + b_off[j]), "n"(pl * WS_PLANE));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][pl][1]) : "v"(base + b_off[j]), "n"(pl * WS_PLANE + 4 * WS_ROWB));
            }
        const bool more = issued < nsteps;
        if (more) { request_tile(); ++issued; }      // into the stage read in the previous iteration (every wave is past that barrier)
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<0>();
        bf16x8 va[TM][3], vb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                va[i][pl] = __builtin_shufflevector(fa[i][pl][0], fa[i][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(va[i][pl]);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                vb[j][pl] = __builtin_shufflevector(fb[j][pl][0], fb[j][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(vb[j][pl]);
            }
        // six piece products, smallest terms first; consecutive MFMAs write different accumulators
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int ap = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0;
            const int bp = (t == 0 || t == 3 || t == 5) ? 0 : (t == 1 || t == 4) ? 1 : 2;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[i][ap], vb[j][bp], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        rd_stage = rd_stage + 1 == NSTAGE ? 0 : rd_stage + 1;
        if (more) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }

    float* out = p.out + (size_t)split * p.Kc * p.Ng;
    const bool direct = (p.splits == 1);
    ws_store_tile(acc, reinterpret_cast<float*>(smem_raw), out, m0, n0, p.Kc, p.Ng, row_w, col_w, li, lh, tid, direct ? p.beta : 0.f);
}

// =====================================================================================================================
// 1x1 (stride 1, no padding) weight gradient with the operand split FUSED into the kernel:  dW[k][c] = sum_p dy[p][k] * x[p][c].
//
// A 1x1 filter meets every element of x and dy in one tap only, so the element-wise split pass of `wgrads_kernel` (10 bytes of HBM
// traffic per element) costs what it saves, and conv_wgrad.hip's on-the-fly kernel splits every fragment in the wave that feeds it -
// every element twice per tile (both waves of a tile row / column), 13 VALU operations per MFMA measured, 75-100 TFLOP/s.  Here the
// fp32 [pixel][channel] rows of a k-tile (16 pixels x 128 channels per operand) go to LDS by LDS-DMA, every thread splits ONE
// 8-channel group of each operand (88 VALU operations per k-tile: 3.7 per MFMA) and writes the three bf16 pieces into the plane
// images `wgrads_kernel` reads - same row swizzle, same `ds_read_b64_tr_b16` fragment reads, same MFMA order.  The split of tile
// t + 1 runs under the MFMAs of tile t (one plane buffer, two barriers per k-tile: readers of the planes done / pieces written);
// three raw stages keep two k-tiles of copies in flight.  72 KB of LDS: two workgroups per CU.
constexpr int WF_ROWB = 512;                       // bytes of one pixel row of a raw tile (128 channels fp32)
constexpr int WF_RAW_OP = WS_BKP * WF_ROWB;        // 8 KB
constexpr int WF_RAW_STAGE = 2 * WF_RAW_OP;        // dy rows, then x rows
constexpr int WF_NRAW = 3;
constexpr int WF_PLANES = WF_NRAW * WF_RAW_STAGE;  // offset of the plane buffer (A planes 0..2, B planes 0..2 = WS_STAGE bytes)
constexpr int WF_SMEM = WF_PLANES + WS_STAGE;      // 72 KB

struct WgfParams {
    const float* __restrict__ x;      // [npix][C]
    const float* __restrict__ dy;     // [npix][Kc]
    float* __restrict__ out;          // dw (splits == 1) or workspace [splits][Kc][C]
    int Kc, C;
    int npix, chunk;
    int tiles_n, tiles, splits;
    float beta;
};

__global__ __launch_bounds__(256, 2) void wgradf_kernel(WgfParams p) {
    constexpr int TM = 2, TN = 2, WN = 2;
    constexpr int NP = 4;                      // copy instructions per wave and k-tile (two per operand)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;

    const int tid = threadIdx.x;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / p.tiles;
    const int tl = lin - split * p.tiles;
    const int tile_n = tl % p.tiles_n;
    const int tile_m = tl / p.tiles_n;
    const int m0 = tile_m * 128;
    const int n0 = tile_n * 128;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.npix, p_begin + p.chunk);
    const int wave_id = tid >> 6;
    const int lane = tid & 63;

    // ---- copy side: a wave instruction moves two pixel rows (1 KB); wave w owns rows 4 w .. 4 w + 3 of both operands.  LDS unit
    // u = lane & 31 of a row holds the 16-byte source unit 2 (u & 15) + (u >> 4): the two halves of an 8-channel group sit 256 bytes
    // apart, so that the split stage's two ds_read_b128 are each contiguous over 16 lanes
    const int c_u = lane & 31;
    const int c_g = 2 * (c_u & 15) + (c_u >> 4);              // source 16-byte unit (4 channels)
    const bool a_colok = m0 + 4 * c_g < p.Kc;
    const bool b_colok = n0 + 4 * c_g < p.C;
    // buffer loads (KS_BUFLD): the lane's offset inside a k-tile in one VGPR per copy, the k-tile's first pixel in an SGPR offset; a
    // masked lane (column past the tensor, pixel past this split) asks for offset 2 GB = out of range = zeros
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, (short)0, (int)((unsigned)p.npix * (unsigned)p.Kc * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((unsigned)p.npix * (unsigned)p.C * 4u), 0x00020000);
    unsigned a_vo[2], b_vo[2];
    int c_row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        c_row[j] = wave_id * 4 + 2 * j + (lane >> 5);
        a_vo[j] = a_colok ? ((unsigned)c_row[j] * (unsigned)p.Kc + m0 + 4 * c_g) * 4u : 0x80000000u;
        b_vo[j] = b_colok ? ((unsigned)c_row[j] * (unsigned)p.C + n0 + 4 * c_g) * 4u : 0x80000000u;
    }
    int f_pix = p_begin;
    int dma_stage = 0;
    const int lds_wave = __builtin_amdgcn_readfirstlane(wave_id * 2048);      // rows 4 w .. 4 w + 3
    auto request_tile = [&]() {
        const int so_a = __builtin_amdgcn_readfirstlane(f_pix * p.Kc * 4), so_b = __builtin_amdgcn_readfirstlane(f_pix * p.C * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool pok = f_pix + c_row[j] < p_end;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * WF_RAW_STAGE + lds_wave + j * 1024),
                                                     16, (int)(pok ? a_vo[j] : 0x80000000u), so_a, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * WF_RAW_STAGE + WF_RAW_OP + lds_wave + j * 1024),
                                                     16, (int)(pok ? b_vo[j] : 0x80000000u), so_b, 0, 0);
        }
        dma_stage = dma_stage + 1 == WF_NRAW ? 0 : dma_stage + 1;
        f_pix += WS_BKP;
    };

    // ---- split side: thread t owns channels 8 q .. 8 q + 7 of pixel row r of both operands
    const int s_r = tid >> 4, s_q = tid & 15;
    const int s_raw = s_r * WF_ROWB + s_q * 16;                                   // + 256: the upper four channels
    const int s_pl = WF_PLANES + s_r * WS_ROWB + ((s_q ^ ws_sw(s_r)) << 4);       // + pl * WS_PLANE (+ WS_OP for x)
    f32x4 raw[2][2];
    bf16x8 pc[2][3];
    auto read_raw = [&](int stage) {
        const lds_char* b = lds0 + stage * WF_RAW_STAGE + s_raw;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(raw[op][0]) : "v"(b), "n"(op * WF_RAW_OP));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(raw[op][1]) : "v"(b), "n"(op * WF_RAW_OP + 256));
        }
    };
    auto split_raw = [&]() {
#pragma unroll
        for (int op = 0; op < 2; ++op) split8(raw[op][0], raw[op][1], pc[op][0], pc[op][1], pc[op][2]);
    };
    auto write_pieces = [&]() {
        lds_char* b = lds0 + s_pl;
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(b), "v"(pc[op][pl]), "n"(op * WS_OP + pl * WS_PLANE) : "memory");
    };

    // ---- matrix-core side (wgrads_kernel's fragment geometry)
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int i16 = lane & 15;
    const int wm = wave_id / WN;
    const int wn = wave_id % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;
    const int tr_pix = 8 * lh + (i16 >> 2);
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = row_w + i * 32 + (li & 16) + 4 * (i16 & 3);
        a_off[i] = WF_PLANES + tr_pix * WS_ROWB + (((ch >> 3) ^ ws_sw(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = col_w + j * 32 + (li & 16) + 4 * (i16 & 3);
        b_off[j] = WF_PLANES + WS_OP + tr_pix * WS_ROWB + (((ch >> 3) ^ ws_sw(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nsteps = (p_end - p_begin + WS_BKP - 1) / WS_BKP;
    int issued = 0;
    // the copies of tile t have landed (for this wave) once at most `later` tiles requested after it are outstanding
    auto wait_tile = [&](int t) {
        const int later = issued - 1 - t;
        if (later >= 2) wait_vm<2 * NP>();
        else if (later == 1) wait_vm<NP>();
        else wait_vm<0>();
    };
    if (nsteps > 0) {
        for (int s = 0; s < WF_NRAW && s < nsteps; ++s) { request_tile(); ++issued; }
        wait_tile(0);
        __builtin_amdgcn_s_barrier();
        read_raw(0);
        wait_lgkm<0>();
        pin(raw[0][0]); pin(raw[0][1]); pin(raw[1][0]); pin(raw[1][1]);
        split_raw();
        write_pieces();
        wait_lgkm<0>();
        if (nsteps > 1) wait_tile(1);
        __builtin_amdgcn_s_barrier();                 // planes(0) written, raw tile 1 landed for every wave, raw stage 0 read by all
        if (issued < nsteps) { request_tile(); ++issued; }      // tile 3 -> stage 0
    }
    for (int it = 0; it < nsteps; ++it) {
        const bool next = it + 1 < nsteps;
        bf16x4s fa[TM][3][2], fb[TN][3][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][pl][0]) : "v"(lds0 + a_off[i]), "n"(pl * WS_PLANE));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][pl][1]) : "v"(lds0 + a_off[i]), "n"(pl * WS_PLANE + 4 * WS_ROWB));
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][pl][0]) : "v"(lds0 + b_off[j]), "n"(pl * WS_PLANE));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][pl][1]) : "v"(lds0 + b_off[j]), "n"(pl * WS_PLANE + 4 * WS_ROWB));
            }
        if (next) read_raw((it + 1) % WF_NRAW);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm<0>();
        bf16x8 va[TM][3], vb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                va[i][pl] = __builtin_shufflevector(fa[i][pl][0], fa[i][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(va[i][pl]);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                vb[j][pl] = __builtin_shufflevector(fb[j][pl][0], fb[j][pl][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(vb[j][pl]);
            }
        if (next) { pin(raw[0][0]); pin(raw[0][1]); pin(raw[1][0]); pin(raw[1][1]); }
        __builtin_amdgcn_sched_barrier(0);
        // six piece products (smallest terms first); the split of the next k-tile is dealt out between the MFMAs of the last five
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int ap = t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0;
            const int bp = (t == 0 || t == 3 || t == 5) ? 0 : (t == 1 || t == 4) ? 1 : 2;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[i][ap], vb[j][bp], acc[i][j], 0, 0, 0);
            if (t == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (next) split_raw();
            }
        }
        if (next) {
#pragma unroll
            for (int n = 0; n < 20; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);        // five VALU (88 in all)
            }
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) pin(pc[op][pl]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!next) break;
        __builtin_amdgcn_s_barrier();                 // A: every wave has read the planes of tile `it` and the raw rows of tile it + 1
        if (issued < nsteps) { request_tile(); ++issued; }      // tile it + 4 -> the raw stage of tile it + 1
        write_pieces();
        wait_lgkm<0>();
        if (it + 2 < nsteps) wait_tile(it + 2);
        __builtin_amdgcn_s_barrier();                 // B: planes(it + 1) written; raw tile it + 2 landed for every wave
    }

    float* out = p.out + (size_t)split * p.Kc * p.C;
    const bool direct = (p.splits == 1);
    ws_store_tile(acc, reinterpret_cast<float*>(smem_raw), out, m0, n0, p.Kc, p.C, row_w, col_w, li, lh, tid, direct ? p.beta : 0.f);
}

struct WgsPlan { int splits, chunk; };

// Pixel split: two workgroups per CU fit with three stages (72 KB of LDS), three with two (48 KB); one round of workgroups; at
// least 8 k-tiles (128 pixels) per workgroup.
WgsPlan plan_wgs(const sscg_conv_desc* d) {
    const int Kc = d->K, Ng = d->R * d->S * d->C;
    const long npix = (long)d->N * d->P * d->Q;
    const long steps = cdiv(npix, WS_BKP);
    const long tiles = (long)cdiv(Kc, 128) * cdiv(Ng, 128);
    const int force = (d->wgrad_tuning >> 8) & 0xffff;
    const bool two_stages = ((d->wgrad_tuning >> 24) & 1) != 0;       // tuning flag: two copy stages, three workgroups per CU
    long s = force > 0 ? force : (two_stages ? 768 : 512) / tiles;
    if (s > steps / 8) s = steps / 8;
    if (s > 1024) s = 1024;
    if (s < 1) s = 1;
    WgsPlan pl;
    pl.chunk = (int)(cdiv(steps, s) * WS_BKP);
    pl.splits = cdiv(npix, pl.chunk);
    return pl;
}

// Pixel split of the fused 1x1 kernel: one round of two workgroups per CU, at least 8 k-tiles (128 pixels) per workgroup; every split
// costs a partial tile in the workspace and a pass of the reduction over it.
WgsPlan plan_wgf(const sscg_conv_desc* d) {
    const long npix = (long)d->N * d->P * d->Q;
    const long steps = cdiv(npix, WS_BKP);
    const long tiles = (long)cdiv(d->K, 128) * cdiv(d->C, 128);
    const int force = (d->wgrad_tuning >> 8) & 0xffff;
    long s = force > 0 ? force : 512 / tiles;
    if (s > steps / 8) s = steps / 8;
    if (s > 1024) s = 1024;
    if (s < 1) s = 1;
    WgsPlan pl;
    pl.chunk = (int)(cdiv(steps, s) * WS_BKP);
    pl.splits = cdiv(npix, pl.chunk);
    return pl;
}

// 1x1, stride 1, no padding, both sides >= 128 channels: the fused-split kernel (tuning class 3 = conv_wgrad.hip's on-the-fly kernel,
// class 4 = the pre-split planes kernel, for comparison)
bool wgf_applies(const sscg_conv_desc* d) {
    const int cls = d->wgrad_tuning & 0xff;
    if (cls == 3 || cls == 4 || cls == 5) return false;       // (5 = everything as planned except this kernel: A/B aid)
    // 2 M outputs and more (1024 <-> 2048): every element meets enough output columns for the split PASS of the planes kernel to pay
    // (tools/wgrad1x1_bench.py: 228 vs 261 us)
    if ((long)d->K * d->C >= (1L << 21)) return false;
    const long npix = (long)d->N * d->P * d->Q;
    if (npix * (d->K > d->C ? d->K : d->C) * 4 >= (1L << 31)) return false;      // 32-bit buffer offsets, 2 GB = the out-of-range marker
    return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->K >= 128 && d->C >= 128 && d->K % 8 == 0 && d->C % 8 == 0 &&
           npix >= 1024;
}

int launch_wgf(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, size_t ws_bytes, hipStream_t st) {
    const WgsPlan pl = plan_wgf(d);
    const size_t part = pl.splits > 1 ? (size_t)pl.splits * d->K * d->C * sizeof(float) : 0;
    if (part && (!ws || ws_bytes < part)) return SSCG_ERR_WORKSPACE;
    WgfParams p = {};
    p.x = reinterpret_cast<const float*>(x); p.dy = reinterpret_cast<const float*>(dy);
    p.out = pl.splits > 1 ? reinterpret_cast<float*>(ws) : dw;
    p.Kc = d->K; p.C = d->C;
    p.npix = d->N * d->P * d->Q; p.chunk = pl.chunk;
    p.tiles_n = cdiv(p.C, 128); p.tiles = cdiv(p.Kc, 128) * p.tiles_n; p.splits = pl.splits;
    p.beta = pl.splits > 1 ? 0.f : beta;
    SSCG_ENSURE_SMEM(wgradf_kernel, WF_SMEM);
    hipLaunchKernelGGL(wgradf_kernel, dim3(p.tiles * pl.splits), dim3(256), WF_SMEM, st, p);
    SSCG_LAUNCH_CHECK();
    if (pl.splits > 1) return sscg_wgrad_reduce(reinterpret_cast<const float*>(ws), dw, (size_t)d->K * d->C, pl.splits, beta, st);
    return SSCG_OK;
}

}  // namespace

static size_t wgs_planes_bytes(const sscg_conv_desc* d) {
    return ((size_t)d->N * d->H * d->W * d->C + (size_t)d->N * d->P * d->Q * d->K) * 3 * sizeof(bf16);
}

bool sscg_wgrads_applies(const sscg_conv_desc* d) {
    if (d->precision != 2 || d->x_dtype != SSCG_F32 || d->y_dtype != SSCG_F32) return false;
    if (wgf_applies(d)) return true;
    if ((d->wgrad_tuning & 0xff) == 3) return false;      // tuning: class 2 = the on-the-fly split kernel of conv_wgrad.hip
    const int Ng = d->R * d->S * d->C;
    if ((d->wgrad_tuning & 0xff) != 4 && d->R * d->S == 1 && (long)d->K * d->C < (1L << 21)) return false;      // 1x1 outside the fused kernel's reach: too few flops per element for a split PASS to pay (class 3 forces it)
    // the scratch planes live in the caller's per-stream workspace, which only grows: above 1.5 GB of planes (6 bytes per element of
    // x and dy: 256-channel 3x3 layers beyond ~120 M pixels-times-channels) the on-the-fly kernel of conv_wgrad.hip serves the layer
    if (wgs_planes_bytes(d) > ((size_t)3 << 29)) return false;
    return d->K >= 128 && Ng >= 128 && d->K % 8 == 0 && d->C % 8 == 0 && (long)d->N * d->P * d->Q >= 1024;
}

size_t sscg_wgrads_workspace(const sscg_conv_desc* d) {
    if (wgf_applies(d)) {
        const WgsPlan pf = plan_wgf(d);
        return pf.splits > 1 ? (size_t)pf.splits * d->K * d->C * sizeof(float) : 0;
    }
    const WgsPlan pl = plan_wgs(d);
    const size_t part = pl.splits > 1 ? (size_t)pl.splits * d->K * d->R * d->S * d->C * sizeof(float) : 0;
    return ((wgs_planes_bytes(d) + 255) & ~(size_t)255) + part;
}

int sscg_wgrads(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, size_t ws_bytes, hipStream_t st) {
    if (wgf_applies(d)) return launch_wgf(d, x, dy, dw, beta, ws, ws_bytes, st);
    if (!ws || ws_bytes < sscg_wgrads_workspace(d)) return SSCG_ERR_WORKSPACE;
    const WgsPlan pl = plan_wgs(d);
    const size_t nx = (size_t)d->N * d->H * d->W * d->C, ny = (size_t)d->N * d->P * d->Q * d->K;
    bf16* x3 = reinterpret_cast<bf16*>(ws);
    bf16* y3 = x3 + 3 * nx;
    float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + ((wgs_planes_bytes(d) + 255) & ~(size_t)255));
    // the two operands, split once (sscg_split3's rounding)
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv((long)((nx + 7) / 8), 256)), dim3(256), 0, st, reinterpret_cast<const float*>(x), x3, nx, nx);
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv((long)((ny + 7) / 8), 256)), dim3(256), 0, st, reinterpret_cast<const float*>(dy), y3, ny, ny);
    SSCG_LAUNCH_CHECK();
    WgsParams p = {};
    p.x3 = x3; p.dy3 = y3; p.xplane = (long)nx; p.yplane = (long)ny;
    p.out = pl.splits > 1 ? part : dw;
    p.Kc = d->K; p.Ng = d->R * d->S * d->C; p.C = d->C;
    p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.pad_mode = d->pad_mode;
    p.npix = d->N * d->P * d->Q; p.chunk = pl.chunk;
    p.tiles_n = cdiv(p.Ng, 128); p.tiles = cdiv(p.Kc, 128) * p.tiles_n; p.splits = pl.splits;
    p.beta = pl.splits > 1 ? 0.f : beta;
    p.div_pq = make_fastdiv(d->P * d->Q);
    p.div_q = make_fastdiv(d->Q);
    if ((d->wgrad_tuning >> 24) & 1) {
        const size_t smem2 = (size_t)128 * 132 * sizeof(float);      // the staged result tile (> the two copy stages' 48 KB)
        SSCG_ENSURE_SMEM(wgrads_kernel<2>, smem2);
        hipLaunchKernelGGL(wgrads_kernel<2>, dim3(p.tiles * pl.splits), dim3(256), smem2, st, p);
    } else {
        const size_t smem = (size_t)3 * WS_STAGE;      // 72 KB
        SSCG_ENSURE_SMEM(wgrads_kernel<3>, smem);
        hipLaunchKernelGGL(wgrads_kernel<3>, dim3(p.tiles * pl.splits), dim3(256), smem, st, p);
    }
    SSCG_LAUNCH_CHECK();
    if (pl.splits > 1) return sscg_wgrad_reduce(part, dw, (size_t)d->K * p.Ng, pl.splits, beta, st);
    return SSCG_OK;
}
