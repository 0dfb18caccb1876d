// bfloat16 convolutions for BASELINE configs 3/5 (Cityscapes, "bf16"): activations and conv weights live in HBM as
// bf16, the LDS tiles are bf16, the contraction is v_mfma_f32_32x32x16_bf16 with fp32 accumulation; bias, activation,
// split-K partial sums and every weight gradient stay fp32 (the optimiser's master weights and moments are fp32).
//
// Same three products as conv_igemm.hip / conv_wgrad.hip (reference call sites: arch/ops.py:43,49,55-56,68;
// arch/generators.py:325-336,373,388,415; arch/discriminators.py:45,58,70-75 and their autograd, model.py:472,539):
//   forward / data gradient / ConvTranspose forward : D[m][n] = sum_k A[m][k] * B[n][k]   ("KC": both operands k-contiguous)
//   weight gradient                                 : D[m][n] = sum_p A[p][m] * B[p][n]   (operands strided along p)
//
// One k-tile is 64 bf16 = 128 bytes per row - byte for byte the LDS image of the fp32 kernel's 32-float k-tile, so the
// LDS-DMA staging (`global_load_lds_dwordx4`: 8 lanes x 16 B per row, lane-linear LDS image, XOR swizzle of the 16-byte
// slot with (row >> 1) & 7 applied to the SOURCE address and to the fragment read alike) carries over unchanged, and a
// lane's ds_read_b128 is now a complete 8-element MFMA operand: no per-fragment convert, half the HBM/L2/LDS bytes and
// 1/16 of the matrix-core cycles of the fp32 walk.
//
// The weight gradient's operands are contiguous along the OUTPUT axes (channels) and strided along the reduction
// (pixels), but a bf16 MFMA operand is 8 consecutive k of one row: the loader transposes 4-pixel x 8-channel pieces in
// registers (16-bit interleaves) and writes [channel][pixel] rows, after which the contraction loop is the forward's.
#include "common.h"
#include "sscg_internal.h"
#include "reduce_common.h"
#include <type_traits>

namespace {

constexpr int BK = 64;                     // bf16 elements per k-tile (128-byte rows)
typedef __bf16 bf16;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// 256 B of zeros: the source of LDS-DMA lanes that fall into padding / outside the tile
__device__ float sscg_zero_page16[64];

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

#ifndef K16_BUFLD
#define K16_BUFLD 1        // LDS-DMA copies as buffer loads: row offset in one VGPR per copy row (per tap), chunk offset in an SGPR (conv_split.hip KS_BUFLD)
#endif
struct K16Params {
    const bf16* __restrict__ src;    // A source (input for fwd, dy for dgrad), NHWC bf16
    const bf16* __restrict__ wgt;    // [Ng][wKtot] bf16
    const float* __restrict__ bias;  // [Ng] or null
    void* __restrict__ dst;          // [M][Ng] bf16 or fp32
    int out_bf16;
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
    float* __restrict__ part;        // [splits][M - m_tail0][Ng] fp32 when splits > 1
    // per-tile column statistics of the fp32 accumulators (fused norm statistics): [tiles_m][2][Ng][2] doubles or null
    double* __restrict__ stats;
    int stat_L;                      // rows per normalisation group (a tile spans at most two groups: stat_L >= BM)
    double* __restrict__ xstats;     // host side: records of the split rows ([blocks][Ng][2]), written by the split reduction
    unsigned src_bytes, wgt_bytes;   // extents of the two buffer resources (< 2 GB: offset 0x80000000 = out of range = zeros)
    // data gradient only: the backward sums of the normalisation layer whose OUTPUT this launch differentiates (conv_split.hip KsParams)
    const bf16* __restrict__ addend;     // [M][Ng] added to the bf16 result (the gradient another consumer of the same tensor left), or null
    const bf16* __restrict__ bn_x;       // [M][Ng] the layer's input (pre-normalisation), or null
    const float* __restrict__ bn_mean;   // [G][Ng]
    const float* __restrict__ bn_rstd;
    const float* __restrict__ bn_gamma;  // [Ng] or null
    const float* __restrict__ bn_beta;
    double* __restrict__ bn_sums;        // [G][bn_chunks][Ng][2]
    int bn_L, bn_G, bn_chunks, bn_act;
    float bn_slope;
    FastDiv div_tn, div_hw, div_w, div_gl;   // by tiles_n, OH * OW, OW, stat_L / bn_L (launch16): tile and row decode without integer divisions (~35 VALU operations each)
};

// activations of the epilogue in every class but the heads' (32 columns): none / ReLU / LeakyReLU.  tanh - the ResNet generators' 3- and
// 21-channel heads, the U-Net's outermost layer - lives in the 32-column class only (sscg_conv16_*_applies): its software expansion
// (192 v_fmaak per instance) otherwise sits in every instance's epilogue (round 6; conv_split.hip did the same in round 5)
__device__ __forceinline__ float k16_act(float v, int act, float slope) {
    const float neg = act == SSCG_ACT_RELU ? 0.f : (act == SSCG_ACT_LRELU ? v * slope : v);
    return v > 0.f ? v : neg;
}

__device__ __forceinline__ void store_out(void* dst, size_t idx, float v, int out_bf16) {
    if (out_bf16) reinterpret_cast<bf16*>(dst)[idx] = (bf16)v;
    else reinterpret_cast<float*>(dst)[idx] = v;
}

// Fragment reads and the waits of the k-loop are written out by hand: with an LDS-DMA in flight hipcc orders every LDS read it
// can see behind `s_waitcnt vmcnt(0)` (the copy "may alias" the read), which caps the copy pipeline at ONE k-tile ahead.  The
// loop below keeps NSTAGE - 1 tiles in flight: the compiler sees no LDS access inside it, the ordering is explicit
// (`s_waitcnt vmcnt(pieces still allowed in flight)` + `s_barrier` before a tile is read, a barrier before it is overwritten).
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pin(bf16x8& v) { asm volatile("" : "+v"(v)); }      // orders the consumer behind the wait above it

// BS: the data gradient also takes the backward sums of the normalisation layer in front, in its store phase (its own instances: the
// sums' code and registers otherwise ride in every data-gradient launch; a BS instance carries no fan-in code - bf16 tensors never
// ask for both)
template <int MODE, int WM, int WN, int TM, int TN, int NSTAGE, bool BS = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 && TM * TN == 2) ? 4 : 2) void conv16_kernel(K16Params p) {
    constexpr int NT = WM * WN * 64;          // 4 waves (256 threads) or 8 waves (512)
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr bool STAGE_OUT = BN >= 64;      // the narrow head tile writes fp32 / few channels: direct stores
    constexpr int RP = NT / 8;                // rows per loader pass (8 lanes x 16 B per row)
    constexpr int PA = BM / RP;
    constexpr int PB = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0, "whole loader passes");
    constexpr int A_STAGE = BM * BK * 2;      // bytes of one A image
    constexpr int B_STAGE = BN * BK * 2;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [NSTAGE][BM][64] A images, then [NSTAGE][BN][64] B images

    const int tid = threadIdx.x;
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

    const int r0 = tid >> 3;                        // row inside a pass
    const int kq = (tid & 7) ^ swz(r0);             // 16-byte k slot this lane fetches: lands at LDS slot tid & 7 of row r0
    const int wave_id = tid >> 6;

#if K16_BUFLD
    unsigned arow[PA];
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, (short)0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, (short)0, (int)p.wgt_bytes, 0x00020000);
#else
    const bf16* arow[PA];
#endif
    int ay0[PA], ax0[PA];
    bool aok[PA];
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
        const int m = m0 + r0 + ps * RP;
        aok[ps] = m < p.M;
        const int mm = aok[ps] ? m : 0;
        const int img = fd_div(mm, p.div_hw);
        const int rem = mm - img * (p.OH * p.OW);
        const int oy = fd_div(rem, p.div_w);
        const int ox = rem - oy * p.OW;
#if K16_BUFLD
        arow[ps] = (unsigned)img * (unsigned)(p.SH * p.SW * p.Cs) * 2u;
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
#if K16_BUFLD
    unsigned brow[PB];
#else
    const bf16* brow[PB];
#endif
    bool bok[PB];
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) {
        const int n = n0 + r0 + ps * RP;
        bok[ps] = n < p.Ng;
#if K16_BUFLD
        brow[ps] = bok[ps] ? ((unsigned)n * (unsigned)p.wKtot + kq * 8) * 2u : 0x80000000u;       // rows past Ng: zeros
#else
        brow[ps] = p.wgt + (size_t)(bok[ps] ? n : 0) * p.wKtot + kq * 8;
#endif
    }

    const bool reflect = p.pad_mode == 1;
    auto locate = [&](int ps, int tdy, int tdx, int& pix) -> bool {
        bool ok = aok[ps];
        int sy, sx;
        if (MODE == MODE_FWD) {
            sy = ay0[ps] + tdy;
            sx = ax0[ps] + tdx;
            if (reflect) {              // (uniform: only reflection-padded convolutions pay for the mirror arithmetic)
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
            } else if (p.stride == 2) {
                sy = ty >> 1; sx = tx >> 1;
                ok = ok && (((ty | tx) & 1) == 0);
            } else {
                sy = ty / p.stride; sx = tx / p.stride;
                ok = ok && (ty >= 0) && (tx >= 0) && (sy * p.stride == ty) && (sx * p.stride == tx);
            }
        }
        ok = ok && ((unsigned)sy < (unsigned)p.SH) && ((unsigned)sx < (unsigned)p.SW);
        pix = ok ? sy * p.SW + sx : 0;
        return ok;
    };

    const int nk_all = p.Ktot / BK;                          // Cs % 64 == 0: a k-tile never straddles a tap
    const int kt0 = partial ? split * p.ksplit : 0;
    const int kt1 = partial ? min(nk_all, kt0 + p.ksplit) : nk_all;
    const int f_nchunk = p.Cs / BK;
    // the copy front: (tap = (f_ky, f_kx), channel chunk) of the next k-tile to request - wave-uniform, walked incrementally
    int f_chunk, f_ky, f_kx;
    if (kt0 == 0) {                     // (every whole tile: no integer divisions)
        f_chunk = 0; f_ky = 0; f_kx = 0;
    } else {
        const int tap0 = kt0 / f_nchunk;
        f_chunk = kt0 - tap0 * f_nchunk;
        f_ky = tap0 / p.S;
        f_kx = tap0 - f_ky * p.S;
    }
    int f_k = 0;
#if K16_BUFLD
    unsigned aptr[PA];
#else
    const bf16* aptr[PA];
    const bf16* zero = reinterpret_cast<const bf16*>(sscg_zero_page16);
#endif
    [[maybe_unused]] unsigned f_okbits = 0;
    int dma_stage = 0;

    auto set_tap = [&]() {
        const int tdy = f_ky * p.dil, tdx = f_kx * p.dil;
        f_k = ((p.wt_ky0 + f_ky * p.wt_step) * p.wt_S + p.wt_kx0 + f_kx * p.wt_step) * p.Cs;
        f_okbits = 0;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            int pix;
            const bool ok = locate(ps, tdy, tdx, pix);
            f_okbits |= ok ? (1u << ps) : 0u;
#if K16_BUFLD
            aptr[ps] = ok ? arow[ps] + ((unsigned)pix * (unsigned)p.Cs + kq * 8) * 2u : 0x80000000u;        // out of range: zeros
#else
            aptr[ps] = arow[ps] + (size_t)pix * p.Cs + kq * 8;
#endif
        }
    };
    set_tap();
    f_k += f_chunk * BK;

    // The copy of one k-tile = PA + PB LDS-DMA pieces per wave (1 KB each: this wave's 8 rows of a loader pass).
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;
    const int lds_wave = __builtin_amdgcn_readfirstlane(wave_id * 1024);      // byte offset of this wave's rows inside a pass (SGPR)
    constexpr int NPIECE = PA + PB;
    static_assert((NSTAGE - 2) * NPIECE < 64, "vmcnt is a 6-bit counter");
    auto tile_begin = [&]() {
        if (f_chunk == f_nchunk) {       // wave-uniform: next tap
            f_chunk = 0;
            ++f_kx;
            if (f_kx == p.S) { f_kx = 0; ++f_ky; }
            if (f_ky >= p.R) { f_ky = 0; f_kx = 0; }      // past the last tap (never requested)
            set_tap();
        }
    };
    auto piece = [&](int q) {
#if K16_BUFLD
        if (q < PA) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave + q * (RP * 128)),
                                                     16, (int)aptr[q], __builtin_amdgcn_readfirstlane(f_chunk * (BK * 2)), 0, 0);
        } else {
            const int ps = q - PA;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE + lds_wave +
                                                                                                    ps * (RP * 128)),
                                                     16, (int)brow[ps], __builtin_amdgcn_readfirstlane(f_k * 2), 0, 0);
        }
#else
        if (q < PA) {
            const bf16* g = ((f_okbits >> q) & 1u) ? aptr[q] + f_chunk * BK : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave + q * (RP * 128)), 16, 0, 0);
        } else {
            const int ps = q - PA;
            const bf16* g = bok[ps] ? brow[ps] + f_k : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE + lds_wave + ps * (RP * 128)), 16, 0, 0);
        }
#endif
    };
    auto tile_end = [&]() {
        dma_stage = dma_stage + 1 == NSTAGE ? 0 : dma_stage + 1;
        ++f_chunk;
        f_k += BK;
    };
    auto request_tile = [&]() {
        tile_begin();
#pragma unroll
        for (int q = 0; q < NPIECE; ++q) piece(q);
        tile_end();
    };

    const int lane = tid & 63;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave_id / WN;
    const int wn = wave_id % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = kt1 - kt0;
    // prologue: tiles 0 .. NSTAGE - 2 on their way, tile 0 landed
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (s < nk) { request_tile(); ++issued; }
    }
    if (nk >= NSTAGE - 1) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    // fragment addresses: MFMA kk contracts k = 16 kk .. 16 kk + 15 of the tile; lane half h supplies the 8 elements of slot 2 kk + h
    // (rows row_w + i*32 + li: (row >> 1) & 7 == (li >> 1) & 7)
    const int sw = swz(li);
    int foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) foff[kk] = ((kk * 2 + lh) ^ sw) * 16;
    const lds_char* const a_base = lds0 + (row_w + li) * 128;
    const lds_char* const b_base = lds0 + NSTAGE * A_STAGE + (col_w + li) * 128;
    int rd_stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const lds_char* a = a_base + rd_stage * A_STAGE;
        const lds_char* b = b_base + rd_stage * B_STAGE;
        bf16x8 fa[2][TM], fb[2][TN];
        auto read_frags = [&](int kk, int slot) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[slot][i]) : "v"(a + foff[kk]), "n"(i * 32 * 128));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[slot][j]) : "v"(b + foff[kk]), "n"(j * 32 * 128));
        };
        read_frags(0, 0);
        // the tile NSTAGE - 1 ahead goes into the stage that was read in the previous k-step (every wave is past that barrier)
        const bool more = issued < nk;
        if (more) { request_tile(); ++issued; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1;
            if (kk < 3) {
                read_frags(kk + 1, cur ^ 1);
                wait_lgkm<TM + TN>();           // LDS returns in order: all but the reads just issued have landed
            } else {
                wait_lgkm<0>();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) pin(fa[cur][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) pin(fb[cur][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // reads of group kk + 1 stay in front of the MFMAs of group kk
        }
        rd_stage = rd_stage + 1 == NSTAGE ? 0 : rd_stage + 1;
        // tile kt + 1 must have landed; the NSTAGE - 2 tiles behind it may stay in flight (while tiles are still being requested:
        // in the last steps fewer are outstanding and the count says nothing - wait for all)
        if (more) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // Fused normalisation statistics (InstanceNorm / BatchNorm of the layer that follows, arch/ops.py:40-57): column sums of
    // x and x^2 over this tile's rows, taken from the fp32 accumulators (+ bias), in fp64; a tile may straddle ONE group
    // boundary (rows m < gb belong to the tile's first group, the rest to the next one).
    const bool want_stats = MODE == MODE_FWD && p.stats != nullptr && !partial;      // (a data gradient never takes forward statistics)
    const bool want_bsp = BS && MODE == MODE_DGRAD && p.bn_sums != nullptr;          // (the host plans these launches without split-K)
    int gb = 0x7fffffff;
    if (want_stats) {
        const int g0 = fd_div(m0, p.div_gl);
        gb = (g0 + 1) * p.stat_L;
    }
    int bg = 0;
    if (want_bsp) { bg = fd_div(m0, p.div_gl); gb = (bg + 1) * p.bn_L; }
    // tiles inside one group and inside the tensor (nearly all): four consecutive rows are summed in fp32, the 4-row sums in fp64
    const bool slow_stats = want_stats && (m0 + BM > gb || m0 + BM > p.M);
    const bool fast_stats = want_stats && !slow_stats;
    auto fast_sums = [&](int j, float bv, double& s0, double& q0) {
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
    };
    // bf16 results leave through LDS: in the MFMA layout a lane owns single elements of 16 x TM x TN different rows - 64 two-byte
    // stores per lane and tile, each wave-instruction touching 2 x 64 B.  Staged through LDS (the k-loop is over, its images
    // are dead), every thread then writes 16 bytes = 8 consecutive channels of a row: 8x fewer, full-width stores.
    const bool staged = STAGE_OUT && !partial && p.out_bf16;
    auto put_stats = [&](int n, bool nok, double s0, double q0, double s1, double q1) {
        // the two lane halves hold different rows of the same column
        s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
        if (lh == 0 && nok) {
            // record (tile_m, wm): [ (tile_m * WM + wm) ][2 groups][Ng][2]
            double* rec = p.stats + ((size_t)(tile_m * WM + wm) * 2) * p.Ng * 2;
            rec[(size_t)n * 2] = s0; rec[(size_t)n * 2 + 1] = q0;
            rec[((size_t)p.Ng + n) * 2] = s1; rec[((size_t)p.Ng + n) * 2 + 1] = q1;
        }
    };
    if (staged) {
        // The staging tile holds bf16 PAIRS: a lane's registers 4g, 4g+1 (and 4g+2, 4g+3) are two consecutive rows of one column -
        // converted and packed into one word they halve the ds_write_b32 count (the 64 B/clk store path is what a short-k 1x1
        // conv's epilogue waits for) and the tile (33 KB).  Word [row / 2][column]: low half = even row, high half = odd row.
        constexpr int OLD = BN + 4;
        uint32_t* ot = reinterpret_cast<uint32_t*>(smem_raw);
        uint32_t* ob = ot + ((row_w + 4 * lh) / 2) * OLD + col_w + li;        // this lane's corner; every word is a constant offset away
        // A launch without bias and activation (round 6: every convolution of DeepLab - a norm layer follows) stages its accumulators
        // as they are: the bias add and the activation's compare / select chain are six VALU operations per element, ~190 per tile
        // and wave beside the 32 MFMAs of a 256-channel 1x1 reduction.  One uniform branch picks the instance of the loop.
        auto stage_tile = [&](auto plain_tag) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + col_w + j * 32 + li;
            const bool nok = n < p.Ng;
            const float bv = (!PLAIN && p.bias && nok) ? p.bias[n] : 0.f;
            double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
            if (fast_stats) fast_sums(j, bv, s0, q0);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int ro = i * 32 + (e & 3) + 8 * (e >> 2);           // even: rows ro, ro + 1
                    const int m = m0 + row_w + 4 * lh + ro;
                    const float pre0 = PLAIN ? acc[i][j][e] : acc[i][j][e] + bv, pre1 = PLAIN ? acc[i][j][e + 1] : acc[i][j][e + 1] + bv;
                    if (slow_stats) {
                        if (m < p.M) {
                            const double d = (double)pre0;
                            if (m < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
                        }
                        if (m + 1 < p.M) {
                            const double d = (double)pre1;
                            if (m + 1 < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
                        }
                    }
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t pr = {PLAIN ? pre0 : k16_act(pre0, p.act, p.slope), PLAIN ? pre1 : k16_act(pre1, p.act, p.slope)};
                    const bf16x2_t pk = __builtin_convertvector(pr, bf16x2_t);      // RNE, v_cvt_pk_bf16_f32
                    ob[(ro / 2) * OLD + j * 32] = __builtin_bit_cast(uint32_t, pk);
                }
            }
            if (want_stats) {
                // ONE record per tile (round 6; it was one per wave row): the wave rows' sums meet in LDS behind the staged tile and are
                // added in a fixed order after the barrier below - the finalize launch behind every conv -> norm link reads WM x fewer records
                s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
                s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
                if (lh == 0) {
                    double* const sr = reinterpret_cast<double*>(smem_raw + (BM / 2) * OLD * 4) + (wm * BN + col_w + j * 32 + li) * 4;
                    sr[0] = s0; sr[1] = q0; sr[2] = s1; sr[3] = q1;
                }
            }
        }
        };
        if (!p.bias && p.act == SSCG_ACT_NONE) stage_tile(std::true_type{});
        else stage_tile(std::false_type{});
        __syncthreads();
        if (want_stats && tid < BN && n0 + tid < p.Ng) {
            const double* const sr = reinterpret_cast<const double*>(smem_raw + (BM / 2) * OLD * 4);
            double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                a += sr[(w * BN + tid) * 4]; b += sr[(w * BN + tid) * 4 + 1]; c += sr[(w * BN + tid) * 4 + 2]; d += sr[(w * BN + tid) * 4 + 3];
            }
            const int nn = n0 + tid;
            double* rec = p.stats + ((size_t)tile_m * 2) * p.Ng * 2;
            rec[(size_t)nn * 2] = a; rec[(size_t)nn * 2 + 1] = b;
            rec[((size_t)p.Ng + nn) * 2] = c; rec[((size_t)p.Ng + nn) * 2 + 1] = d;
        }
        constexpr int TPR = BN / 8;             // threads per row PAIR (8 channels = two 16-byte row segments each)
        constexpr int RPP = NT / TPR;           // row pairs per pass
        const int c8 = (tid % TPR) * 8;
        const int n = n0 + c8;
        bf16* out = reinterpret_cast<bf16*>(p.dst);
        const uint32_t* src = ot + (tid / TPR) * OLD + c8;
        auto out_row = [&](int m) -> size_t {
            if (MODE == MODE_FWD || p.o_step == 1) return (size_t)m;
            const int img = fd_div(m, p.div_hw);        // parity class of a strided data gradient: rows interleave into dx
            const int rem = m - img * (p.OH * p.OW);
            const int oi = fd_div(rem, p.div_w);
            const int oj = rem - oi * p.OW;
            return (size_t)img * p.o_HW + (size_t)(oi * p.o_step + p.o_a) * p.o_W + oj * p.o_step + p.o_b;
        };
        // dx = dgrad + addend (conv_split.hip's store phase): the two rows' 16-byte segments of every pass are requested up front
        constexpr int NPS = (BM / 2 + RPP - 1) / RPP;
        constexpr bool CAN_JOIN = NPS <= 2;     // the classes the plan uses (64x64, 128x64, 128x128 of 8 waves); host: sscg_conv16_dgrad_add_applies
        constexpr int NAD = CAN_JOIN ? NPS : 1;
        uint4 ad0[NAD], ad1[NAD];
        const bool joins = !BS && CAN_JOIN && MODE == MODE_DGRAD && p.addend != nullptr;      // (host: Ng % 8 == 0; never together with the sums on bf16 tensors)
        if (joins) {
#pragma unroll
            for (int ps = 0; ps < NAD; ++ps) {
                const int m = m0 + 2 * (tid / TPR + ps * RPP);
                ad0[ps] = uint4{0u, 0u, 0u, 0u}; ad1[ps] = uint4{0u, 0u, 0u, 0u};
                if ((tid / TPR + ps * RPP) < BM / 2 && n < p.Ng) {
                    if (m < p.M) ad0[ps] = *reinterpret_cast<const uint4*>(p.addend + out_row(m) * p.Ng + n);
                    if (m + 1 < p.M) ad1[ps] = *reinterpret_cast<const uint4*>(p.addend + out_row(m + 1) * p.Ng + n);
                }
            }
        }
        // Backward sums of the normalisation layer in front, taken HERE (round 6; conv_split.hip's store phase): a thread owns eight fixed
        // channels of whole rows - the layer's input arrives as 16-byte row segments like the result leaves, requested up front with
        // the addend's; sum g and sum g * xhat in fp32 over the thread's <= 4 rows, across the wave's row lanes by shuffles, across
        // the waves in fp64 through LDS: ONE record per tile and group.  A tile that straddles a group boundary takes the rows of
        // its second group in a sweep of its own (it reads its own dx rows back).
        uint4 xr0[NAD], xr1[NAD];
        float bmu[8], brs[8], bga[8], bbe[8], bsl[8], bql[8];
        const float neg_scale = p.bn_act == SSCG_ACT_RELU ? 0.f : (p.bn_act == SSCG_ACT_LRELU ? p.bn_slope : 1.f);
        const bool bsp = BS && CAN_JOIN && want_bsp;
        auto bsp_params = [&](int g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { bmu[e] = 0.f; brs[e] = 0.f; bga[e] = 1.f; bbe[e] = 0.f; bsl[e] = 0.f; bql[e] = 0.f; }
            if (n < p.Ng) {
                ld8<float>(p.bn_mean + (size_t)g * p.Ng + n, bmu);
                ld8<float>(p.bn_rstd + (size_t)g * p.Ng + n, brs);
                if (p.bn_gamma) { ld8<float>(p.bn_gamma + n, bga); ld8<float>(p.bn_beta + n, bbe); }
            }
        };
        auto bsp_row = [&](const uint4& v, const uint4& x) {
            const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int e = 2 * w + hh;
                    const float vv = __uint_as_float(hh ? (vw[w] & 0xffff0000u) : (vw[w] << 16));
                    const float xx = __uint_as_float(hh ? (xw[w] & 0xffff0000u) : (xw[w] << 16));
                    const float xh = (xx - bmu[e]) * brs[e];
                    const float ym = xh * bga[e] + bbe[e];
                    const float gg = ym > 0.f ? vv : vv * neg_scale;
                    bsl[e] += gg; bql[e] = fmaf(gg, xh, bql[e]);
                }
            }
        };
        auto bsp_put = [&](int g, int chunk) {
            // row lanes of a wave (lanes TPR apart), then the waves through LDS behind the staged tile, then one record
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                for (int o = TPR; o < 64; o <<= 1) { bsl[e] += __shfl_xor(bsl[e], o, 64); bql[e] += __shfl_xor(bql[e], o, 64); }
            }
            float* const sf = reinterpret_cast<float*>(smem_raw + (BM / 2) * OLD * 4);       // [NT / 64 waves][BN][2]
            __syncthreads();
            if ((tid & 63) < TPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sf[((tid >> 6) * BN + c8 + e) * 2] = bsl[e]; sf[((tid >> 6) * BN + c8 + e) * 2 + 1] = bql[e]; }
            }
            __syncthreads();
            if (tid < BN && n0 + tid < p.Ng) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) { a += (double)sf[(w * BN + tid) * 2]; b += (double)sf[(w * BN + tid) * 2 + 1]; }
                double* r = p.bn_sums + (((size_t)g * p.bn_chunks + chunk) * p.Ng + n0 + tid) * 2;
                r[0] = a; r[1] = b;
            }
        };
        if (bsp) {
            bsp_params(bg);
#pragma unroll
            for (int ps = 0; ps < NAD; ++ps) {
                const int m = m0 + 2 * (tid / TPR + ps * RPP);
                xr0[ps] = uint4{0u, 0u, 0u, 0u}; xr1[ps] = uint4{0u, 0u, 0u, 0u};
                if ((tid / TPR + ps * RPP) < BM / 2 && n < p.Ng) {
                    if (m < p.M) xr0[ps] = *reinterpret_cast<const uint4*>(p.bn_x + (size_t)m * p.Ng + n);
                    if (m + 1 < p.M) xr1[ps] = *reinterpret_cast<const uint4*>(p.bn_x + (size_t)(m + 1) * p.Ng + n);
                }
            }
        }
        auto add2 = [](uint32_t v, uint32_t a) -> uint32_t {       // two bf16 sums, each rounded to nearest even (= the add kernel's)
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t sm = {__uint_as_float(v << 16) + __uint_as_float(a << 16),
                                __uint_as_float(v & 0xffff0000u) + __uint_as_float(a & 0xffff0000u)};
            return __builtin_bit_cast(uint32_t, __builtin_convertvector(sm, bf16x2_t));
        };
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int pr = tid / TPR + ps * RPP;
            if ((BM / 2) % RPP != 0 && pr >= BM / 2) continue;
            const int m = m0 + 2 * pr;
            if (m >= p.M || n >= p.Ng) continue;
            const uint4 w0 = *reinterpret_cast<const uint4*>(src + ps * RPP * OLD);
            const uint4 w1 = *reinterpret_cast<const uint4*>(src + ps * RPP * OLD + 4);
            uint4 lo, hi;      // v_perm_b32 D, S0, S1, sel: byte i of D = byte sel[i] of {S0 (bytes 4-7), S1 (bytes 0-3)}
            lo.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x05040100u); hi.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x07060302u);
            lo.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x05040100u); hi.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x07060302u);
            lo.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x05040100u); hi.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x07060302u);
            lo.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x05040100u); hi.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x07060302u);
            if (joins) {
                lo.x = add2(lo.x, ad0[ps % NAD].x); lo.y = add2(lo.y, ad0[ps % NAD].y); lo.z = add2(lo.z, ad0[ps % NAD].z); lo.w = add2(lo.w, ad0[ps % NAD].w);
                hi.x = add2(hi.x, ad1[ps % NAD].x); hi.y = add2(hi.y, ad1[ps % NAD].y); hi.z = add2(hi.z, ad1[ps % NAD].z); hi.w = add2(hi.w, ad1[ps % NAD].w);
            }
            if (bsp) {
                if (m < gb) bsp_row(lo, xr0[ps % NAD]);
                if (m + 1 < p.M && m + 1 < gb) bsp_row(hi, xr1[ps % NAD]);
            }
            const bool full = n + 8 <= p.Ng;
            bf16* r0 = out + out_row(m) * p.Ng + n;
            if (full) {
                *reinterpret_cast<uint4*>(r0) = lo;
            } else {
                const uint32_t wv[4] = {lo.x, lo.y, lo.z, lo.w};
                for (int e = 0; e < 8 && n + e < p.Ng; ++e) reinterpret_cast<uint16_t*>(r0)[e] = (uint16_t)(wv[e >> 1] >> (16 * (e & 1)));
            }
            if (m + 1 < p.M) {
                bf16* r1 = out + out_row(m + 1) * p.Ng + n;
                if (full) {
                    *reinterpret_cast<uint4*>(r1) = hi;
                } else {
                    const uint32_t wv[4] = {hi.x, hi.y, hi.z, hi.w};
                    for (int e = 0; e < 8 && n + e < p.Ng; ++e) reinterpret_cast<uint16_t*>(r1)[e] = (uint16_t)(wv[e >> 1] >> (16 * (e & 1)));
                }
            }
        }
        if (bsp) {
            bsp_put(bg, tile_m - (int)(((long)bg * p.bn_L) / BM));
            if (m0 + BM > gb && bg + 1 < p.bn_G) {       // (workgroup-uniform, one tile per group) rows of the NEXT group: its first record
                bsp_params(bg + 1);
                if (n + 8 <= p.Ng) {
#pragma unroll 1
                    for (int pr = tid / TPR; pr < BM / 2; pr += RPP) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int m = m0 + 2 * pr + hh;
                            if (m >= gb && m < p.M)
                                bsp_row(*reinterpret_cast<const uint4*>(out + out_row(m) * p.Ng + n), *reinterpret_cast<const uint4*>(p.bn_x + (size_t)m * p.Ng + n));
                        }
                    }
                }
                bsp_put(bg + 1, 0);
            }
        }
        return;
    }
    if (STAGE_OUT && partial && (p.Ng & 3) == 0) {
        // A partial tile of a split-K tail (fp32 sums for the reduction pass) leaves through LDS as well: [row][BN + 4] floats, then
        // 16-byte row segments (conv_split.hip: the tail's 100-200 workgroups are the launch's last; their MFMA-layout stores - 32 to
        // 64 four-byte stores per lane, each wave-instruction touching 2 x 128 bytes - were what the launch ended on)
        constexpr int OLDF = BN + 4;
        float* otf = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    otf[(row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * OLDF + col_w + j * 32 + li] = acc[i][j][e];
        __syncthreads();
        constexpr int TPRF = BN / 4, RPPF = NT / TPRF;
        const int c4 = (tid % TPRF) * 4;
        const int n = n0 + c4;
        if (n < p.Ng) {
            float* const base = p.part + ((long)split * (p.M - p.m_tail0) - p.m_tail0) * (long)p.Ng;
#pragma unroll 4
            for (int r = tid / TPRF; r < BM; r += RPPF) {
                const int m = m0 + r;
                if (m >= p.M) break;
                *reinterpret_cast<f32x4*>(base + (size_t)m * p.Ng + n) = *reinterpret_cast<const f32x4*>(otf + r * OLDF + c4);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + col_w + j * 32 + li;
        const bool nok = n < p.Ng;
        const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
        double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
        if (fast_stats) fast_sums(j, bv, s0, q0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < p.M && nok) {
                    if (partial) {
                        p.part[((size_t)split * (p.M - p.m_tail0) + (m - p.m_tail0)) * p.Ng + n] = acc[i][j][e];
                    } else {
                        const float pre = acc[i][j][e] + bv;
                        if (slow_stats) {
                            const double d = (double)pre;
                            if (m < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
                        }
                        size_t row = (size_t)m;
                        if (MODE == MODE_DGRAD && p.o_step != 1) {   // parity class of a strided data gradient: rows interleave into dx
                            const int img = fd_div(m, p.div_hw);
                            const int rem = m - img * (p.OH * p.OW);
                            const int oi = fd_div(rem, p.div_w);
                            const int oj = rem - oi * p.OW;
                            row = (size_t)img * p.o_HW + (size_t)(oi * p.o_step + p.o_a) * p.o_W + oj * p.o_step + p.o_b;
                        }
                        store_out(p.dst, row * p.Ng + n, STAGE_OUT ? k16_act(pre, p.act, p.slope) : sscg_act(pre, p.act, p.slope), p.out_bf16);
                    }
                }
            }
        }
        if (want_stats) put_stats(n, nok, s0, q0, s1, q1);
    }
}

// y[i] = act(sum_s part[s][i] + bias[i % Ng]) (fixed order => deterministic)
__global__ __launch_bounds__(256) void k16_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, void* __restrict__ y,
                                                          int out_bf16, size_t n, int Ng, int splits, int act, float slope) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    if (bias) s += bias[(int)(i % Ng)];
    store_out(y, i, sscg_act(s, act, slope), out_bf16);
}

// ---- host-side plan
// tile classes: block tile, waves, copy stages (k-tiles in LDS).  The deep classes hold 2-3 tiles in flight per workgroup.
enum { CFG_128x128 = 0, CFG_64x64 = 1, CFG_128x32 = 2, CFG_128x64 = 3, CFG_128x128_W8 = 4, CFG_256x128_W8 = 5, NCFG16 = 6 };
const int C16_BM[NCFG16] = {128, 64, 128, 128, 128, 256};
const int C16_BN[NCFG16] = {128, 64, 32, 64, 128, 128};
const int C16_WM[NCFG16] = {2, 2, 4, 2, 2, 4};      // wave rows of a tile (= statistics records per tile row)

// Measured on the step's shapes (tools/conv16_bench.py, profiles/r02_conv16_shapes.txt).  What decides the rate is the number of
// waves per SIMD, not the depth of the copy pipeline or the tile's arithmetic intensity: 128x128 tiles as 4 waves of 64x64 (2
// workgroups = 2 waves per SIMD) reach 600 TF/s on the 34320-row 256-channel 3x3, the same tile as 8 waves of 64x32 (4 per SIMD)
// 710; three or four k-tiles in flight with ONE workgroup per CU: 470; 256x128 tiles: no gain.  Few output channels: 128x64 /
// 64x64 tiles of 4 waves.
// `tuning` = sscg_conv_desc.tuning: bits 0..7 = 1 + forced tile class
int choose16(long M, int Ng, int Ktot, int tuning) {
    const int forced = (tuning & 0xff) - 1;
    if (forced >= 0 && forced < NCFG16 && Ng > 32) return forced;
    (void)Ktot;
    if (Ng <= 32) return CFG_128x32;
    const long tm = cdiv(M, 128);
    if (Ng <= 64) return tm >= 384 ? CFG_128x64 : CFG_64x64;
    if (tm * cdiv(Ng, 128) >= 384) return CFG_128x128_W8;
    return CFG_64x64;
}

struct K16Split { int splits, ksplit, full_tiles, m_tail0; };

// Which tiles are cut along K (same policy as conv_igemm.hip): few-channel heads on few rows split every tile; 64x64
// launches split only the tail beyond the last whole round of 256 workgroups.
K16Split plan16_raw(long M, int Ng, int Ktot, int tuning);

// stat_L > 0: the launch also produces normalisation statistics; the rows of split tiles are summed separately as ONE
// extra group of records, so they must lie in one normalisation group (else the launch is not split).
K16Split plan16(long M, int Ng, int Ktot, int tuning, long stat_L = 0) {
    K16Split r = plan16_raw(M, Ng, Ktot, tuning);
    if (stat_L > 0 && r.splits > 1 && (r.full_tiles == 0 || r.m_tail0 / stat_L != (M - 1) / stat_L)) {
        const int cfg = choose16(M, Ng, Ktot, tuning);
        r.splits = 1; r.ksplit = Ktot / BK;
        r.full_tiles = cdiv(M, C16_BM[cfg]) * cdiv(Ng, C16_BN[cfg]); r.m_tail0 = (int)M;
    }
    return r;
}

K16Split plan16_raw(long M, int Ng, int Ktot, int tuning) {
    const int nk = Ktot / BK;
    const int cfg = choose16(M, Ng, Ktot, tuning);
    const int bm = C16_BM[cfg], bn = C16_BN[cfg];
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(Ng, bn);
    const int tiles = tiles_m * tiles_n;
    K16Split r = {1, nk, tiles, (int)M};
    if (Ng <= 32) {
        if (tiles >= 256 || nk < 16) return r;
        int s = cdiv(512, tiles);
        if (s > nk / 4) s = nk / 4;
        if (s > 32) s = 32;
        if (s < 2) return r;
        r.ksplit = cdiv(nk, s);
        r.splits = cdiv(nk, r.ksplit);
        r.full_tiles = 0; r.m_tail0 = 0;
        return r;
    }
    // 64x64 and 128x128 launches: only the TAIL beyond the last whole round of 256 workgroups is cut along K (8712 rows x 256
    // channels = 548 tiles of 64x64; 34320 rows = 538 tiles of 128x128: 512 run whole, two per CU side by side, the other
    // 26 would keep a tenth of the chip busy for a whole tile time)
    if (cfg == CFG_128x32 || cfg == CFG_128x64 || nk < 8 || tiles > 2300) return r;
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

size_t split16_bytes(const K16Split& sp, long M, int Ng) {
    return sp.splits > 1 ? (size_t)sp.splits * (M - sp.m_tail0) * Ng * sizeof(float) : 0;
}

template <int MODE, int WM, int WN, int TM, int TN, int NSTAGE, bool BS = false>
int launch16(const K16Params& p0, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int NT = WM * WN * 64;
    K16Params p = p0;
    p.tiles_n = cdiv(p.Ng, BN);
    const int tiles_m = cdiv(p.M, BM);
    p.tiles = tiles_m * p.tiles_n;
    p.div_tn = make_fastdiv(p.tiles_n);
    p.div_hw = make_fastdiv(p.OH * p.OW);
    p.div_w = make_fastdiv(p.OW);
    p.div_gl = make_fastdiv((BS && MODE == MODE_DGRAD && p.bn_sums != nullptr) ? p.bn_L : (p.stat_L > 0 ? p.stat_L : 1));
    size_t smem = (size_t)NSTAGE * (BM + BN) * BK * sizeof(bf16);
    const size_t stage = BN >= 64 ? (size_t)(BM / 2) * (BN + 4) * sizeof(uint32_t) + (size_t)WM * BN * 4 * sizeof(double) : 0;      // output tile of the staged epilogue (bf16 row pairs) + the wave rows' statistics behind it
    if (stage > smem) smem = stage;
    const size_t stage_f = (BN >= 64 && p.splits > 1) ? (size_t)BM * (BN + 4) * sizeof(float) : 0;      // fp32 tile of a split-K tail's partial workgroups
    if (stage_f > smem) smem = stage_f;
    auto kern = conv16_kernel<MODE, WM, WN, TM, TN, NSTAGE, BS>;
    SSCG_ENSURE_SMEM((kern), smem);
    if (p.splits <= 1) { p.full_tiles = p.tiles; p.m_tail0 = p.M; }
    const int grid = p.full_tiles + (p.tiles - p.full_tiles) * p.splits;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, st, p);
    SSCG_LAUNCH_CHECK();
    if (p.splits > 1) {
        const size_t n = (size_t)(p.M - p.m_tail0) * p.Ng;
        const size_t esz = p.out_bf16 ? 2 : 4;
        void* yt = reinterpret_cast<char*>(p.dst) + (size_t)p.m_tail0 * p.Ng * esz;
        if (p.xstats)
            return launch_split_reduce_stats(p.part, p.bias, yt, p.out_bf16, p.M - p.m_tail0, p.Ng, p.splits, p.act, p.slope, p.xstats, st);
        hipLaunchKernelGGL(k16_reduce_kernel, dim3(cdiv((long)n, 256)), dim3(256), 0, st, p.part, p.bias, yt, p.out_bf16, n, p.Ng,
                           p.splits, p.act, p.slope);
        SSCG_LAUNCH_CHECK();
    }
    return SSCG_OK;
}

template <int MODE>
int dispatch16(const K16Params& p, int tuning, hipStream_t st) {
    if (MODE == MODE_DGRAD && p.bn_sums != nullptr) {       // the instances that carry the backward sums' epilogue
        switch (choose16(p.M, p.Ng, p.Ktot, tuning)) {
            case CFG_128x128: return launch16<MODE_DGRAD, 2, 2, 2, 2, 2, true>(p, st);
            case CFG_64x64: return launch16<MODE_DGRAD, 2, 2, 1, 1, 2, true>(p, st);
            case CFG_128x32: return launch16<MODE_DGRAD, 4, 1, 1, 1, 2, true>(p, st);
            case CFG_128x64: return launch16<MODE_DGRAD, 2, 2, 2, 1, 2, true>(p, st);
            case CFG_128x128_W8: return launch16<MODE_DGRAD, 2, 4, 2, 1, 2, true>(p, st);
            case CFG_256x128_W8: return launch16<MODE_DGRAD, 4, 2, 2, 2, 2, true>(p, st);
            default: return SSCG_ERR_BAD_ARG;
        }
    }
    switch (choose16(p.M, p.Ng, p.Ktot, tuning)) {
        case CFG_128x128: return launch16<MODE, 2, 2, 2, 2, 2>(p, st);
        case CFG_64x64: return launch16<MODE, 2, 2, 1, 1, 2>(p, st);
        case CFG_128x32: return launch16<MODE, 4, 1, 1, 1, 2>(p, st);
        case CFG_128x64: return launch16<MODE, 2, 2, 2, 1, 2>(p, st);
        case CFG_128x128_W8: return launch16<MODE, 2, 4, 2, 1, 2>(p, st);      // 8 waves of 64x32: 4 waves per SIMD with 2 workgroups per CU
        case CFG_256x128_W8: return launch16<MODE, 4, 2, 2, 2, 2>(p, st);      // 8 waves of 64x64: 96 KB of LDS, one workgroup per CU
        default: return SSCG_ERR_BAD_ARG;
    }
}

void dense_taps(K16Params& p) {
    p.pad_x = p.pad; p.wKtot = p.Ktot;
    p.wt_ky0 = 0; p.wt_kx0 = 0; p.wt_step = 1; p.wt_S = p.S;
    p.o_step = 1; p.o_a = 0; p.o_b = 0; p.o_W = 0; p.o_HW = 0;
}

bool dgrad16_by_parity(const sscg_conv_desc* d) { return d->stride == 2 && d->dil == 1 && d->pad_mode == 0; }

}  // namespace

// ---- entry points used by conv_igemm.hip's dispatch (same argument meaning as the public sscg_conv2d_* functions)
// (the copies address both operands through 32-bit buffer offsets; a masked row's offset, 2 GB, must lie outside the tensor)
static bool k16_extents_ok(const sscg_conv_desc* d, bool dgrad) {
    const size_t src = (dgrad ? (size_t)d->N * d->P * d->Q * d->K : (size_t)d->N * d->H * d->W * d->C) * sizeof(bf16);
    return src < ((size_t)1 << 31) && (size_t)d->K * d->R * d->S * d->C * sizeof(bf16) < ((size_t)1 << 31);
}

bool sscg_conv16_fwd_applies(const sscg_conv_desc* d) {
    return d->x_dtype == SSCG_BF16 && d->w_dtype == SSCG_BF16 && d->C % BK == 0 && (d->act != SSCG_ACT_TANH || d->K <= 32) &&
           k16_extents_ok(d, false);
}

bool sscg_conv16_dgrad_applies(const sscg_conv_desc* d) {
    return d->y_dtype == SSCG_BF16 && d->w_dtype == SSCG_BF16 && d->K % BK == 0 && k16_extents_ok(d, true);
}

// geometry of the statistics records of a forward launch (records = [tiles_m * wm][2 groups][K][2] doubles)
bool sscg_conv16_stats_geometry(const sscg_conv_desc* d, long L, int* bm, int* wm, int* tiles_n, int* splits, int* full_tiles, int* m_tail0) {
    const long M = (long)d->N * d->P * d->Q;
    const int cfg = choose16(M, d->K, d->R * d->S * d->C, d->tuning);
    if (cfg == CFG_128x32 || L < C16_BM[cfg]) return false;
    *bm = C16_BM[cfg];
    *wm = d->y_dtype == SSCG_BF16 ? 1 : C16_WM[cfg];       // bf16 results leave through a staged tile: its wave rows' sums meet in LDS (one record per tile)
    *tiles_n = cdiv(d->K, C16_BN[cfg]);
    K16Split sp = plan16(M, d->K, d->R * d->S * d->C, d->tuning, L);
    *splits = sp.splits; *full_tiles = sp.full_tiles; *m_tail0 = sp.m_tail0;
    return true;
}

size_t sscg_conv16_fwd_workspace(const sscg_conv_desc* d, long stat_L) {
    const long M = (long)d->N * d->P * d->Q;
    return split16_bytes(plan16(M, d->K, d->R * d->S * d->C, d->tuning, stat_L), M, d->K);
}

size_t sscg_conv16_dgrad_workspace(const sscg_conv_desc* d) {
    if (dgrad16_by_parity(d)) return 0;
    const long M = (long)d->N * d->H * d->W;
    return split16_bytes(plan16(M, d->C, d->R * d->S * d->K, d->tuning), M, d->C);
}

int sscg_conv16_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* stats, long stat_L,
                    double* xstats, void* ws, size_t ws_bytes, hipStream_t st) {
    K16Params p = {};
    p.src = reinterpret_cast<const bf16*>(x); p.wgt = reinterpret_cast<const bf16*>(w); p.bias = bias; p.dst = y;
    p.src_bytes = (unsigned)((size_t)d->N * d->H * d->W * d->C * sizeof(bf16));
    p.wgt_bytes = (unsigned)((size_t)d->K * d->R * d->S * d->C * sizeof(bf16));
    p.out_bf16 = d->y_dtype == SSCG_BF16;
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Cs = d->C; p.Ktot = d->R * d->S * d->C;
    p.SH = d->H; p.SW = d->W; p.OH = d->P; p.OW = d->Q;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = d->pad_mode; p.act = d->act; p.slope = d->slope;
    p.stats = stats; p.stat_L = (int)stat_L; p.xstats = xstats;
    dense_taps(p);
    K16Split sp = plan16(p.M, p.Ng, p.Ktot, d->tuning, stats ? stat_L : 0);
    if (sp.splits > 1 && (!ws || ws_bytes < split16_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch16<MODE_FWD>(p, d->tuning, st);
}

// backward sums of the normalisation layer in front from this launch's epilogue (conv_split.hip sscg_convs_bsums_geometry)
bool sscg_conv16_bsums_geometry(const sscg_conv_desc* d, int G, long L, int* bm, int* wm, int* chunks) {
    if (!sscg_conv16_dgrad_applies(d) || d->x_dtype != SSCG_BF16 || dgrad16_by_parity(d) || d->stride != 1) return false;
    const long M = (long)d->N * d->H * d->W;
    if (G <= 0 || L <= 0 || (long)G * L != M) return false;
    const int cfg = choose16(M, d->C, d->R * d->S * d->K, d->tuning);
    if (!(cfg == CFG_64x64 || cfg == CFG_128x64 || cfg == CFG_128x128_W8) || L < C16_BM[cfg] || d->C % 8 != 0) return false;   // (the store phase's classes, whole 16-byte segments)
    *bm = C16_BM[cfg];
    *wm = 1;                                // one record per tile and group (the sums are taken in the store phase, per workgroup)
    *chunks = (int)(cdiv(L, (long)C16_BM[cfg]) + 1);
    return true;
}

// dx = dgrad + addend in the bf16 store phase: bf16 result tiles of at least 64 columns, whole 16-byte row segments
bool sscg_conv16_dgrad_add_applies(const sscg_conv_desc* d) {
    if (!sscg_conv16_dgrad_applies(d) || d->x_dtype != SSCG_BF16 || d->C <= 32 || d->C % 8 != 0) return false;
    const long M = dgrad16_by_parity(d) ? (long)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (long)d->N * d->H * d->W;
    const int cfg = choose16(M, d->C, d->R * d->S * d->K, d->tuning);
    return cfg == CFG_64x64 || cfg == CFG_128x64 || cfg == CFG_128x128_W8;
}

int sscg_conv16_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias, void* dx, int act, float slope,
                      void* ws, size_t ws_bytes, hipStream_t st, const sscg_bsums* bs, const void* addend) {
    if (act == SSCG_ACT_TANH && d->C > 32) return SSCG_ERR_UNSUPPORTED;      // (tanh lives in the 32-column class only)
    K16Params p = {};
    bool fused = false;
    if (addend) {
        if (bs || bias || act != SSCG_ACT_NONE || !sscg_conv16_dgrad_add_applies(d)) return SSCG_ERR_UNSUPPORTED;
        p.addend = reinterpret_cast<const bf16*>(addend);
        fused = true;           // (never split: the partial tiles' reduction does not know the addend)
    }
    if (bs) {
        int bm, wm, chunks;
        if (bias || act != SSCG_ACT_NONE || !sscg_conv16_bsums_geometry(d, bs->G, bs->L, &bm, &wm, &chunks)) return SSCG_ERR_UNSUPPORTED;
        p.bn_x = reinterpret_cast<const bf16*>(bs->nx); p.bn_mean = bs->mean; p.bn_rstd = bs->rstd; p.bn_gamma = bs->gamma; p.bn_beta = bs->beta;
        p.bn_sums = reinterpret_cast<double*>(bs->sums); p.bn_L = (int)bs->L; p.bn_G = bs->G; p.bn_chunks = chunks;
        p.bn_act = bs->act; p.bn_slope = bs->slope;
        fused = true;
    }
    p.src = reinterpret_cast<const bf16*>(dy); p.wgt = reinterpret_cast<const bf16*>(wt); p.bias = bias; p.dst = dx;
    p.src_bytes = (unsigned)((size_t)d->N * d->P * d->Q * d->K * sizeof(bf16));
    p.wgt_bytes = (unsigned)((size_t)d->K * d->R * d->S * d->C * sizeof(bf16));
    p.out_bf16 = d->x_dtype == SSCG_BF16;
    p.M = d->N * d->H * d->W; p.Ng = d->C; p.Cs = d->K; p.Ktot = d->R * d->S * d->K;
    p.SH = d->P; p.SW = d->Q; p.OH = d->H; p.OW = d->W;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = 0; p.act = act; p.slope = slope;
    dense_taps(p);
    if (dgrad16_by_parity(d)) {
        // stride 2: four parity classes, each a stride-1 data gradient over its sub-lattice of taps (conv_igemm.hip)
        p.splits = 1; p.ksplit = 0; p.part = nullptr;
        p.stride = 1; p.wt_step = 2; p.wt_S = d->S;
        p.o_step = 2; p.o_W = d->W; p.o_HW = d->H * d->W;
        for (int a = 0; a < 2; ++a) {
            for (int b = 0; b < 2; ++b) {
                const int Ha = (d->H - a + 1) / 2, Wb = (d->W - b + 1) / 2;
                if (Ha <= 0 || Wb <= 0) continue;
                const int ky0 = (a + d->pad) & 1, kx0 = (b + d->pad) & 1;
                K16Params q = p;
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
                int rc = dispatch16<MODE_DGRAD>(q, d->tuning, st);
                if (rc) return rc;
            }
        }
        return SSCG_OK;
    }
    K16Split sp = plan16(p.M, p.Ng, p.Ktot, d->tuning);
    if (fused && sp.splits > 1) {          // fused sums: the launch is not split (its partial tiles would need the sums in the reduction)
        const int cfg = choose16(p.M, p.Ng, p.Ktot, d->tuning);
        sp.splits = 1; sp.ksplit = p.Ktot / BK;
        sp.full_tiles = cdiv(p.M, C16_BM[cfg]) * cdiv(p.Ng, C16_BN[cfg]); sp.m_tail0 = p.M;
    }
    if (sp.splits > 1 && (!ws || ws_bytes < split16_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch16<MODE_DGRAD>(p, d->tuning, st);
}

// =====================================================================================================================
// weight gradient, both operands bf16:  dW[k][tap][c] = sum_p dy[p][k] * x[src(p, tap)][c]   (fp32 out)
namespace {

constexpr int BKP = 64;      // pixels per k-step

struct Wg16Params {
    const bf16* __restrict__ x;
    const bf16* __restrict__ dy;
    float* __restrict__ out;      // dw (splits == 1) or workspace [splits][Kc][Ng]
    int Kc, Ng, C;
    int H, W, P, Q, S;
    int stride, pad, dil, pad_mode;
    int npix, chunk;
    int tiles_n, tiles, splits;
    float beta;
    FastDiv div_pq, div_q;
    int buf_ok;                   // both operand tensors < 2 GB: their pixel rows are copied by buffer loads (32-bit offsets)
};

// ---- weight gradient, LDS-DMA + transpose-read version -------------------------------------------------------------------------
// The operands are [pixel][channel] in HBM: contiguous along the OUTPUT axes, strided along the reduction (pixels), while an
// MFMA operand is 8 consecutive k (= pixels) of one output row.  (Rounds 1-5 kept a first version that transposed 8x8 pieces in
// registers - 32 v_perm + 8 ds_write_b128 per thread and k-step, the ds_write path alone ~400 of a k-step's 512 MFMA cycles: 350-450
// against 480-780 TF/s; deleted in round 6.)  gfx950's `ds_read_b64_tr_b16` transposes on the way OUT of LDS: a 16-lane group reads a [4 pixels][16 channels]
// block (lane i supplies the address of 4 consecutive channels of pixel i >> 2) and lane i receives the 4 pixels of channel i.
// So the [pixel][channel] rows go to LDS as they are, by LDS-DMA (no staging registers, no ds_write), NSTAGE - 1 k-tiles in
// flight, and the fragments come from two transpose-reads per operand.
//   LDS image of one operand and stage: [64 pixels][BT channels] bf16, rows of BT * 2 bytes; the 16-byte chunk q of pixel row r
//   sits at chunk q ^ wsw(r): the four pixel rows a transpose-read touches land in four different quarters of the 256-byte
//   bank line (plain rows of 256 B would hit the same 64 bytes four times).
template <int BT> __device__ __forceinline__ int wsw(int pixel) { return BT == 128 ? ((pixel & 3) << 2) : (((pixel >> 1) & 1) << 2); }

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) char wg_lds_char;

template <int KK, int TM, int TN, int A_ROWB, int B_ROWB>
__device__ __forceinline__ void wg_read_frags(bf16x4 (&fa)[TM][2], bf16x4 (&fb)[TN][2], const wg_lds_char* a, const wg_lds_char* b,
                                              const int (&a_off)[TM], const int (&b_off)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][0]) : "v"(a + a_off[i]), "n"((16 * KK) * A_ROWB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[i][1]) : "v"(a + a_off[i]), "n"((16 * KK + 4) * A_ROWB));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][0]) : "v"(b + b_off[j]), "n"((16 * KK) * B_ROWB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[j][1]) : "v"(b + b_off[j]), "n"((16 * KK + 4) * B_ROWB));
    }
}

template <int WM, int WN, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 8 && TM * TN == 2) ? 4 : 2) void wgrad16t_kernel(Wg16Params p) {
    constexpr int NW = WM * WN;               // 4 or 8 waves
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    static_assert((NW == 4 || NW == 8) && (BM == 64 || BM == 128) && (BN == 64 || BN == 128), "64- or 128-channel operand tiles");
    constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;            // bytes per pixel row
    constexpr int A_STAGE = BKP * A_ROWB, B_STAGE = BKP * B_ROWB;
    constexpr int A_PR = 1024 / A_ROWB, B_PR = 1024 / B_ROWB;   // pixel rows per 1 KB DMA piece (4 or 8)
    constexpr int A_NP = A_STAGE / 1024 / NW, B_NP = B_STAGE / 1024 / NW;   // pieces per wave and k-tile (4, 2 or 1)
    constexpr int NPIECE = A_NP + B_NP;
    static_assert((NSTAGE - 2) * NPIECE < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [NSTAGE] A images, then [NSTAGE] B images
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem_raw;

    const int tid = threadIdx.x;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / p.tiles;
    const int tl = lin - split * p.tiles;
    const int tile_n = tl % p.tiles_n;
    const int tile_m = tl / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.npix, p_begin + p.chunk);
    const bool reflect = p.pad_mode == 1;
    const int wave_id = tid >> 6;
    const int lane = tid & 63;
    const bf16* const zero = reinterpret_cast<const bf16*>(sscg_zero_page16);

    // ---- copy side.  A piece = 1 KB of LDS = A_PR (B_PR) whole pixel rows; lane l writes chunk l % (ROWB / 16) of row l / (ROWB / 16),
    // i.e. it FETCHES source chunk (that chunk ^ wsw(row)).  Wave w owns pieces w * NP .. w * NP + NP - 1 of every k-tile.
    const int a_row = lane / (A_ROWB / 16);                  // pixel row inside a piece (pieces start at multiples of 4 / 8 pixels)
    const int a_q = (lane % (A_ROWB / 16)) ^ wsw<BM>(a_row);
    const bool a_colok = m0 + a_q * 8 < p.Kc;
    const bf16* const a_src = p.dy + (a_colok ? m0 + a_q * 8 : 0);
    const int b_row = lane / (B_ROWB / 16);
    const int b_q = (lane % (B_ROWB / 16)) ^ wsw<BN>(b_row);
    const int b_n = n0 + b_q * 8;
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
        b_src = p.x + c;
    }
    const bool plain = p.Ng == p.C && p.stride == 1 && p.pad == 0;      // 1x1, stride 1: source pixel = output pixel
    int dma_stage = 0;
    int f_pix = p_begin;                                      // first pixel of the next k-tile to request
    const int lds_wave_a = __builtin_amdgcn_readfirstlane(wave_id * A_NP * 1024);
    const int lds_wave_b = __builtin_amdgcn_readfirstlane(wave_id * B_NP * 1024);
    // Round 6: the dy rows - and the x rows of a 1x1 stride-1 filter - are plain pixel rows: copied by buffer loads, the lane's offset
    // inside a k-tile in ONE VGPR per copy (computed here), the k-tile's first pixel in the instruction's SGPR offset, a row past this
    // workgroup's last pixel or a masked column at offset 2 GB (out of range = zeros).  The 64-bit address form
    // below cost ~12 VALU operations per copy, 30-50 per k-tile of 8 MFMAs (7 VALU per MFMA measured: 8 % of config 3's VALU time).
    const bool buf_a = p.buf_ok != 0, buf_b = p.buf_ok != 0 && plain;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, (short)0, (int)((unsigned)p.npix * (unsigned)p.Kc * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((unsigned)p.npix * (unsigned)p.C * 2u), 0x00020000);
    unsigned a_vo[A_NP], b_vo[B_NP];
#pragma unroll
    for (int s = 0; s < A_NP; ++s)
        a_vo[s] = a_colok ? ((unsigned)((wave_id * A_NP + s) * A_PR + a_row) * (unsigned)p.Kc + (unsigned)(m0 + a_q * 8)) * 2u : 0x80000000u;
#pragma unroll
    for (int s = 0; s < B_NP; ++s)
        b_vo[s] = b_colok ? ((unsigned)((wave_id * B_NP + s) * B_PR + b_row) * (unsigned)p.C + (unsigned)b_n) * 2u : 0x80000000u;
    auto request_tile = [&]() {
        if (buf_a) {
            const int so = __builtin_amdgcn_readfirstlane(f_pix * p.Kc * 2);
#pragma unroll
            for (int s = 0; s < A_NP; ++s) {
                const bool pok = f_pix + (wave_id * A_NP + s) * A_PR + a_row < p_end;      // (the SGPR offset takes no part in the range check)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave_a + s * 1024),
                                                         16, (int)(pok ? a_vo[s] : 0x80000000u), so, 0, 0);
            }
        } else {
#pragma unroll
        for (int s = 0; s < A_NP; ++s) {
            const int pix = f_pix + (wave_id * A_NP + s) * A_PR + a_row;
            const bf16* g = (a_colok && pix < p_end) ? a_src + (size_t)pix * p.Kc : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds0 + dma_stage * A_STAGE + lds_wave_a + s * 1024), 16, 0, 0);
        }
        }
        if (buf_b) {
            const int so = __builtin_amdgcn_readfirstlane(f_pix * p.C * 2);
#pragma unroll
            for (int s = 0; s < B_NP; ++s) {
                const bool pok = f_pix + (wave_id * B_NP + s) * B_PR + b_row < p_end;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE + lds_wave_b + s * 1024),
                                                         16, (int)(pok ? b_vo[s] : 0x80000000u), so, 0, 0);
            }
        } else {
#pragma unroll
        for (int s = 0; s < B_NP; ++s) {
            const int pix = f_pix + (wave_id * B_NP + s) * B_PR + b_row;
            bool ok = b_colok && pix < p_end;
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
                if (reflect) {          // wave-uniform (ReflectionPad2d folded into the conv: the ResNet generators' stems / blocks)
                    sy = sy < 0 ? -sy : sy;
                    sx = sx < 0 ? -sx : sx;
                    sy = sy >= p.H ? 2 * (p.H - 1) - sy : sy;
                    sx = sx >= p.W ? 2 * (p.W - 1) - sx : sx;
                }
                ok = ok & ((unsigned)sy < (unsigned)p.H) & ((unsigned)sx < (unsigned)p.W);
                spix = (size_t)((img * p.H + sy) * p.W + sx);
            }
            const bf16* g = ok ? b_src + spix * p.C : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lds0 + NSTAGE * A_STAGE + dma_stage * B_STAGE + lds_wave_b + s * 1024), 16, 0, 0);
        }
        }
        dma_stage = dma_stage + 1 == NSTAGE ? 0 : dma_stage + 1;
        f_pix += BKP;
    };

    // ---- MFMA side
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int i16 = lane & 15;                // lane inside its 16-lane transpose group
    const int wm = wave_id / WN;
    const int wn = wave_id % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;
    // transpose-read address of operand rows (channels) cb .. cb + 15 handled by this lane's group, pixels 8 lh + (0..3) [+ 4 t]:
    // the lane supplies 4 consecutive channels (8 bytes) of pixel 8 lh + (i16 >> 2)
    const int tr_pix = 8 * lh + (i16 >> 2);
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = row_w + i * 32 + (li & 16) + 4 * (i16 & 3);
        a_off[i] = tr_pix * A_ROWB + (((ch >> 3) ^ wsw<BM>(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ch = col_w + j * 32 + (li & 16) + 4 * (i16 & 3);
        b_off[j] = NSTAGE * A_STAGE + tr_pix * B_ROWB + (((ch >> 3) ^ wsw<BN>(tr_pix)) << 4) + ((ch >> 2) & 1) * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nsteps = (p_end - p_begin + BKP - 1) / BKP;
    int issued = 0;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) {
        if (s < nsteps) { request_tile(); ++issued; }
    }
    if (nsteps >= NSTAGE - 1) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    int rd_stage = 0;
    for (int it = 0; it < nsteps; ++it) {
        const lds_char* a = lds0 + rd_stage * A_STAGE;
        const lds_char* b = lds0 + rd_stage * B_STAGE;
        bf16x4 fa[2][TM][2], fb[2][TN][2];
        // MFMA kk contracts pixels 16 kk .. 16 kk + 15: lane half lh feeds pixels 16 kk + 8 lh + (0..7) = two transpose-reads
        auto mfma_group = [&](int cur) {
            bf16x8 va[TM], vb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                va[i] = __builtin_shufflevector(fa[cur][i][0], fa[cur][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(va[i]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                vb[j] = __builtin_shufflevector(fb[cur][j][0], fb[cur][j][1], 0, 1, 2, 3, 4, 5, 6, 7);
                pin(vb[j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[i], vb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        wg_read_frags<0, TM, TN, A_ROWB, B_ROWB>(fa[0], fb[0], a, b, a_off, b_off);
        const bool more = issued < nsteps;
        if (more) { request_tile(); ++issued; }
        __builtin_amdgcn_sched_barrier(0);
        wg_read_frags<1, TM, TN, A_ROWB, B_ROWB>(fa[1], fb[1], a, b, a_off, b_off);
        wait_lgkm<2 * (TM + TN)>();
        mfma_group(0);
        wg_read_frags<2, TM, TN, A_ROWB, B_ROWB>(fa[0], fb[0], a, b, a_off, b_off);
        wait_lgkm<2 * (TM + TN)>();
        mfma_group(1);
        wg_read_frags<3, TM, TN, A_ROWB, B_ROWB>(fa[1], fb[1], a, b, a_off, b_off);
        wait_lgkm<2 * (TM + TN)>();
        mfma_group(0);
        wait_lgkm<0>();
        mfma_group(1);
        rd_stage = rd_stage + 1 == NSTAGE ? 0 : rd_stage + 1;
        if (more) wait_vm<(NSTAGE - 2) * NPIECE>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }

    float* out = p.out + (size_t)split * p.Kc * p.Ng;
    const bool direct = (p.splits == 1);
#ifndef K16_STAGE_WG
#define K16_STAGE_WG 1
#endif
#if K16_STAGE_WG
    // the result tile leaves through LDS as 16-byte row segments (common.h): Ng = R * S * C with C % 8 == 0
    sscg_stage_store_tile<BM, BN, NW * 64, TM, TN>(acc, reinterpret_cast<float*>(smem_raw), out, m0, n0, p.Kc, p.Ng, row_w, col_w, li, lh, tid,
                                                   direct ? p.beta : 0.f);
#else
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + col_w + j * 32 + li;
        if (n >= p.Ng) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < p.Kc) {
                    const size_t o = (size_t)m * p.Ng + n;
                    float v = acc[i][j][e];
                    if (direct && p.beta != 0.f) v += p.beta * out[o];
                    out[o] = v;
                }
            }
        }
    }
#endif
}

struct Wg16Plan { int cfg, bm, bn, splits, chunk; };

// Split of the pixel range.  The 128x128 kernel (8 waves, 64 KB of LDS) runs two workgroups per CU: the split is the largest that
// still fits ONE round of 512 workgroups (256-channel 3x3: 36 tiles x 14 = 504; one more split = 540 workgroups = a second, nearly
// empty round: 690 -> 560 TF/s); the 64x64 kernel holds four per CU.  At least 4 k-steps per workgroup.
Wg16Plan plan_wg16(const sscg_conv_desc* d) {
    Wg16Plan pl;
    const int Kc = d->K, Ng = d->R * d->S * d->C;
    const long npix = (long)d->N * d->P * d->Q;
    const long steps = cdiv(npix, BKP);
    pl.cfg = (Kc >= 128 && Ng >= 128 && (long)cdiv(Kc, 128) * cdiv(Ng, 128) >= 8) ? 0 : 1;
    pl.bm = pl.bn = pl.cfg == 0 ? 128 : 64;
    const long tiles = (long)cdiv(Kc, pl.bm) * cdiv(Ng, pl.bn);
    const int flags = (d->wgrad_tuning >> 24) & 0xff;      // kernel-variant switches (tools/conv16_bench.py): bits 4..7 = workgroup slots / 128
    const long slots = ((flags >> 4) & 15) ? 128L * ((flags >> 4) & 15) : (pl.cfg == 0 ? 512 : 1024);
    long s = slots / tiles;
    if (s > steps / 4) s = steps / 4;
    if (s > 1024) s = 1024;
    if (s < 1) s = 1;
    const long steps_per = cdiv(steps, s);
    pl.chunk = (int)(steps_per * BKP);
    pl.splits = cdiv(npix, pl.chunk);
    return pl;
}

template <int WM, int WN, int TM, int TN, int NSTAGE>
int launch_wg16t(Wg16Params p, int splits, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    p.tiles_n = cdiv(p.Ng, BN);
    p.tiles = cdiv(p.Kc, BM) * p.tiles_n;
    p.splits = splits;
    size_t smem = (size_t)NSTAGE * BKP * (BM + BN) * sizeof(bf16);
    if (smem < (size_t)BM * (BN + 4) * sizeof(float)) smem = (size_t)BM * (BN + 4) * sizeof(float);      // the staged result tile
    constexpr int NT = WM * WN * 64;
    auto kern = wgrad16t_kernel<WM, WN, TM, TN, NSTAGE>;
    SSCG_ENSURE_SMEM((kern), smem);
    hipLaunchKernelGGL(kern, dim3(p.tiles * splits), dim3(NT), smem, st, p);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

}  // namespace

bool sscg_wgrad16_applies(const sscg_conv_desc* d) {
    return d->x_dtype == SSCG_BF16 && d->y_dtype == SSCG_BF16 && d->K % 8 == 0 && d->C % 8 == 0 && d->K >= 32 &&
           (long)d->R * d->S * d->C >= 32;
}

size_t sscg_wgrad16_workspace(const sscg_conv_desc* d) {
    Wg16Plan pl = plan_wg16(d);
    return pl.splits > 1 ? (size_t)pl.splits * d->K * d->R * d->S * d->C * sizeof(float) : 0;
}

int sscg_wgrad16(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, size_t ws_bytes, hipStream_t st) {
    Wg16Plan pl = plan_wg16(d);
    const size_t need = sscg_wgrad16_workspace(d);
    if (need > 0 && (!ws || ws_bytes < need)) return SSCG_ERR_WORKSPACE;
    Wg16Params p = {};
    p.x = reinterpret_cast<const bf16*>(x); p.dy = reinterpret_cast<const bf16*>(dy);
    p.out = pl.splits > 1 ? reinterpret_cast<float*>(ws) : dw;
    p.Kc = d->K; p.Ng = d->R * d->S * d->C; p.C = d->C;
    p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.pad_mode = d->pad_mode;
    p.npix = d->N * d->P * d->Q; p.chunk = pl.chunk;
    p.buf_ok = ((size_t)p.npix * d->K * 2 < ((size_t)1 << 31) && (size_t)d->N * d->H * d->W * d->C * 2 < ((size_t)1 << 31)) ? 1 : 0;
    p.beta = pl.splits > 1 ? 0.f : beta;
    p.div_pq = make_fastdiv(d->P * d->Q);
    p.div_q = make_fastdiv(d->Q);
    int rc;
    const int flags = (d->wgrad_tuning >> 24) & 0xff;
    const int variant = (flags >> 1) & 7;                // tools/conv16_bench.py: 0 = default
    if (pl.cfg == 0) {
        // 128x128 tile as 8 waves of 64x32: 4 waves per SIMD with two workgroups per CU (550 -> 665 TF/s on the 256-ch 3x3)
        rc = variant == 1 ? launch_wg16t<2, 2, 2, 2, 2>(p, pl.splits, st) : launch_wg16t<2, 4, 2, 1, 2>(p, pl.splits, st);
    } else {
        rc = variant == 1 ? launch_wg16t<2, 2, 1, 1, 4>(p, pl.splits, st) : launch_wg16t<2, 2, 1, 1, 2>(p, pl.splits, st);
    }
    if (rc) return rc;
    if (pl.splits > 1) return sscg_wgrad_reduce(reinterpret_cast<const float*>(ws), dw, (size_t)d->K * p.Ng, pl.splits, beta, st);
    return SSCG_OK;
}
