// Implicit-GEMM convolution for NHWC fp32 tensors on CDNA4 (gfx950) matrix cores.
//
// One kernel family ("KC": both GEMM operands are K-contiguous in HBM) serves
//   * Conv2d forward            (reference call sites: arch/ops.py:43,49,68; arch/generators.py:325,331,336,373,388)
//   * Conv2d data-gradient      (autograd of the above; gather form, no atomics)
//   * ConvTranspose2d forward   (arch/ops.py:55-56) == data-gradient of the mirrored conv
//
// GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//   m  = output pixel (n_img, oy, ox)      n = output channel       k = (tap, source channel)
//   A is gathered on the fly from the NHWC source (zero / reflection padding, stride, dilation
//   are index arithmetic in the tile loader - no im2col buffer, no padded copy);
//   B is the weight tensor in its physical [Cout][kh][kw][Cin] order (K-contiguous rows).
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/issue/SIMD).  Each wave owns
// TM x TN accumulators of 32x32; a 256-thread workgroup is WM x WN waves.
// LDS image of both operands is [row][BK + 4] (row = m or n, k contiguous, +4 floats of pad so that
// ds_read_b128 fragment reads are bank-conflict free).  A lane (i = lane & 31, h = lane >> 5) reads
// four consecutive k of row i at k-offset h*4 with one ds_read_b128 and feeds them to four MFMAs:
// the k -> (instruction, half) assignment is a permutation applied identically to A and B, which
// a reduction does not care about.
//
// Loader: a thread always stages the same PA rows of A / PB rows of B and the same k column, so all
// row decoding happens once.  FAST path (source channels a multiple of BK, i.e. every >=64-channel
// layer): a k-tile never straddles a tap, so the gather address of a row changes only at tap
// boundaries (wave-uniform branch every Cs/BK tiles) and the per-tile loader work is a pointer bump.
#include "common.h"
#include "sscg_internal.h"
#include "reduce_common.h"

namespace {

#ifndef SSCG_BK
#define SSCG_BK 32
#endif
constexpr int BK = SSCG_BK;
constexpr int LDK = BK + 4;

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// XOR swizzle of the 16-byte slot inside a row of the LDS-DMA image: 16 consecutive rows must put one fragment slot into
// 16 different bank quads (128-byte rows: two rows share the 64 banks; 256-byte rows: one row spans them)
__device__ __forceinline__ int dma_swizzle(int row) { return BK == 32 ? ((row >> 1) & 7) : (row & (BK / 4 - 1)); }

struct KcParams {
    const float* __restrict__ src;   // A source (input for fwd, dy for dgrad)
    const float* __restrict__ wgt;   // [Ng][Ktot]
    const float* __restrict__ bias;  // [Ng] or null
    void* __restrict__ dst;          // [M][Ng] fp32 or bf16 (out_bf16)
    int out_bf16;
    int precision;                   // host side: 1 = bf16 contraction of the fp32 tiles (sscg_conv_desc.precision)
    int tuning;                      // host side: sscg_conv_desc.tuning
    int M, Ng, Ktot, Cs;
    int SH, SW;   // source spatial
    int OH, OW;   // destination spatial (row decode)
    int R, S;
    int stride, pad, dil;
    int pad_x;      // horizontal padding (== pad except for the parity classes of a strided data gradient)
    int wKtot;      // row stride of `wgt` (== Ktot unless the launch walks a sub-lattice of the taps)
    int wt_ky0, wt_kx0, wt_step, wt_S;   // tap (ty, tx) of this launch = tap (wt_ky0 + ty*wt_step, wt_kx0 + tx*wt_step) of the [R][wt_S] weight
    int o_step, o_a, o_b, o_W, o_HW;     // output row (img, i, j) -> pixel img*o_HW + (i*o_step + o_a)*o_W + j*o_step + o_b
    int pad_mode;
    int act;
    float slope;
    int tiles_n;
    int tiles;      // tiles_m * tiles_n
    int splits;     // split-K factor of the tiles >= full_tiles (partials to `part`, reduced by kc_reduce_kernel)
    int ksplit;     // k-tiles per split
    int full_tiles; // tiles [0, full_tiles) are computed whole by one workgroup each; the rest ("tail") are split
    int m_tail0;    // first output row of the tail tiles
    float* __restrict__ part;  // [splits][M - m_tail0][Ng] when splits > 1
    // fused normalisation statistics (see conv_bf16.hip): records [tile_m * WM + wave_row][2 groups][Ng][2] doubles, or null
    double* __restrict__ stats;
    int stat_L;
    double* __restrict__ xstats;     // host side: records of the split rows, written by the split reduction ([blocks][Ng][2])
};

__device__ __forceinline__ void store_out(void* dst, size_t idx, float v, int out_bf16) {
    if (out_bf16) reinterpret_cast<__bf16*>(dst)[idx] = (__bf16)v;
    else reinterpret_cast<float*>(dst)[idx] = v;
}

// 256 B of zeros: the source of LDS-DMA lanes that fall into padding / outside the tile
__device__ float sscg_zero_page[64];

template <int MODE, int WM, int WN, int TM, int TN, int VEC, bool FAST, int NBUF = 2, bool DMA = false, int BF16 = 0, bool N4 = false>
__global__ __launch_bounds__(256) void conv_kc_kernel(KcParams p) {
    // N4: at most four output columns (the 3-channel heads).  The 256 x 32 tile is staged exactly like any other, but the
    // contraction runs on v_mfma_f32_4x4x1_f32 - 16 independent 4x4 outer products per instruction = 64 rows x 4 columns x
    // 1 k: lane l feeds row l of the wave's 64 (a 16-byte fragment = 4 k = 4 instructions) and column l % 4 of the filter
    // tile.  1/8 of the matrix-core time of the 32-wide tile, which spends 29 of its 32 columns on padding.
    static_assert(!N4 || (WM == 4 && WN == 1 && TM == 2 && TN == 1 && !BF16 && NBUF == 2), "N4 tile = 4 waves x 64 rows, one column quad");
    // BF16: the operands are rounded to bfloat16 (RNE) when the fragments leave LDS and contracted by
    // v_mfma_f32_32x32x16_bf16 with fp32 accumulation - tensors in HBM and LDS stay fp32, so loader, swizzle and epilogue are
    // those of the fp32 kernel.  16x fewer MFMA cycles; the kernel is then bound by LDS reads and the global->LDS copies.
    static_assert(!BF16 || (NBUF == 2 && BK == 32), "the bf16 contraction regroups the four 8-wide k-groups of a 32-wide k-tile");
    static_assert(!DMA || (FAST && NBUF == 2), "LDS-DMA staging is built on the fast path");
    // DMA staging: `global_load_lds_dwordx4` writes lane-linear (wave-uniform base + lane*16 B), so the LDS image is
    // the unpadded [row][32]; bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of the
    // 16-B slot with (row >> 1) & 7, applied to the per-lane SOURCE address and to the read address alike.
    constexpr int LDR = DMA ? BK : LDK;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(!FAST || VEC == 4, "fast path is vectorised");
    // loader geometry
    constexpr int KQ = BK / VEC;          // threads along k
    constexpr int RPP = 256 / KQ;         // rows per pass
    constexpr int PA = BM / RPP;          // passes for A
    constexpr int PB = BN / RPP;          // passes for B
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);                 // [NBUF][BM][LDR]
    float* Bs = As + NBUF * BM * LDR;                               // [NBUF][BN][LDR]
    int* tapinfo = reinterpret_cast<int*>(Bs + NBUF * BN * LDR);       // [R*S]: (dy << 16) | dx
    int* wtapinfo = tapinfo + (p.R * p.S > 0 ? p.R * p.S : 1);         // [R*S]: tap index inside the weight row

    const int tid = threadIdx.x;
    // Whole tiles first (one workgroup each), then the tail tiles cut along K: the tail of a launch that does not
    // fill a whole number of rounds on the 256 CUs is spread over all of them instead of leaving most CUs idle.
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
    const int tile_n = tile % p.tiles_n;
    const int tile_m = tile / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    for (int t = tid; t < p.R * p.S; t += 256) {
        int ky = t / p.S;
        int kx = t - ky * p.S;
        tapinfo[t] = ((ky * p.dil) << 16) | (kx * p.dil);
        wtapinfo[t] = (p.wt_ky0 + ky * p.wt_step) * p.wt_S + p.wt_kx0 + kx * p.wt_step;
    }

    const int r0 = tid / KQ;
    // k slot this thread stages: identity, or (DMA) the slot whose data must land at LDS slot tid%8 of row r0
    const int kq = DMA ? ((tid % KQ) ^ dma_swizzle(r0)) : (tid % KQ);
    const int wave_id = tid >> 6;

    // ---- per-thread loader state (decoded once)
    const float* arow[PA];   // image base of the row's source
    int ay0[PA], ax0[PA];
    bool aok[PA];
#pragma unroll
    for (int ps = 0; ps < PA; ++ps) {
        const int m = m0 + r0 + ps * RPP;
        aok[ps] = m < p.M;
        const int mm = aok[ps] ? m : 0;
        const int img = mm / (p.OH * p.OW);
        const int rem = mm - img * (p.OH * p.OW);
        const int oy = rem / p.OW;
        const int ox = rem - oy * p.OW;
        arow[ps] = p.src + (size_t)img * p.SH * p.SW * p.Cs;
        if (MODE == MODE_FWD) {
            ay0[ps] = oy * p.stride - p.pad;
            ax0[ps] = ox * p.stride - p.pad_x;
        } else {
            ay0[ps] = oy + p.pad;
            ax0[ps] = ox + p.pad_x;
        }
    }
    const float* brow[PB];
    bool bok[PB];
#pragma unroll
    for (int ps = 0; ps < PB; ++ps) {
        const int n = n0 + r0 + ps * RPP;
        bok[ps] = n < p.Ng;
        brow[ps] = p.wgt + (size_t)(bok[ps] ? n : 0) * p.wKtot + kq * VEC;
    }
    __syncthreads();              // tapinfo visible

    const bool reflect = p.pad_mode == 1;

    // source pixel of row `ps` for tap offsets (tdy, tdx); returns validity
    auto locate = [&](int ps, int tdy, int tdx, int& pix) -> bool {
        bool ok = aok[ps];
        int sy, sx;
        if (MODE == MODE_FWD) {
            sy = ay0[ps] + tdy;
            sx = ax0[ps] + tdx;
            int ry = sy < 0 ? -sy : sy;
            int rx = sx < 0 ? -sx : sx;
            ry = ry >= p.SH ? 2 * (p.SH - 1) - ry : ry;
            rx = rx >= p.SW ? 2 * (p.SW - 1) - rx : rx;
            sy = reflect ? ry : sy;
            sx = reflect ? rx : sx;
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

    float ra[PA][VEC];
    float rb[PB][VEC];
    unsigned okmask = 0;   // validity of the staged loads; applied at LDS-store time so that the loaded registers are
                           // not touched (=> not waited for) until the MFMAs of the current tile have been issued

    const int nk_all = (p.Ktot + BK - 1) / BK;
    const int kt0 = partial ? split * p.ksplit : 0;         // this workgroup's k-tile range (split-K)
    const int kt1 = partial ? min(nk_all, kt0 + p.ksplit) : nk_all;
    // generic-path state: this thread's k column, advanced by BK per tile
    int lk = kt0 * BK + kq * VEC;
    // fast-path state
    const int f_nchunk = FAST ? p.Cs / BK : 1;
    int f_tap = FAST ? kt0 / f_nchunk : 0;
    int f_chunk = FAST ? kt0 - f_tap * f_nchunk : 0;
    int f_k = kt0 * BK;
    const float* aptr[PA];
    unsigned f_okbits = 0;
    int dma_buf = 0;          // LDS image the next DMA tile lands in

    auto fast_set_tap = [&](int tap) {
        const int ti = tapinfo[tap];
        const int tdy = ti >> 16, tdx = ti & 0xffff;
        f_k = wtapinfo[tap] * p.Cs;          // first channel of this tap inside the weight row
        f_okbits = 0;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            int pix;
            const bool ok = locate(ps, tdy, tdx, pix);
            f_okbits |= ok ? (1u << ps) : 0u;
            aptr[ps] = arow[ps] + (size_t)pix * p.Cs + kq * VEC;
        }
    };
    if (FAST) {
        fast_set_tap(f_tap < p.R * p.S ? f_tap : 0);
        f_k += f_chunk * BK;
    }

    auto load_tile = [&]() {
        if constexpr (FAST) {
            if (f_chunk == f_nchunk) {       // wave-uniform: next tap
                f_chunk = 0;
                ++f_tap;
                fast_set_tap(f_tap < p.R * p.S ? f_tap : 0);
            }
            okmask = f_okbits;
            const int coff = f_chunk * BK;
            if constexpr (DMA) {
                float* la = As + dma_buf * BM * LDR + wave_id * 256;
                float* lb = Bs + dma_buf * BN * LDR + wave_id * 256;
#pragma unroll
                for (int ps = 0; ps < PA; ++ps) {
                    const float* g = ((f_okbits >> ps) & 1u) ? aptr[ps] + coff : sscg_zero_page;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(la + ps * 1024), 16, 0, 0);
                }
#pragma unroll
                for (int ps = 0; ps < PB; ++ps) {
                    const float* g = bok[ps] ? brow[ps] + f_k : sscg_zero_page;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(lb + ps * 1024), 16, 0, 0);
                }
                dma_buf ^= 1;
            } else {
#pragma unroll
                for (int ps = 0; ps < PA; ++ps) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(aptr[ps] + coff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[ps][e] = v[e];
                }
#pragma unroll
                for (int ps = 0; ps < PB; ++ps) {
                    okmask |= bok[ps] ? (1u << (16 + ps)) : 0u;
                    f32x4 v = *reinterpret_cast<const f32x4*>(brow[ps] + f_k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = v[e];
                }
            }
            ++f_chunk;
            f_k += BK;
        } else {
            okmask = 0;
            const bool kvalid = lk < p.Ktot;
            const int tap = kvalid ? lk / p.Cs : 0;
            const int ci = kvalid ? lk - tap * p.Cs : 0;
            const int ti = tapinfo[tap];
            const int tdy = ti >> 16, tdx = ti & 0xffff;
#pragma unroll
            for (int ps = 0; ps < PA; ++ps) {
                int pix;
                const bool ok = locate(ps, tdy, tdx, pix) && kvalid;
                okmask |= ok ? (1u << ps) : 0u;
                const float* g = arow[ps] + ((size_t)(ok ? pix : 0) * p.Cs + ci);
                if constexpr (VEC == 4) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[ps][e] = v[e];
                } else {
                    ra[ps][0] = *g;
                }
            }
#pragma unroll
            for (int ps = 0; ps < PB; ++ps) {
                const bool ok = kvalid && bok[ps];
                okmask |= ok ? (1u << (16 + ps)) : 0u;
                // masked lanes read the tensor's first element(s): `brow` already carries +kq*VEC, which would run past
                // the end of a weight that is shorter than one k-tile (e.g. the 128->1 head: Ktot = 1)
                const float* g = ok ? brow[ps] + (lk - kq * VEC) : p.wgt;
                if constexpr (VEC == 4) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[ps][e] = v[e];
                } else {
                    rb[ps][0] = *g;
                }
            }
            lk += BK;
        }
    };

    auto store_tile = [&](int buf) {
        if constexpr (DMA) return;   // the DMA already put the tile into LDS
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            float* d = a + (r0 + ps * RPP) * LDK + kq * VEC;
            const bool ok = (okmask >> ps) & 1u;
            if constexpr (VEC == 4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? ra[ps][e] : 0.f;
                *reinterpret_cast<f32x4*>(d) = v;
            } else {
                *d = ok ? ra[ps][0] : 0.f;
            }
        }
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) {
            float* d = b + (r0 + ps * RPP) * LDK + kq * VEC;
            const bool ok = (okmask >> (16 + ps)) & 1u;
            if constexpr (VEC == 4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? rb[ps][e] : 0.f;
                *reinterpret_cast<f32x4*>(d) = v;
            } else {
                *d = ok ? rb[ps][0] : 0.f;
            }
        }
    };

    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int row_w = wm * TM * 32;
    const int col_w = wn * TN * 32;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};    // N4: rows 4 * (lane / 4) + r of the wave's 64, column lane % 4
    const int nk = kt1 - kt0;
    if (nk > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = NBUF == 2 ? (kt & 1) : 0;
        const float* a = As + buf * BM * LDR + (row_w + li) * LDR + (DMA ? 0 : lh * 4);
        const float* b = Bs + buf * BN * LDR + (col_w + li) * LDR + (DMA ? 0 : lh * 4);
        const int swz = dma_swizzle(li);  // DMA image: 16-B slot c of a row lives at slot c ^ swz
        auto koff = [&](int kk) { return DMA ? (((kk * 2 + lh) ^ swz) * 4) : kk * 8; };
        if constexpr (N4) {
            const int arow_l = row_w + lane;                          // this lane's row of the A image
            const float* a4p = As + buf * BM * LDR + arow_l * LDR;
            const float* b4p = Bs + buf * BN * LDR + (lane & 3) * LDR;
            const int sa = DMA ? dma_swizzle(arow_l) : 0, sb = DMA ? dma_swizzle(lane & 3) : 0;
            f32x4 qa[BK / 4], qb[BK / 4];
#pragma unroll
            for (int c = 0; c < BK / 4; ++c) {
                qa[c] = *reinterpret_cast<const f32x4*>(a4p + ((c ^ sa) * 4));
                qb[c] = *reinterpret_cast<const f32x4*>(b4p + ((c ^ sb) * 4));
            }
            if (kt + 1 < nk) load_tile();
#pragma unroll
            for (int c = 0; c < BK / 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(qa[c][e], qb[c][e], acc4, 0, 0, 0);
        } else if constexpr (BF16) {
            f32x4 ga[4][TM], gb[4][TN];           // the four 8-wide k-groups of this k-tile
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) ga[kk][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDR + koff(kk));
#pragma unroll
                for (int j = 0; j < TN; ++j) gb[kk][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDR + koff(kk));
            }
            if (kt + 1 < nk) load_tile();
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {      // two MFMAs of k = 16: lane half h contributes the k-groups (2*g2, 2*g2+1), slot h
                bf16x8 pa[TM], pb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pa[i][e] = (__bf16)ga[2 * g2][i][e]; pa[i][4 + e] = (__bf16)ga[2 * g2 + 1][i][e]; }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pb[j][e] = (__bf16)gb[2 * g2][j][e]; pb[j][4 + e] = (__bf16)gb[2 * g2 + 1][j][e]; }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i], pb[j], acc[i][j], 0, 0, 0);
            }
        } else {
        f32x4 fa[2][TM], fb[2][TN];
        // fragments of the first k-group are requested right after the barrier ...
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDR + koff(0));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDR + koff(0));
        // ... and the global loads of tile kt+1 are issued under their latency; they land under the MFMAs
        if (kt + 1 < nk) load_tile();
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDR + koff(kk + 1));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDR + koff(kk + 1));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][t], fb[cur][j][t], acc[i][j], 0, 0, 0);
        }
        }
        if (NBUF == 1) __syncthreads();      // single LDS image: every wave is done reading before it is overwritten
        if (kt + 1 < nk) store_tile(NBUF == 2 ? (buf ^ 1) : 0);
        // keep every MFMA of this k-tile in front of the barrier's `s_waitcnt vmcnt(0)`: the copies of the next tile land under
        // them (left alone, the scheduler sinks part of the MFMA block behind the wait)
        if constexpr (DMA) __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    if constexpr (N4) {
        // D layout of v_mfma_f32_4x4x1: lane = 4 * block + column, register = row of the block
        const int n = n0 + (lane & 3);
        if (n < p.Ng) {
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + row_w + (lane & ~3) + r;
                if (m >= p.M) continue;
                if (partial) {
                    p.part[((size_t)split * (p.M - p.m_tail0) + (m - p.m_tail0)) * p.Ng + n] = acc4[r];
                } else {
                    size_t row = (size_t)m;
                    if (p.o_step != 1) {
                        const int img = m / (p.OH * p.OW);
                        const int rem = m - img * (p.OH * p.OW);
                        const int oi = rem / p.OW;
                        const int oj = rem - oi * p.OW;
                        row = (size_t)img * p.o_HW + (size_t)(oi * p.o_step + p.o_a) * p.o_W + oj * p.o_step + p.o_b;
                    }
                    store_out(p.dst, row * p.Ng + n, sscg_act(acc4[r] + bv, p.act, p.slope), p.out_bf16);
                }
            }
        }
        return;
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    // Fused statistics of the normalisation layer that follows (arch/ops.py:40-57; arch/generators.py:345-365): fp64 column
    // sums of y and y^2 over this tile's rows; rows m >= gb belong to the next normalisation group (a tile straddles at most
    // one group boundary: stat_L >= BM).
    const bool want_stats = p.stats != nullptr && !partial;
    int gb = 0x7fffffff;
    if (want_stats) gb = (m0 / p.stat_L + 1) * p.stat_L;
    // Nearly every tile lies inside one group and inside the tensor: its sums are taken four rows at a time in fp32 (a lane's
    // registers 4g .. 4g+3 are four consecutive rows) and only the 4-row sums go to fp64 - a quarter of the fp64 work of the
    // element-wise walk, which cost the short-k 1x1 convs 8 % (relative error of a 4-term fp32 sum: 1e-7, not accumulating).
    // fp32 tensors take the 4-row shortcut only with bf16-rounded contractions (precision 1: the stems of the bf16 networks);
    // the exact-fp32 path keeps every element in fp64: a 1e-7 change of a BatchNorm statistic is one more sample of DeepLab's
    // chaotic fp32 trajectory, and the chained-loss parity test (4 x the reference's own fp32 noise) then sits on its edge.
    const bool slow_stats = want_stats && (p.precision != 1 || m0 + BM > gb || m0 + BM > p.M);
    const bool fast_stats = want_stats && !slow_stats;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + col_w + j * 32 + li;
        const bool nok = n < p.Ng;
        const float bv = (p.bias && nok) ? p.bias[n] : 0.f;
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
                        if (p.o_step != 1) {   // parity class of a strided data gradient: rows interleave into dx
                            const int img = m / (p.OH * p.OW);
                            const int rem = m - img * (p.OH * p.OW);
                            const int oi = rem / p.OW;
                            const int oj = rem - oi * p.OW;
                            row = (size_t)img * p.o_HW + (size_t)(oi * p.o_step + p.o_a) * p.o_W + oj * p.o_step + p.o_b;
                        }
                        store_out(p.dst, row * p.Ng + n, sscg_act(pre, p.act, p.slope), p.out_bf16);
                    }
                }
            }
        }
        if (want_stats) {
            s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);      // the lane halves hold different rows of a column
            s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
            if (lh == 0 && nok) {
                double* rec = p.stats + ((size_t)(tile_m * WM + wm) * 2) * p.Ng * 2;
                rec[(size_t)n * 2] = s0; rec[(size_t)n * 2 + 1] = q0;
                rec[((size_t)p.Ng + n) * 2] = s1; rec[((size_t)p.Ng + n) * 2 + 1] = q1;
            }
        }
    }
}

// y[i] = act(sum_s part[s][i] + bias[i % Ng])   (fixed order => deterministic); V floats per thread
template <int V>
__global__ __launch_bounds__(256) void kc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                         void* __restrict__ y, int out_bf16, size_t n, int Ng, int splits, int act, float slope) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= n) return;
    typedef float vec_t __attribute__((ext_vector_type(V)));
    vec_t s = 0.f;
#pragma unroll 8
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const vec_t*>(part + (size_t)k * n + i);
    const int c = (int)(i % Ng);
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        float v = s[e];
        if (bias) v += bias[c + e];     // V == 4 only when Ng % 4 == 0: the V channels are consecutive
        o[e] = sscg_act(v, act, slope);
    }
    if (out_bf16) {
        __bf16* yb = reinterpret_cast<__bf16*>(y) + i;
        if constexpr (V == 4) st4<__bf16>(yb, o); else st1<__bf16>(yb, o[0]);
    } else {
        float* yf = reinterpret_cast<float*>(y) + i;
        if constexpr (V == 4) st4<float>(yf, o); else st1<float>(yf, o[0]);
    }
}

// ---- host-side plan: tile configuration + split-K of the tail
// Tile choice (measured on MI355X, tools/conv_bench.py): 128x128 only when it still yields >= 2 workgroups per CU on a
// long reduction; otherwise 64x64; a 128x32 tile for heads with a handful of output channels.  Shapes whose source
// channel count is a multiple of BK stage through LDS-DMA (cfg 6/7).
static const int KC_BM[9] = {128, 128, 64, 64, 128, 128, 64, 128, 256};
static const int KC_BN[9] = {128, 64, 128, 64, 32, 128, 64, 128, 32};
// `tuning` = sscg_conv_desc.tuning (include/sscg.h): bits 0..7 = 1 + forced tile class, bits 8..15 = forced split-K
static int kc_choose_cfg(int M, int Ng, int Ktot, int Cs, int tuning) {
    const int forced = (tuning & 0xff) - 1;
    if (forced >= 0 && forced <= 8) return forced;
    const bool fast = Cs % BK == 0;
    if (Ng <= 4 && fast) return 8;
    if (Ng <= 32) return 4;
    const long w128 = (long)cdiv(M, 128) * cdiv(Ng, 128);
    if (w128 >= 512 && Ktot >= 1024 && Ng > 64) return fast ? 7 : 0;   // long reductions amortise the big tile's prologue (and N fills it)
    return fast ? 6 : 3;
}

struct KcSplit { int splits, ksplit, full_tiles, m_tail0; };

// Which tiles are cut along K, and how often.
//  * few-channel heads (Ng <= 32) on few rows: every tile (one 128x32 tile column gives only M/128 workgroups);
//  * 64x64-tile launches: only the TAIL - the tiles beyond the last whole round of 256 workgroups.  The DeepLab
//    stride-8 maps give 8712 rows -> 548 tiles: 512 whole tiles (2 per CU) + 36 tail tiles cut in 7, so every CU gets
//    2 1/7 tiles of work instead of 2 or 3 (71 % balance), and only 6.5 % of the output goes through partial sums.
static KcSplit plan_kc_split_raw(int M, int Ng, int Ktot, int Cs, int tuning);

// stat_L > 0: the launch also produces normalisation statistics.  Split tiles write partial sums, not results, so their
// rows are summed separately (one extra group of records): they must all lie in ONE normalisation group.
static KcSplit plan_kc_split(int M, int Ng, int Ktot, int Cs, int tuning, long stat_L = 0) {
    KcSplit r = plan_kc_split_raw(M, Ng, Ktot, Cs, tuning);
    if (stat_L > 0 && r.splits > 1 && (r.full_tiles == 0 || r.m_tail0 / stat_L != (M - 1) / stat_L)) {
        const int cfg = kc_choose_cfg(M, Ng, Ktot, Cs, tuning);
        r.splits = 1; r.ksplit = (Ktot + BK - 1) / BK;
        r.full_tiles = cdiv(M, KC_BM[cfg]) * cdiv(Ng, KC_BN[cfg]); r.m_tail0 = M;
    }
    return r;
}

static KcSplit plan_kc_split_raw(int M, int Ng, int Ktot, int Cs, int tuning) {
    const int nk = (Ktot + BK - 1) / BK;
    const int cfg = kc_choose_cfg(M, Ng, Ktot, Cs, tuning);
    const int force_split = (tuning >> 8) & 0xff;
    const int bm = KC_BM[cfg], bn = KC_BN[cfg];
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(Ng, bn);
    const int tiles = tiles_m * tiles_n;
    KcSplit r = {1, nk, tiles, M};
    if (force_split == 1) return r;   // tuning: never split
    if (force_split > 1) {            // tuning: split every tile
        r.ksplit = cdiv(nk, force_split);
        r.splits = cdiv(nk, r.ksplit);
        r.full_tiles = 0; r.m_tail0 = 0;
        return r;
    }
    if (Ng <= 32) {
        if (tiles >= 256 || nk < 32) return r;
        int s = cdiv(512, tiles);
        if (s > nk / 8) s = nk / 8;
        if (s > 32) s = 32;
        if (s < 2) return r;
        r.ksplit = cdiv(nk, s);
        r.splits = cdiv(nk, r.ksplit);
        r.full_tiles = 0; r.m_tail0 = 0;
        return r;
    }
    if (bm != 64 || bn != 64 || nk < 8 || tiles > 2300) return r;
    const int q = tiles / 256;
    const int full_m = (q * 256) / tiles_n;          // whole tile rows handled unsplit
    const int tail = tiles - full_m * tiles_n;
    if (tail <= 0 || tail > 208) return r;             // an almost complete round is left alone
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

static size_t kc_split_bytes(const KcSplit& sp, int M, int Ng) {
    return sp.splits > 1 ? (size_t)sp.splits * (M - sp.m_tail0) * Ng * sizeof(float) : 0;
}

template <int MODE, int WM, int WN, int TM, int TN, int VEC, bool FAST, int NBUF = 2, bool DMA = false, int BF16 = 0, bool N4 = false>
int launch_kc(const KcParams& p0, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    KcParams p = p0;
    p.tiles_n = cdiv(p.Ng, BN);
    int tiles_m = cdiv(p.M, BM);
    p.tiles = tiles_m * p.tiles_n;
    constexpr int LDR = DMA ? BK : LDK;
    size_t smem = (size_t)(NBUF * BM * LDR + NBUF * BN * LDR) * sizeof(float) + (size_t)(p.R * p.S > 0 ? p.R * p.S : 1) * 8;
    auto kern = conv_kc_kernel<MODE, WM, WN, TM, TN, VEC, FAST, NBUF, DMA, BF16, N4>;
    SSCG_ENSURE_SMEM((kern), smem);
    if (p.splits <= 1) { p.full_tiles = p.tiles; p.m_tail0 = p.M; }
    const int grid = p.full_tiles + (p.tiles - p.full_tiles) * p.splits;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, p);
    SSCG_LAUNCH_CHECK();
    if (p.splits > 1) {
        size_t n = (size_t)(p.M - p.m_tail0) * p.Ng;
        void* yt = reinterpret_cast<char*>(p.dst) + (size_t)p.m_tail0 * p.Ng * (p.out_bf16 ? 2 : 4);
        if (p.xstats)    // the split rows' column statistics come out of their reduction pass
            return launch_split_reduce_stats(p.part, p.bias, yt, p.out_bf16, p.M - p.m_tail0, p.Ng, p.splits, p.act, p.slope, p.xstats, st);
        if (p.Ng % 4 == 0 && (((size_t)yt | (size_t)p.part) & 15) == 0)
            hipLaunchKernelGGL(kc_reduce_kernel<4>, dim3(cdiv((long)(n / 4), 256)), dim3(256), 0, st, p.part, p.bias, yt, p.out_bf16, n,
                               p.Ng, p.splits, p.act, p.slope);
        else
            hipLaunchKernelGGL(kc_reduce_kernel<1>, dim3(cdiv((long)n, 256)), dim3(256), 0, st, p.part, p.bias, yt, p.out_bf16, n,
                               p.Ng, p.splits, p.act, p.slope);
        SSCG_LAUNCH_CHECK();
    }
    return SSCG_OK;
}

template <int MODE, int VEC, bool FAST>
int dispatch_kc(const KcParams& p, hipStream_t st) {
    switch (kc_choose_cfg(p.M, p.Ng, p.Ktot, p.Cs, p.tuning)) {
        case 0: if (p.precision == 1) return launch_kc<MODE, 2, 2, 2, 2, VEC, FAST, 2, false, true>(p, st);
                return launch_kc<MODE, 2, 2, 2, 2, VEC, FAST>(p, st);
        case 1: if (p.precision == 1) return launch_kc<MODE, 2, 2, 2, 1, VEC, FAST, 2, false, true>(p, st);
                return launch_kc<MODE, 2, 2, 2, 1, VEC, FAST>(p, st);
        case 2: if (p.precision == 1) return launch_kc<MODE, 2, 2, 1, 2, VEC, FAST, 2, false, true>(p, st);
                return launch_kc<MODE, 2, 2, 1, 2, VEC, FAST>(p, st);
        case 3: if (p.precision == 1) return launch_kc<MODE, 2, 2, 1, 1, VEC, FAST, 2, false, true>(p, st);
                return launch_kc<MODE, 2, 2, 1, 1, VEC, FAST>(p, st);
        case 4: if (p.precision == 1) return launch_kc<MODE, 4, 1, 1, 1, VEC, FAST, 2, false, true>(p, st);
                return launch_kc<MODE, 4, 1, 1, 1, VEC, FAST>(p, st);
        case 5: return launch_kc<MODE, 2, 2, 2, 2, VEC, FAST, 1>(p, st);   // 128x128, single LDS image (experimental)
        case 6: if constexpr (FAST) { if (p.precision == 1) return launch_kc<MODE, 2, 2, 1, 1, VEC, FAST, 2, true, 1>(p, st); return launch_kc<MODE, 2, 2, 1, 1, VEC, FAST, 2, true>(p, st); } else return SSCG_ERR_UNSUPPORTED;   // 64x64, LDS-DMA staging
        case 7: if constexpr (FAST) { if (p.precision == 1) return launch_kc<MODE, 2, 2, 2, 2, VEC, FAST, 2, true, 1>(p, st); return launch_kc<MODE, 2, 2, 2, 2, VEC, FAST, 2, true>(p, st); } else return SSCG_ERR_UNSUPPORTED;   // 128x128, LDS-DMA staging
        case 8: if constexpr (FAST) return launch_kc<MODE, 4, 1, 2, 1, VEC, FAST, 2, true, false, true>(p, st); else return SSCG_ERR_UNSUPPORTED;   // 256 x (<= 4): 4x4x1 MFMA
        default: return SSCG_ERR_BAD_ARG;
    }
}

template <int MODE>
int dispatch_mode(const KcParams& p, hipStream_t st) {
    if (p.Cs % BK == 0) return dispatch_kc<MODE, 4, true>(p, st);
    if (p.Cs % 4 == 0) return dispatch_kc<MODE, 4, false>(p, st);
    return dispatch_kc<MODE, 1, false>(p, st);
}

}  // namespace

// launch walks every tap of a dense [R][S] weight and writes rows in order
static void kc_dense_taps(KcParams& p) {
    p.pad_x = p.pad; p.wKtot = p.Ktot;
    p.wt_ky0 = 0; p.wt_kx0 = 0; p.wt_step = 1; p.wt_S = p.S;
    p.o_step = 1; p.o_a = 0; p.o_b = 0; p.o_W = 0; p.o_HW = 0;
}

// stride-2 data gradients are decomposed into parity classes when the vectorised tap walk applies (K % BK == 0)
static bool dgrad_by_parity(const sscg_conv_desc* d) {
    return (d->tuning & 0xff) == 0 && d->stride == 2 && d->dil == 1 && d->pad_mode == 0 && d->K % BK == 0;
}

static bool dt_ok(int dt) { return dt == SSCG_F32 || dt == SSCG_BF16; }
static bool wdt_ok(int dt) { return dt_ok(dt) || dt == SSCG_BF16X3; }

static int check_desc(const sscg_conv_desc* d) {
    if (!d) return SSCG_ERR_BAD_ARG;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->R <= 0 || d->S <= 0) return SSCG_ERR_BAD_ARG;
    if (d->stride <= 0 || d->dil <= 0 || d->pad < 0) return SSCG_ERR_BAD_ARG;
    if (!dt_ok(d->x_dtype) || !wdt_ok(d->w_dtype) || !dt_ok(d->y_dtype) || (d->precision < 0 || d->precision > 2)) return SSCG_ERR_BAD_ARG;
    int P = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
    int Q = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    if (P != d->P || Q != d->Q) return SSCG_ERR_BAD_ARG;
    if (d->pad_mode == 1 && (d->pad >= d->H || d->pad >= d->W)) return SSCG_ERR_BAD_ARG;
    // per-image extents are indexed with 32-bit arithmetic
    if ((long)d->H * d->W * (long)d->C >= (1L << 31)) return SSCG_ERR_UNSUPPORTED;
    if ((long)d->P * d->Q * (long)d->K >= (1L << 31)) return SSCG_ERR_UNSUPPORTED;
    if ((long)d->N * d->H * d->W >= (1L << 31) || (long)d->N * d->P * d->Q >= (1L << 31)) return SSCG_ERR_UNSUPPORTED;
    return SSCG_OK;
}

extern "C" size_t sscg_conv2d_fwd_workspace(const sscg_conv_desc* d) {
    if (!d) return 0;
    if (sscg_conv16_fwd_applies(d)) return sscg_conv16_fwd_workspace(d, 0);
    if (sscg_convs_fwd_applies(d)) return sscg_convs_fwd_workspace(d, 0);
    return kc_split_bytes(plan_kc_split(d->N * d->P * d->Q, d->K, d->R * d->S * d->C, d->C, d->tuning), d->N * d->P * d->Q, d->K);
}

extern "C" size_t sscg_conv2d_dgrad_workspace(const sscg_conv_desc* d) {
    if (!d) return 0;
    if (sscg_conv16_dgrad_applies(d)) return sscg_conv16_dgrad_workspace(d);
    if (sscg_convs_dgrad_applies(d)) return sscg_convs_dgrad_workspace(d);
    if (dgrad_by_parity(d)) return 0;
    return kc_split_bytes(plan_kc_split(d->N * d->H * d->W, d->C, d->R * d->S * d->K, d->K, d->tuning), d->N * d->H * d->W, d->C);
}

// ---- fused normalisation statistics: layout of the `stats` buffer = [tiles_m * WM records][2][K][2] doubles, then the
// records of the split-K rows [xrec][K][2]
struct StatPlan { int tiles_m, bm, wm, valid_tiles, xrec, xgroup; long m_tail0; size_t main_bytes, bytes; };

static bool fwd_stats_plan(const sscg_conv_desc* d, int G, long L, StatPlan* sp) {
    const long M = (long)d->N * d->P * d->Q;
    if (G <= 0 || L <= 0 || (long)G * L != M || d->act != SSCG_ACT_NONE || d->K <= 32) return false;   // (thin 1x1 shapes keep the matrix-core path when statistics are asked for)
    int splits;
    if (sscg_conv16_fwd_applies(d) || sscg_convs_fwd_applies(d)) {
        int full_tiles, m_tail0, tiles_n;
        if (sscg_conv16_fwd_applies(d)) {
            if (!sscg_conv16_stats_geometry(d, L, &sp->bm, &sp->wm, &tiles_n, &splits, &full_tiles, &m_tail0)) return false;
        } else if (!sscg_convs_stats_geometry(d, L, &sp->bm, &sp->wm, &tiles_n, &splits, &full_tiles, &m_tail0)) {
            return false;
        }
        sp->tiles_m = cdiv(M, sp->bm);
        sp->valid_tiles = splits > 1 ? full_tiles / tiles_n : sp->tiles_m;
        sp->m_tail0 = splits > 1 ? m_tail0 : M;
    } else {
        if (d->x_dtype != SSCG_F32 || d->w_dtype != SSCG_F32) return false;
        const int Ktot = d->R * d->S * d->C;
        const int cfg = kc_choose_cfg((int)M, d->K, Ktot, d->C, d->tuning);
        if (cfg == 8) return false;
        sp->bm = KC_BM[cfg];
        sp->wm = (cfg == 4) ? 4 : 2;
        if (L < sp->bm) return false;
        KcSplit ks = plan_kc_split((int)M, d->K, Ktot, d->C, d->tuning, L);
        sp->tiles_m = cdiv(M, sp->bm);
        const int tiles_n = cdiv(d->K, KC_BN[cfg]);
        splits = ks.splits;
        sp->valid_tiles = splits > 1 ? ks.full_tiles / tiles_n : sp->tiles_m;
        sp->m_tail0 = splits > 1 ? ks.m_tail0 : M;
    }
    sp->xrec = sp->m_tail0 < M ? split_stats_records(M - sp->m_tail0, d->K) : 0;
    sp->xgroup = sp->m_tail0 < M ? (int)(sp->m_tail0 / L) : -1;
    sp->main_bytes = (size_t)sp->tiles_m * sp->wm * 2 * d->K * 2 * sizeof(double);
    sp->bytes = sp->main_bytes + (size_t)sp->xrec * d->K * 2 * sizeof(double);
    return true;
}

extern "C" size_t sscg_conv2d_fwd_stats_bytes(const sscg_conv_desc* d, int G, int64_t L) {
    StatPlan sp;
    if (!d || check_desc(d) != SSCG_OK || !fwd_stats_plan(d, G, (long)L, &sp)) return 0;
    return sp.bytes;
}

extern "C" size_t sscg_conv2d_fwd_stats_workspace(const sscg_conv_desc* d) {
    if (!d) return 0;
    // the split plan of a statistics launch is a subset of the plain launch's
    return sscg_conv2d_fwd_workspace(d);
}

static int conv_fwd_impl(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, double* stats, long stat_L,
                         double* xstats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return SSCG_ERR_BAD_ARG;
    if (!stats && sscg_thin1x1_fwd_applies(d)) return sscg_thin1x1_fwd(d, x, w, bias, y, (hipStream_t)stream);
    if (sscg_conv16_fwd_applies(d)) return sscg_conv16_fwd(d, x, w, bias, y, stats, stat_L, xstats, ws, ws_bytes, (hipStream_t)stream);
    if (sscg_convs_fwd_applies(d)) return sscg_convs_fwd(d, x, w, bias, y, stats, stat_L, xstats, ws, ws_bytes, (hipStream_t)stream);
    if (d->x_dtype != SSCG_F32 || d->w_dtype != SSCG_F32) return SSCG_ERR_UNSUPPORTED;
    KcParams p = {};
    p.src = reinterpret_cast<const float*>(x); p.wgt = reinterpret_cast<const float*>(w); p.bias = bias; p.dst = y;
    p.out_bf16 = d->y_dtype == SSCG_BF16; p.precision = d->precision; p.tuning = d->tuning;
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Cs = d->C; p.Ktot = d->R * d->S * d->C;
    p.SH = d->H; p.SW = d->W; p.OH = d->P; p.OW = d->Q;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = d->pad_mode; p.act = d->act; p.slope = d->slope; p.tiles_n = 0; p.tiles = 0;
    p.stats = stats; p.stat_L = (int)stat_L; p.xstats = xstats;
    kc_dense_taps(p);
    KcSplit sp = plan_kc_split(p.M, p.Ng, p.Ktot, p.Cs, p.tuning, stats ? stat_L : 0);
    if (sp.splits > 1 && (!ws || ws_bytes < kc_split_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch_mode<MODE_FWD>(p, (hipStream_t)stream);
}

extern "C" int sscg_conv2d_fwd(const sscg_conv_desc* d, const void* x, const void* w, const float* bias,
                               void* y, void* ws, size_t ws_bytes, void* stream) {
    return conv_fwd_impl(d, x, w, bias, y, nullptr, 0, nullptr, ws, ws_bytes, stream);
}

extern "C" int sscg_conv2d_fwd_stats(const sscg_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int G,
                                     int64_t L, void* stats, size_t stats_bytes, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    StatPlan sp;
    if (!fwd_stats_plan(d, G, (long)L, &sp)) return SSCG_ERR_UNSUPPORTED;
    if (!stats || stats_bytes < sp.bytes) return SSCG_ERR_WORKSPACE;
    // rows that went through split-K: their statistics are produced by the split reduction itself (records after the main ones)
    double* xr = sp.xrec > 0 ? reinterpret_cast<double*>(reinterpret_cast<char*>(stats) + sp.main_bytes) : nullptr;
    return conv_fwd_impl(d, x, w, bias, y, reinterpret_cast<double*>(stats), (long)L, xr, ws, ws_bytes, stream);
}

// PixelDiscriminator's front half in one launch (arch/discriminators.py:70-73): y = conv2(lrelu(conv1(xf))) [+ conv2's bias] and - with
// G > 0 - the statistics of the normalisation layer behind it, records as sscg_conv2d_fwd_stats leaves them for `d` (the descriptor
// of conv2: sscg_norm_stats_from_conv(d, ...) finalises them).  The 64-channel map between the two convs is written only when the caller
// passes `h1` ([N*H*W][64] fp32: a backward pass that wants it stored), and never read back by this launch.
extern "C" int sscg_conv2d_front_applies(const sscg_conv_desc* d, int cin) {
    return d && check_desc(d) == SSCG_OK && sscg_convs_front_applies(d, cin) ? 1 : 0;
}

extern "C" int sscg_conv2d_front_fwd(const sscg_conv_desc* d, const void* xf, int cin, const float* w1, const float* b1, float slope1,
                                     void* h1, const void* w, const float* bias, void* y, int G, int64_t L, void* stats,
                                     size_t stats_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!xf || !w1 || !w || !y) return SSCG_ERR_BAD_ARG;
    if (!sscg_convs_front_applies(d, cin)) return SSCG_ERR_UNSUPPORTED;
    double* st = nullptr;
    if (G > 0) {
        StatPlan sp;
        if (!fwd_stats_plan(d, G, (long)L, &sp) || sp.xrec > 0) return SSCG_ERR_UNSUPPORTED;
        if (!stats || stats_bytes < sp.bytes) return SSCG_ERR_WORKSPACE;
        st = reinterpret_cast<double*>(stats);
    }
    return sscg_convs_fwd_front(d, xf, cin, w1, b1, slope1, h1, w, bias, y, st, (long)L, (hipStream_t)stream);
}

extern "C" int sscg_norm_stats_from_conv(const sscg_conv_desc* d, const void* stats, int G, int64_t L, float eps, float* mean,
                                         float* rstd, float* running_mean, float* running_var, float momentum, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    StatPlan sp;
    if (!stats || !mean || !rstd || !fwd_stats_plan(d, G, (long)L, &sp)) return SSCG_ERR_BAD_ARG;
    if ((running_mean != nullptr) != (running_var != nullptr)) return SSCG_ERR_BAD_ARG;
    const double* main = reinterpret_cast<const double*>(stats);
    const double* xr = reinterpret_cast<const double*>(reinterpret_cast<const char*>(stats) + sp.main_bytes);
    return sscg_finalize_conv_stats(main, sp.valid_tiles, sp.bm, sp.wm, xr, sp.xrec, sp.xgroup, G, (long)L, d->K, eps, mean, rstd,
                                    running_mean, running_var, momentum, (hipStream_t)stream);
}

// Data gradient (and ConvTranspose2d forward): dx[n][iy][ix][c] = sum_{ky,kx,k} dy[n][oy][ox][k] * wt[c][ky][kx][k]
// with oy*stride = iy + pad - ky*dil.  `wt` is the weight re-laid as [C][R][S][K] (sscg_weight_krsc_to_crsk).
// Zero padding only: reflection padding is used by the reference exclusively inside the frozen
// generators (arch/generators.py:73,84,89 via model.py:225-228), which never see a backward pass;
// the Python layer materialises the pad for any other caller.
extern "C" int sscg_conv2d_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, const float* bias,
                                 void* dx, int act, float slope, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dy || !wt || !dx) return SSCG_ERR_BAD_ARG;
    if (d->pad_mode != 0) return SSCG_ERR_UNSUPPORTED;
    if (sscg_thin1x1_dgrad_applies(d, bias, act)) return sscg_thin1x1_dgrad(d, dy, wt, dx, (hipStream_t)stream);
    if (sscg_conv16_dgrad_applies(d)) return sscg_conv16_dgrad(d, dy, wt, bias, dx, act, slope, ws, ws_bytes, (hipStream_t)stream);
    if (sscg_convs_dgrad_applies(d) && act != SSCG_ACT_TANH) return sscg_convs_dgrad(d, dy, wt, bias, dx, act, slope, ws, ws_bytes, (hipStream_t)stream);      // (the split family's epilogue knows none / ReLU / LeakyReLU)
    if (d->y_dtype != SSCG_F32 || d->w_dtype != SSCG_F32) return SSCG_ERR_UNSUPPORTED;
    KcParams p = {};
    p.src = reinterpret_cast<const float*>(dy); p.wgt = reinterpret_cast<const float*>(wt); p.bias = bias; p.dst = dx;
    p.out_bf16 = d->x_dtype == SSCG_BF16; p.precision = d->precision; p.tuning = d->tuning;
    p.M = d->N * d->H * d->W; p.Ng = d->C; p.Cs = d->K; p.Ktot = d->R * d->S * d->K;
    p.SH = d->P; p.SW = d->Q; p.OH = d->H; p.OW = d->W;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.pad_mode = 0; p.act = act; p.slope = slope; p.tiles_n = 0; p.tiles = 0;
    kc_dense_taps(p);
    if (dgrad_by_parity(d)) {
        // Stride 2: an output pixel (2i+a, 2j+b) only meets the taps with ky = a+pad, kx = b+pad (mod 2).  Each of the
        // four parity classes is a stride-1 data gradient over its own sub-lattice of taps, written interleaved into
        // dx: a quarter of the multiply-adds of walking all R*S taps with three quarters of them masked.
        p.splits = 1; p.ksplit = 0; p.part = nullptr;
        p.stride = 1; p.wt_step = 2; p.wt_S = d->S;
        p.o_step = 2; p.o_W = d->W; p.o_HW = d->H * d->W;
        for (int a = 0; a < 2; ++a) {
            for (int b = 0; b < 2; ++b) {
                const int Ha = (d->H - a + 1) / 2, Wb = (d->W - b + 1) / 2;
                if (Ha <= 0 || Wb <= 0) continue;
                const int ky0 = (a + d->pad) & 1, kx0 = (b + d->pad) & 1;
                KcParams q = p;
                q.R = ky0 < d->R ? (d->R - ky0 + 1) / 2 : 0;
                q.S = kx0 < d->S ? (d->S - kx0 + 1) / 2 : 0;
                if (q.R == 0 || q.S == 0) { q.R = 0; q.S = 0; }     // no tap meets this class: dx = act(bias)
                q.pad = (a + d->pad - ky0) / 2;
                q.pad_x = (b + d->pad - kx0) / 2;
                q.wt_ky0 = ky0; q.wt_kx0 = kx0;
                q.o_a = a; q.o_b = b;
                q.OH = Ha; q.OW = Wb;
                q.M = d->N * Ha * Wb;
                q.Ktot = q.R * q.S * q.Cs;
                rc = dispatch_mode<MODE_DGRAD>(q, (hipStream_t)stream);
                if (rc) return rc;
            }
        }
        return SSCG_OK;
    }
    KcSplit sp = plan_kc_split(p.M, p.Ng, p.Ktot, p.Cs, p.tuning);
    if (sp.splits > 1 && (!ws || ws_bytes < kc_split_bytes(sp, p.M, p.Ng))) return SSCG_ERR_WORKSPACE;
    p.splits = sp.splits; p.ksplit = sp.ksplit; p.full_tiles = sp.full_tiles; p.m_tail0 = sp.m_tail0;
    p.part = reinterpret_cast<float*>(ws);
    return dispatch_mode<MODE_DGRAD>(p, (hipStream_t)stream);
}

// ---- data gradient + the backward sums of the normalisation layer in front (include/sscg.h)
static bool bsums_geometry(const sscg_conv_desc* d, int G, int64_t L, int* bm, int* wm, int* chunks) {
    if (check_desc(d) || d->pad_mode != 0) return false;
    if (sscg_thin1x1_dgrad_applies(d, nullptr, SSCG_ACT_NONE)) return false;
    if (sscg_conv16_dgrad_applies(d)) return sscg_conv16_bsums_geometry(d, G, (long)L, bm, wm, chunks);
    return sscg_convs_bsums_geometry(d, G, (long)L, bm, wm, chunks);
}

extern "C" size_t sscg_conv2d_dgrad_bsums_bytes(const sscg_conv_desc* d, int G, int64_t L) {
    int bm, wm, chunks;
    if (!d || !bsums_geometry(d, G, L, &bm, &wm, &chunks)) return 0;
    return (size_t)G * chunks * d->C * 2 * sizeof(double);
}

extern "C" int sscg_conv2d_dgrad_bsums(const sscg_conv_desc* d, const void* dy, const void* wt, void* dx, const void* nx, const void* nz,
                                       const void* addend, const float* mean, const float* rstd, const float* gamma, const float* beta, int G,
                                       int64_t L, int act, float slope, void* sums, size_t sums_bytes, void* ws, size_t ws_bytes,
                                       void* stream) {
    if (!d || !dy || !wt || !dx || !nx || !mean || !rstd || !sums) return SSCG_ERR_BAD_ARG;
    if ((gamma != nullptr) != (beta != nullptr)) return SSCG_ERR_BAD_ARG;
    if (act != SSCG_ACT_NONE && act != SSCG_ACT_RELU && act != SSCG_ACT_LRELU) return SSCG_ERR_UNSUPPORTED;      // the mask is recomputed from nx / read off nz
    const size_t need = sscg_conv2d_dgrad_bsums_bytes(d, G, L);
    if (need == 0) return SSCG_ERR_UNSUPPORTED;
    if (sums_bytes < need) return SSCG_ERR_WORKSPACE;
    sscg_bsums bs = {nx, nz, mean, rstd, gamma, beta, sums, G, (long)L, act, slope};
    if (sscg_conv16_dgrad_applies(d)) {
        if (nz || addend) return SSCG_ERR_UNSUPPORTED;          // (the bf16 kernel: sums of residual-free units only)
        return sscg_conv16_dgrad(d, dy, wt, nullptr, dx, SSCG_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream, &bs);
    }
    return sscg_convs_dgrad(d, dy, wt, nullptr, dx, SSCG_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream, &bs, addend);
}

// dx = dgrad(dy, wt) + addend: the fan-in of a tensor with two consumers (a residual block's input: conv1 and the shortcut) joins in
// the data gradient's store phase instead of a separate add pass.  The split family (fp32 tensors) and the bf16 family.
extern "C" int sscg_conv2d_dgrad_add_applies(const sscg_conv_desc* d) {
    if (!d || check_desc(d) != SSCG_OK || d->pad_mode != 0 || sscg_thin1x1_dgrad_applies(d, nullptr, SSCG_ACT_NONE)) return 0;
    if (sscg_conv16_dgrad_applies(d)) return sscg_conv16_dgrad_add_applies(d) ? 1 : 0;
    return sscg_convs_dgrad_applies(d) ? 1 : 0;
}

extern "C" int sscg_conv2d_dgrad_add(const sscg_conv_desc* d, const void* dy, const void* wt, const void* addend, void* dx, void* ws,
                                     size_t ws_bytes, void* stream) {
    if (!d || !dy || !wt || !addend || !dx) return SSCG_ERR_BAD_ARG;
    if (!sscg_conv2d_dgrad_add_applies(d)) return SSCG_ERR_UNSUPPORTED;
    if (sscg_conv16_dgrad_applies(d))
        return sscg_conv16_dgrad(d, dy, wt, nullptr, dx, SSCG_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream, nullptr, addend);
    return sscg_convs_dgrad(d, dy, wt, nullptr, dx, SSCG_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream, nullptr, addend);
}

bool sscg_bsums_records(const sscg_conv_desc* d, int G, int64_t L, int* bm, int* wm, int* chunks) { return d && bsums_geometry(d, G, L, bm, wm, chunks); }

// [K][RS][C] -> [C][RS][K] (weights are a few MB; one pass per optimiser step per conv that needs dgrad)
template <typename S, typename D>
__global__ void krsc_to_crsk_kernel(const S* __restrict__ w, D* __restrict__ wt, int K, int RS, int C) {
    __shared__ float t[32][33];
    const int rs = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int k = k0 + r, c = c0 + tx;
        t[r][tx] = (k < K && c < C) ? ld1<S>(w + ((size_t)k * RS + rs) * C + c) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, k = k0 + tx;
        if (k < K && c < C) st1<D>(wt + ((size_t)c * RS + rs) * K + k, t[tx][r]);
    }
}

// every stale operand copy of a step in ONE launch: a block finds its job by bisection over the table's first-block column
__global__ __launch_bounds__(256) void krsc_to_crsk_batch_kernel(const sscg_wt_job* __restrict__ jobs, int n_jobs) {
    __shared__ float t[32][33];
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const sscg_wt_job j = jobs[lo];
    int b = (int)blockIdx.x - j.block0;
    const int K = j.K, RS = j.RS, C = j.C;
    const int tc = (C + 31) >> 5, tk = (K + 31) >> 5;
    const int c0 = (b % tc) * 32; b /= tc;
    const int k0 = (b % tk) * 32;
    const int rs = b / tk;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, c = c0 + tx;
        float v = 0.f;
        if (k < K && c < C) {
            const size_t i = ((size_t)k * RS + rs) * C + c;
            v = j.w_dtype == SSCG_F32 ? reinterpret_cast<const float*>(j.w)[i] : ld1<__bf16>(reinterpret_cast<const __bf16*>(j.w) + i);
        }
        t[r][tx] = v;
    }
    __syncthreads();
    const size_t plane = (size_t)K * RS * C;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, k = k0 + tx;
        if (k < K && c < C) {
            const size_t o = ((size_t)c * RS + rs) * K + k;
            const float v = t[tx][r];
            if (j.wt_dtype == SSCG_F32) reinterpret_cast<float*>(j.wt)[o] = v;
            else if (j.wt_dtype == SSCG_BF16) st1<__bf16>(reinterpret_cast<__bf16*>(j.wt) + o, v);
            else {
                const sscg_bf3 s3 = sscg_split3(v);
                __bf16* wt = reinterpret_cast<__bf16*>(j.wt);
                wt[o] = s3.h; wt[plane + o] = s3.m; wt[2 * plane + o] = s3.l;
            }
        }
    }
}

extern "C" int sscg_weight_krsc_to_crsk_batch(const sscg_wt_job* jobs, int n_jobs, int n_blocks, void* stream) {
    if (!jobs || n_jobs <= 0 || n_blocks <= 0) return SSCG_ERR_BAD_ARG;
    hipLaunchKernelGGL(krsc_to_crsk_batch_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

extern "C" int sscg_weight_krsc_to_crsk(const void* w, int w_dtype, void* wt, int wt_dtype, int K, int RS, int C, void* stream) {
    if (!w || !wt || K <= 0 || RS <= 0 || C <= 0 || !dt_ok(w_dtype) || !wdt_ok(wt_dtype)) return SSCG_ERR_BAD_ARG;
    if (wt_dtype == SSCG_BF16X3) {      // three dense planes of [C][RS][K] bf16
        if (w_dtype != SSCG_F32) return SSCG_ERR_UNSUPPORTED;
        return sscg_krsc_to_crsk_split(reinterpret_cast<const float*>(w), wt, K, RS, C, (hipStream_t)stream);
    }
    dim3 grid(cdiv(C, 32), cdiv(K, 32), RS);
    hipStream_t st = (hipStream_t)stream;
    if (w_dtype == SSCG_F32 && wt_dtype == SSCG_F32)
        hipLaunchKernelGGL((krsc_to_crsk_kernel<float, float>), grid, dim3(256), 0, st, (const float*)w, (float*)wt, K, RS, C);
    else if (w_dtype == SSCG_F32)
        hipLaunchKernelGGL((krsc_to_crsk_kernel<float, __bf16>), grid, dim3(256), 0, st, (const float*)w, (__bf16*)wt, K, RS, C);
    else if (wt_dtype == SSCG_F32)
        hipLaunchKernelGGL((krsc_to_crsk_kernel<__bf16, float>), grid, dim3(256), 0, st, (const __bf16*)w, (float*)wt, K, RS, C);
    else
        hipLaunchKernelGGL((krsc_to_crsk_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)w, (__bf16*)wt, K, RS, C);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}
