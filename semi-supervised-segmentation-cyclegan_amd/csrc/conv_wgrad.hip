// Weight gradient of Conv2d for NHWC fp32 tensors on gfx950 matrix cores.
//
//   dW[k][ky][kx][c] = sum_{n,oy,ox} dy[n][oy][ox][k] * x[n][oy*s - pad + ky*dil][ox*s - pad + kx*dil][c]
//
// (autograd of the convolutions at arch/generators.py:325-336,373,388 and arch/discriminators.py:70-75
//  that model.py:472,539 triggers via gen_loss.backward() / discriminator_loss.backward().)
//
// GEMM view: D[m][n] = sum_p A[p][m] * B[p][n],  m = output channel k, n = (tap, c), p = output pixel.
// Both operands are contiguous along their M/N index in HBM (channels last), strided along the
// reduction index p ("MC" operands): the LDS image is [p][m] / [p][n]; fragment reads are
// ds_read_b32 with lane -> consecutive m (conflict free), lane half -> p parity.
// The reduction (N*P*Q pixels) is long and the output (K x R*S*C) small, so the pixel range is split
// across workgroups; partial tiles go to a caller-provided workspace and a second kernel reduces them
// in a fixed order (deterministic: no atomics).
#include "common.h"
#include <cstdlib>
#include "sscg_internal.h"

namespace {

constexpr int BKP = 32;  // pixels per k-step

struct WgParams {
    const void* __restrict__ x;       // fp32 or bf16 (kernel template TX)
    const void* __restrict__ dy;      // fp32 or bf16 (kernel template TY)
    float* __restrict__ out;  // dw (splits == 1) or workspace [splits][Kc][Ng]
    int Kc;                   // output channels (GEMM M)
    int Ng;                   // R*S*C (GEMM N)
    int C;
    int H, W, P, Q;
    int S;                    // kernel width (tap decode)
    int stride, pad, dil, pad_mode;
    int npix;                 // N*P*Q
    int chunk;                // pixels per split (multiple of BKP)
    int tiles_n;
    int tiles;                // tiles_m * tiles_n
    int splits;
    float beta;               // applied only when splits == 1
    FastDiv div_pq, div_q;
};

// 256 B of zeros: the source of masked LDS-DMA lanes (device code is not linked across translation units)
__device__ float sscg_zero_page[64];

// TX / TY: element types of x / dy in HBM (the LDS image and the contraction are fp32 either way; bf16 operands are widened by
// the register-staged loader - the mixed pairs are the layers at a network's fp32 boundary: stems read fp32 images, heads emit
// fp32 logits, everything between is bf16 and runs on conv_bf16.hip)
template <int WM, int WN, int TM, int TN, int VA, int VB, bool DMA = false, int BF16 = 0, typename TX = float, typename TY = float>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgParams p) {
    static_assert(!DMA || (VA == 4 && VB == 4), "LDS-DMA staging needs 16-byte granules");
    static_assert(!DMA || (sizeof(TX) == 4 && sizeof(TY) == 4), "LDS-DMA copies fp32 tiles");
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    // register staging pads the rows; LDS-DMA writes lane-linear, i.e. the unpadded [pixel][channel] image, which the
    // ds_read_b32 fragment reads (lane -> consecutive channel) take without bank conflicts anyway
    constexpr int LDA = DMA ? BM : BM + 4;
    constexpr int LDB = DMA ? BN : BN + 4;
    constexpr int CA = BM / VA;          // threads per A row
    constexpr int CB = BN / VB;
    constexpr int RA = 256 / CA;         // rows per pass
    constexpr int RB = 256 / CB;
    constexpr int PA = BKP / RA;
    constexpr int PB = BKP / RB;
    static_assert(256 % CA == 0 && 256 % CB == 0 && BKP % RA == 0 && BKP % RB == 0, "loader geometry");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* As = reinterpret_cast<float*>(smem_raw);  // [2][BKP][LDA]
    float* Bs = As + 2 * BKP * LDA;                  // [2][BKP][LDB]

    const int tid = threadIdx.x;
    // flattened (split, tile) space, XCD-aware: an XCD works on few pixel ranges => dy/x chunks stay in its L2
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int split = lin / p.tiles;
    const int tl = lin - split * p.tiles;
    const int tile_n = tl % p.tiles_n;
    const int tile_m = tl / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int p_begin = split * p.chunk;
    const int p_end = min(p.npix, p_begin + p.chunk);

    // A loader: fixed column group per thread
    const int ca = tid % CA, ra0 = tid / CA;
    const int ma = m0 + ca * VA;
    const bool a_col_ok = ma < p.Kc;
    // B loader: fixed column group -> fixed (tap, c)
    const int cb = tid % CB, rb0 = tid / CB;
    const int nb = n0 + cb * VB;
    const bool b_col_ok = nb < p.Ng;
    int tdy = 0, tdx = 0, cch = 0;
    if (b_col_ok) {
        int tap = nb / p.C;
        cch = nb - tap * p.C;
        int ky = tap / p.S;
        int kx = tap - ky * p.S;
        tdy = ky * p.dil - p.pad;
        tdx = kx * p.dil - p.pad;
    }

    float ra[PA][VA];
    float rb[PB][VB];
    unsigned okmask = 0;   // applied at LDS-store time: the staged registers stay untouched under the MFMAs

    const TY* dyc = reinterpret_cast<const TY*>(p.dy) + (a_col_ok ? ma : 0);      // this thread's dy column
    const TX* xc = reinterpret_cast<const TX*>(p.x) + cch;                        // this thread's x channel group
    const bool reflect = p.pad_mode == 1;

    int dma_buf = 0;
    const int wave_id = tid >> 6;
    auto load_tile = [&](int pt) {
        okmask = 0;
        if constexpr (DMA) {
            float* la = As + dma_buf * BKP * LDA + wave_id * 256;
            float* lb = Bs + dma_buf * BKP * LDB + wave_id * 256;
#pragma unroll
            for (int ps = 0; ps < PA; ++ps) {
                const int pix = pt + ra0 + ps * RA;
                const bool ok = a_col_ok && pix < p_end;
                const void* g = ok ? (const void*)(dyc + (size_t)pix * p.Kc) : (const void*)sscg_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(la + ps * RA * LDA), 16, 0, 0);
            }
#pragma unroll
            for (int ps = 0; ps < PB; ++ps) {
                const int pix = pt + rb0 + ps * RB;
                bool ok = b_col_ok && pix < p_end;
                const int pp = ok ? pix : 0;
                const int img = fd_div(pp, p.div_pq);
                const int rem = pp - img * (p.P * p.Q);
                const int oy = fd_div(rem, p.div_q);
                const int ox = rem - oy * p.Q;
                int sy = oy * p.stride + tdy;
                int sx = ox * p.stride + tdx;
                int ry = sy < 0 ? -sy : sy;
                int rx = sx < 0 ? -sx : sx;
                ry = ry >= p.H ? 2 * (p.H - 1) - ry : ry;
                rx = rx >= p.W ? 2 * (p.W - 1) - rx : rx;
                sy = reflect ? ry : sy;
                sx = reflect ? rx : sx;
                ok = ok && ((unsigned)sy < (unsigned)p.H) && ((unsigned)sx < (unsigned)p.W);
                const void* g = ok ? (const void*)(xc + (size_t)((img * p.H + sy) * p.W + sx) * p.C) : (const void*)sscg_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lb + ps * RB * LDB), 16, 0, 0);
            }
            dma_buf ^= 1;
            return;
        }
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            const int pix = pt + ra0 + ps * RA;
            const bool ok = a_col_ok && pix < p_end;
            okmask |= ok ? (1u << ps) : 0u;
            const TY* g = dyc + (size_t)(ok ? pix : 0) * p.Kc;
            if constexpr (VA == 4) ld4<TY>(g, ra[ps]);
            else ra[ps][0] = ld1<TY>(g);
        }
#pragma unroll
        for (int ps = 0; ps < PB; ++ps) {
            const int pix = pt + rb0 + ps * RB;
            bool ok = b_col_ok && pix < p_end;
            const int pp = ok ? pix : 0;
            const int img = fd_div(pp, p.div_pq);
            const int rem = pp - img * (p.P * p.Q);
            const int oy = fd_div(rem, p.div_q);
            const int ox = rem - oy * p.Q;
            int sy = oy * p.stride + tdy;
            int sx = ox * p.stride + tdx;
            int ry = sy < 0 ? -sy : sy;
            int rx = sx < 0 ? -sx : sx;
            ry = ry >= p.H ? 2 * (p.H - 1) - ry : ry;
            rx = rx >= p.W ? 2 * (p.W - 1) - rx : rx;
            sy = reflect ? ry : sy;
            sx = reflect ? rx : sx;
            ok = ok && ((unsigned)sy < (unsigned)p.H) && ((unsigned)sx < (unsigned)p.W);
            okmask |= ok ? (1u << (16 + ps)) : 0u;
            const int spix = ok ? (img * p.H + sy) * p.W + sx : 0;
            const TX* g = xc + (size_t)spix * p.C;
            if constexpr (VB == 4) ld4<TX>(g, rb[ps]);
            else rb[ps][0] = ld1<TX>(g);
        }
    };

    auto store_tile = [&](int buf) {
        if constexpr (DMA) return;
        float* a = As + buf * BKP * LDA;
        float* b = Bs + buf * BKP * LDB;
#pragma unroll
        for (int ps = 0; ps < PA; ++ps) {
            float* d = a + (ra0 + ps * RA) * LDA + ca * VA;
            const bool ok = (okmask >> ps) & 1u;
            if constexpr (VA == 4) {
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
            float* d = b + (rb0 + ps * RB) * LDB + cb * VB;
            const bool ok = (okmask >> (16 + ps)) & 1u;
            if constexpr (VB == 4) {
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

    const int nsteps = (p_end - p_begin + BKP - 1) / BKP;
    if (nsteps > 0) {
        load_tile(p_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int it = 0; it < nsteps; ++it) {
        const int buf = it & 1;
        if (it + 1 < nsteps) load_tile(p_begin + (it + 1) * BKP);
        const float* a = As + buf * BKP * LDA + lh * LDA + row_w + li;
        const float* b = Bs + buf * BKP * LDB + lh * LDB + col_w + li;
        if constexpr (BF16) {
            // bf16 contraction (see conv_igemm.hip): eight of this lane's pixel samples -> one operand of a k = 16 MFMA;
            // the lane half h keeps the pixel parity it has in the fp32 walk, A and B alike
#pragma unroll
            for (int g2 = 0; g2 < BKP / 16; ++g2) {
                if constexpr (BF16 == 2) {      // split contraction (common.h sscg_split8): fp32-accurate on the bf16 matrix cores
                    bf16x8 a0[TM], a1[TM], a2[TM], b0[TN], b1[TN], b2[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = a[(g2 * 8 + e) * 2 * LDA + i * 32];
                        sscg_split8(v, a0[i], a1[i], a2[i]);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = b[(g2 * 8 + e) * 2 * LDB + j * 32];
                        sscg_split8(v, b0[j], b1[j], b2[j]);
                    }
                    // six piece products, smallest terms first; consecutive MFMAs write different accumulators
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                const bf16x8& pa = t == 0 ? a2[i] : (t == 1 || t == 3) ? a1[i] : a0[i];
                                const bf16x8& pb = (t == 0 || t == 3 || t == 5) ? b0[j] : (t == 1 || t == 4) ? b1[j] : b2[j];
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[i][j], 0, 0, 0);
                            }
                    continue;
                }
                bf16x8 pa[TM], pb[TN];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int kp = g2 * 8 + e;
#pragma unroll
                    for (int i = 0; i < TM; ++i) pa[i][e] = (__bf16)a[kp * 2 * LDA + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) pb[j][e] = (__bf16)b[kp * 2 * LDB + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i], pb[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int kp = 0; kp < BKP / 2; ++kp) {
            float fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = a[kp * 2 * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = b[kp * 2 * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        }
        if (it + 1 < nsteps) store_tile(buf ^ 1);
        __syncthreads();
    }

    float* out = p.out + (size_t)split * p.Kc * p.Ng;
    const bool direct = (p.splits == 1);
    if constexpr (VB == 4) {            // VB == 4 <=> C % 4 == 0 <=> Ng % 4 == 0: the tile leaves through LDS as 16-byte row segments (common.h)
        sscg_stage_store_tile<BM, BN, 256, TM, TN>(acc, reinterpret_cast<float*>(smem_raw), out, m0, n0, p.Kc, p.Ng, row_w, col_w, li, lh, tid,
                                                   direct ? p.beta : 0.f);
        return;
    }
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
                    size_t o = (size_t)m * p.Ng + n;
                    float v = acc[i][j][e];
                    if (direct && p.beta != 0.f) v += p.beta * out[o];
                    out[o] = v;
                }
            }
        }
    }
}

// dw[i] = beta * dw[i] + sum_s ws[s][i]  (fixed order => deterministic).  V floats per thread; the split loop is
// unrolled so that 16 independent loads are in flight per thread (the kernel is a pure stream of splits * |dw| bytes; a 1x1 layer's
// 256 K outputs are 256 workgroups = four waves per CU: 8 loads each left a third of the memory system's latency-bandwidth product unused).
template <int V>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n, int splits,
                                                            float beta) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= n) return;
    typedef float vec_t __attribute__((ext_vector_type(V)));
    vec_t s = 0.f;
#pragma unroll 16
    for (int k = 0; k < splits; ++k) s += *reinterpret_cast<const vec_t*>(ws + (size_t)k * n + i);
    vec_t* d = reinterpret_cast<vec_t*>(dw + i);
    *d = (beta != 0.f) ? beta * *d + s : s;
}

// The same sum with the splits dealt out to the four waves of a block (round 6): wave w adds splits w, w + 4, ... of 64 consecutive
// 16-byte output pieces - with 32 splits every thread has its 8 loads in flight at once and the launch is one memory round trip
// instead of two, on 4x the workgroups (a 1x1 layer's 256 K outputs: 1024 blocks instead of 256) - the four partial sums meet in LDS
// in a fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_w4_kernel(const float* __restrict__ ws, float* __restrict__ dw, size_t n, int splits,
                                                               float beta) {
    __shared__ f32x4 sm[3][64];
    const int v = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + v) * 4;
    f32x4 s = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int k = sl; k < splits; k += 4) s += *reinterpret_cast<const f32x4*>(ws + (size_t)k * n + i);
    }
    if (sl) sm[sl - 1][v] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        s = ((s + sm[0][v]) + sm[1][v]) + sm[2][v];
        f32x4* d = reinterpret_cast<f32x4*>(dw + i);
        *d = (beta != 0.f) ? beta * *d + s : s;
    }
}

struct WgPlan {
    int cfg;
    int bm, bn;
    int splits;
    int chunk;
};


// Cost model of one (tile class, pixel splits) choice, fitted to tools/wgrad_sweep.py on the MI355X:
//   workgroups spread over 256 CUs => a CU runs r = ceil(tiles*splits / 256) of them (co-resident or back to back, the
//   time is the same: they share the CU's matrix cores), each ceil(steps/splits) k-steps plus a prologue/epilogue worth
//   c0 steps; fewer than 2 (128x128) / 4 (64x64) co-resident workgroups leave load latency exposed (factor f);
//   the partial tiles cost splits * |dw| bytes written and read again by the reduction.
struct WgClass { int cfg, bm, bn; float t_step_us, c0; float f[4]; };
const WgClass WG_CLASSES[] = {
    {0, 128, 128, 2.13f, 1.5f, {1.25f, 1.f, 1.f, 1.f}},
    {1, 64, 64, 0.59f, 2.f, {1.5f, 1.17f, 1.08f, 1.f}},
    {2, 32, 128, 0.70f, 2.f, {1.5f, 1.17f, 1.08f, 1.f}},
    {3, 128, 32, 0.70f, 2.f, {1.5f, 1.17f, 1.08f, 1.f}},
};

float wgrad_cost_us(const WgClass& c, long tiles, long steps, long s, double out_bytes) {
    long r = cdiv(tiles * s, 256);
    float f = c.f[r > 4 ? 3 : r - 1];
    float t = (float)r * ((float)cdiv(steps, s) + c.c0) * c.t_step_us * f;
    if (s > 1) t += (float)(2.0 * s * out_bytes / 4.0e6) + 4.f;
    return t;
}

WgPlan plan_wgrad(const sscg_conv_desc* d) {
    WgPlan pl;
    const int Kc = d->K;
    const int Ng = d->R * d->S * d->C;
    const long npix = (long)d->N * d->P * d->Q;
    const long steps = cdiv(npix, BKP);
    const double out_bytes = (double)Kc * Ng * sizeof(float);
    int first = 0, last = 1;   // candidate classes
    // split contraction: the 64x64 class (one fragment per operand and wave) spends 14.7 VALU operations per MFMA on the operand split,
    // the 128x128 class 7.3 - and the VALU pipe is what the step's concurrent kernels contend for (149.5 -> 147.0 ms per step)
    if (d->precision == 2 && Kc >= 128 && Ng >= 128) last = 0;
    if (Kc <= 32) first = last = 2;
    else if (Ng <= 32 || d->C < 32) first = last = 3;
    else if (Kc <= 64 || Ng <= 64) first = last = 1;
    float best = 0.f;
    long best_s = 1;
    pl.cfg = -1;
    for (int ci = first; ci <= last; ++ci) {
        const WgClass& c = WG_CLASSES[ci];
        const long tiles = (long)cdiv(Kc, c.bm) * cdiv(Ng, c.bn);
        const long smax = steps / 2 < 1 ? 1 : (steps / 2 > 512 ? 512 : steps / 2);
        long prev = 0;
        for (int k = 0; k <= 16; ++k) {       // splits that fill k rounds of 256 workgroups (k = 0: unsplit)
            long sp = k == 0 ? 1 : (256L * k) / tiles;
            if (sp < 1) sp = 1;
            if (sp > smax) sp = smax;
            if (sp == prev) continue;
            prev = sp;
            float t = wgrad_cost_us(c, tiles, steps, sp, out_bytes);
            if (pl.cfg < 0 || t < best) { best = t; best_s = sp; pl.cfg = c.cfg; pl.bm = c.bm; pl.bn = c.bn; }
        }
    }
    long splits = best_s;
    // sscg_conv_desc.wgrad_tuning (tools/wgrad_sweep.py, tile-class tests): bits 0..7 = 1 + forced class (0 = 128x128, 1 = 64x64),
    // bits 8..23 = forced pixel splits
    const int force_cfg = (d->wgrad_tuning & 0xff) - 1, force_splits = (d->wgrad_tuning >> 8) & 0xffff;
    if (force_cfg >= 0 && force_cfg <= 1 && force_splits > 0 && pl.cfg <= 1) {
        pl.cfg = force_cfg;
        pl.bm = pl.bn = pl.cfg == 0 ? 128 : 64;
        splits = force_splits;
        if (splits > steps) splits = steps;
    }
    long steps_per = cdiv(steps, splits);
    pl.chunk = (int)(steps_per * BKP);
    pl.splits = cdiv(npix, pl.chunk);
    return pl;
}

template <int WM, int WN, int TM, int TN, int VA, int VB, bool DMA = false, int BF16 = 0, typename TX = float, typename TY = float>
int launch_wg(WgParams p, int splits, hipStream_t st) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    p.tiles_n = cdiv(p.Ng, BN);
    int tiles_m = cdiv(p.Kc, BM);
    p.tiles = tiles_m * p.tiles_n;
    p.splits = splits;
    size_t smem = (size_t)(2 * BKP * (DMA ? BM : BM + 4) + 2 * BKP * (DMA ? BN : BN + 4)) * sizeof(float);
    if (VB == 4 && smem < (size_t)BM * (BN + 4) * sizeof(float)) smem = (size_t)BM * (BN + 4) * sizeof(float);      // the staged result tile
    auto kern = conv_wgrad_kernel<WM, WN, TM, TN, VA, VB, DMA, BF16, TX, TY>;
    SSCG_ENSURE_SMEM((kern), smem);
    hipLaunchKernelGGL(kern, dim3(p.tiles * splits), dim3(256), smem, st, p);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

template <int VA, int VB>
int dispatch_wg(const WgParams& p, const WgPlan& pl, int precision, hipStream_t st) {
    switch (pl.cfg) {
        case 0: if (precision == 1) return launch_wg<2, 2, 2, 2, VA, VB, (VA == 4 && VB == 4), 1>(p, pl.splits, st);
                if constexpr (VA == 4 && VB == 4) { if (precision == 2) return launch_wg<2, 2, 2, 2, VA, VB, true, 2>(p, pl.splits, st); }
                return launch_wg<2, 2, 2, 2, VA, VB, (VA == 4 && VB == 4)>(p, pl.splits, st);   // LDS-DMA staging when vectorisable
        case 1: if (precision == 1) return launch_wg<2, 2, 1, 1, VA, VB, (VA == 4 && VB == 4), 1>(p, pl.splits, st);
                if constexpr (VA == 4 && VB == 4) { if (precision == 2) return launch_wg<2, 2, 1, 1, VA, VB, true, 2>(p, pl.splits, st); }
                return launch_wg<2, 2, 1, 1, VA, VB, (VA == 4 && VB == 4)>(p, pl.splits, st);
        case 2: if (precision == 1) return launch_wg<1, 4, 1, 1, VA, VB, false, true>(p, pl.splits, st);
                return launch_wg<1, 4, 1, 1, VA, VB>(p, pl.splits, st);
        case 3: if (precision == 1) return launch_wg<4, 1, 1, 1, VA, VB, false, true>(p, pl.splits, st);
                return launch_wg<4, 1, 1, 1, VA, VB>(p, pl.splits, st);
        default: return SSCG_ERR_BAD_ARG;
    }
}

// mixed element types (one operand bf16): register-staged tiles, bf16 contraction (the bf16 operand is exact in it)
template <int VA, int VB, typename TX, typename TY>
int dispatch_wg_mixed(const WgParams& p, const WgPlan& pl, hipStream_t st) {
    switch (pl.cfg) {   // only the few-channel tile classes occur with mixed types (stems: C < 32; heads: K <= 32)
        case 2: return launch_wg<1, 4, 1, 1, VA, VB, false, true, TX, TY>(p, pl.splits, st);
        case 3: return launch_wg<4, 1, 1, 1, VA, VB, false, true, TX, TY>(p, pl.splits, st);
        default: return SSCG_ERR_UNSUPPORTED;
    }
}

template <typename TX, typename TY>
int dispatch_wg_types(const WgParams& p, const WgPlan& pl, bool va4, bool vb4, hipStream_t st) {
    if (va4 && vb4) return dispatch_wg_mixed<4, 4, TX, TY>(p, pl, st);
    if (va4) return dispatch_wg_mixed<4, 1, TX, TY>(p, pl, st);
    if (vb4) return dispatch_wg_mixed<1, 4, TX, TY>(p, pl, st);
    return dispatch_wg_mixed<1, 1, TX, TY>(p, pl, st);
}

}  // namespace

// ---- "thin" weight gradients: 1x1 convolutions with a handful of channels on one side (PixelDiscriminator:
// 3 -> 64, 21 -> 64, 128 -> 1 at 256x256).  dw[k][c] = sum_p dy[p][k] * x[p][c] over 524288 pixels is a pure stream of
// the wide operand (134 .. 268 MB); on the matrix-core kernel it ran at 0.3 .. 3 TFLOP/s, 6 .. 16x off the HBM roofline.
// Here a thread owns four wide channels, walks a stripe of pixels with the <= 32 thin values of each pixel broadcast to
// it, and keeps T x 4 running sums in registers; stripes are combined through LDS, workgroups through the workspace and
// a fixed-order second stage (deterministic).
namespace {

constexpr int THIN_MAX = 32;
constexpr int THIN_BLOCKS = 1024;

struct ThinParams {
    const void* __restrict__ wide;    // [npix][Wd]  (TW)
    const void* __restrict__ thin;    // [npix][T]   (TT)
    float* __restrict__ part;         // [blocks][T][Wd]
    int npix, Wd, T;
};

template <int T, typename TW, typename TT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(ThinParams p) {
    const TW* wide = reinterpret_cast<const TW*>(p.wide);
    const TT* thin = reinterpret_cast<const TT*>(p.thin);
    constexpr int TB = T < 4 ? T : 4;              // thin channels combined per LDS round (keeps the block at <= 16 KB)
    extern __shared__ float red[];                 // [lanes][TB][Wd]
    const int wq = p.Wd / 4;                       // threads across the wide channels
    const int lanes = 256 / wq;                    // pixel lanes of the block
    const int q = threadIdx.x % wq;
    const int pl = threadIdx.x / wq;
    float acc[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    if (pl < lanes) {
        const int stride = lanes * gridDim.x;
        int px = blockIdx.x * lanes + pl;
        for (; px + stride < p.npix; px += 2 * stride) {       // two pixels in flight per thread
            float w0[4], w1[4];
            ld4<TW>(wide + (size_t)px * p.Wd + q * 4, w0);
            ld4<TW>(wide + (size_t)(px + stride) * p.Wd + q * 4, w1);
            const TT* th0 = thin + (size_t)px * T;
            const TT* th1 = thin + (size_t)(px + stride) * T;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float v0 = ld1<TT>(th0 + t), v1 = ld1<TT>(th1 + t);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(v1, w1[e], fmaf(v0, w0[e], acc[t][e]));
            }
        }
        if (px < p.npix) {
            float w0[4];
            ld4<TW>(wide + (size_t)px * p.Wd + q * 4, w0);
            const TT* th0 = thin + (size_t)px * T;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float v0 = ld1<TT>(th0 + t);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] = fmaf(v0, w0[e], acc[t][e]);
            }
        }
    }
#pragma unroll
    for (int t0 = 0; t0 < T; t0 += TB) {
        if (pl < lanes) {
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (t0 + t < T)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[((size_t)pl * TB + t) * p.Wd + q * 4 + e] = acc[t0 + t][e];
        }
        __syncthreads();
        const int nt = (T - t0) < TB ? (T - t0) : TB;
        for (int i = threadIdx.x; i < nt * p.Wd; i += 256) {
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[(size_t)l * TB * p.Wd + i];
            p.part[(size_t)blockIdx.x * T * p.Wd + (size_t)t0 * p.Wd + i] = s;
        }
        __syncthreads();
    }
}

// dw[k][c] = beta * dw[k][c] + sum_b part[b][t][w]; one wave per output element, lanes stride the blocks (fixed order)
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int blocks,
                                                                 int T, int Wd, int K, int C, float beta) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= T * Wd) return;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += part[(size_t)b * T * Wd + o];
    s = wave_sum(s);
    if (lane == 0) {
        const int t = o / Wd, w = o - t * Wd;
        const int k = (K == T && K <= C) ? t : w;      // which side is the thin one
        const int c = (K == T && K <= C) ? w : t;
        float* d = dw + (size_t)k * C + c;
        *d = beta != 0.f ? beta * *d + s : s;
    }
}

bool thin_wgrad_applies(const sscg_conv_desc* d) {
    if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->pad_mode != 0) return false;
    const int T = d->K < d->C ? d->K : d->C, Wd = d->K < d->C ? d->C : d->K;
    return T <= THIN_MAX && Wd % 4 == 0 && Wd <= 256 && Wd >= 16 && (long)d->N * d->H * d->W >= 65536;
}

template <int T, typename TW, typename TT>
int launch_thin(const ThinParams& p, int blocks, size_t smem, hipStream_t st) {
    SSCG_ENSURE_SMEM((thin_wgrad_kernel<T, TW, TT>), smem);
    hipLaunchKernelGGL((thin_wgrad_kernel<T, TW, TT>), dim3(blocks), dim3(256), smem, st, p);
    return SSCG_OK;
}

template <typename TW, typename TT>
int launch_thin_T(const ThinParams& p, size_t smem, hipStream_t st) {
    switch (p.T) {
        case 1: return launch_thin<1, TW, TT>(p, THIN_BLOCKS, smem, st);
        case 2: return launch_thin<2, TW, TT>(p, THIN_BLOCKS, smem, st);
        case 3: return launch_thin<3, TW, TT>(p, THIN_BLOCKS, smem, st);
        case 4: return launch_thin<4, TW, TT>(p, THIN_BLOCKS, smem, st);
        case 20: return launch_thin<20, TW, TT>(p, THIN_BLOCKS, smem, st);
        case 21: return launch_thin<21, TW, TT>(p, THIN_BLOCKS, smem, st);
        default: return SSCG_ERR_UNSUPPORTED;
    }
}

int thin_wgrad(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta, void* ws, hipStream_t st) {
    const bool thin_is_k = d->K <= d->C;
    ThinParams p;
    p.T = thin_is_k ? d->K : d->C;
    p.Wd = thin_is_k ? d->C : d->K;
    p.wide = thin_is_k ? x : dy;
    p.thin = thin_is_k ? dy : x;
    const int wide_dt = thin_is_k ? d->x_dtype : d->y_dtype, thin_dt = thin_is_k ? d->y_dtype : d->x_dtype;
    p.part = reinterpret_cast<float*>(ws);
    p.npix = d->N * d->H * d->W;
    const int lanes = 256 / (p.Wd / 4);
    const size_t smem = (size_t)lanes * (p.T < 4 ? p.T : 4) * p.Wd * sizeof(float);
    int rc;
    if (wide_dt == SSCG_F32 && thin_dt == SSCG_F32) rc = launch_thin_T<float, float>(p, smem, st);
    else if (wide_dt == SSCG_BF16 && thin_dt == SSCG_F32) rc = launch_thin_T<__bf16, float>(p, smem, st);   // the fp32 side is a network input / head output
    else if (wide_dt == SSCG_BF16 && thin_dt == SSCG_BF16) rc = launch_thin_T<__bf16, __bf16>(p, smem, st);
    else rc = SSCG_ERR_UNSUPPORTED;
    if (rc) return rc;
    SSCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3(cdiv(p.T * p.Wd, 4)), dim3(256), 0, st, p.part, dw, THIN_BLOCKS, p.T, p.Wd,
                       d->K, d->C, beta);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

bool thin_wgrad_supported_T(int T) { return T == 1 || T == 2 || T == 3 || T == 4 || T == 20 || T == 21; }

}  // namespace

int sscg_wgrad_reduce(const float* ws, float* dw, size_t n, int splits, float beta, hipStream_t st) {
    static const bool w4 = !(getenv("SSCG_WGRAD_REDUCE_W4") && atoi(getenv("SSCG_WGRAD_REDUCE_W4")) == 0);       // A/B aid
    const bool vec_ok = n % 4 == 0 && ((size_t)dw & 15) == 0 && ((size_t)ws & 15) == 0;
    if (vec_ok && w4 && splits >= 8 && n / 4 <= 64 * 16384)
        hipLaunchKernelGGL(wgrad_reduce_w4_kernel, dim3(cdiv((long)(n / 4), 64)), dim3(256), 0, st, ws, dw, n, splits, beta);
    else if (vec_ok)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(cdiv((long)(n / 4), 256)), dim3(256), 0, st, ws, dw, n, splits, beta);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(cdiv((long)n, 256)), dim3(256), 0, st, ws, dw, n, splits, beta);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

static bool wg_dt_ok(int dt) { return dt == SSCG_F32 || dt == SSCG_BF16; }

extern "C" size_t sscg_conv2d_wgrad_workspace(const sscg_conv_desc* d) {
    if (!d) return 0;
    if (thin_wgrad_applies(d) && thin_wgrad_supported_T(d->K < d->C ? d->K : d->C))
        return (size_t)THIN_BLOCKS * d->K * d->C * sizeof(float);
    if (sscg_wgrad16_applies(d)) return sscg_wgrad16_workspace(d);
    if (sscg_wgrads_applies(d)) return sscg_wgrads_workspace(d);
    WgPlan pl = plan_wgrad(d);
    if (pl.splits <= 1) return 0;
    return (size_t)pl.splits * d->K * d->R * d->S * d->C * sizeof(float);
}

// dw = beta * dw + wgrad(x, dy);  dw is [K][R][S][C].  ws must hold sscg_conv2d_wgrad_workspace(d) bytes.
extern "C" int sscg_conv2d_wgrad(const sscg_conv_desc* d, const void* x, const void* dy, float* dw, float beta,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!d || !x || !dy || !dw) return SSCG_ERR_BAD_ARG;
    if (d->N <= 0 || d->C <= 0 || d->K <= 0 || !wg_dt_ok(d->x_dtype) || !wg_dt_ok(d->y_dtype)) return SSCG_ERR_BAD_ARG;
    if ((long)d->N * d->H * d->W * (long)d->C >= (1L << 31)) return SSCG_ERR_UNSUPPORTED;
    if ((long)d->N * d->P * d->Q * (long)d->K >= (1L << 31)) return SSCG_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (thin_wgrad_applies(d) && thin_wgrad_supported_T(d->K < d->C ? d->K : d->C)) {
        if (!ws || ws_bytes < sscg_conv2d_wgrad_workspace(d)) return SSCG_ERR_WORKSPACE;
        return thin_wgrad(d, x, dy, dw, beta, ws, st);
    }
    if (sscg_wgrad16_applies(d)) return sscg_wgrad16(d, x, dy, dw, beta, ws, ws_bytes, st);
    if (sscg_wgrads_applies(d)) return sscg_wgrads(d, x, dy, dw, beta, ws, ws_bytes, st);
    WgPlan pl = plan_wgrad(d);
    size_t need = pl.splits > 1 ? (size_t)pl.splits * d->K * d->R * d->S * d->C * sizeof(float) : 0;
    if (need > 0 && (!ws || ws_bytes < need)) return SSCG_ERR_WORKSPACE;
    WgParams p;
    p.x = x; p.dy = dy;
    p.out = pl.splits > 1 ? reinterpret_cast<float*>(ws) : dw;
    p.Kc = d->K; p.Ng = d->R * d->S * d->C; p.C = d->C;
    p.H = d->H; p.W = d->W; p.P = d->P; p.Q = d->Q; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.pad_mode = d->pad_mode;
    p.npix = d->N * d->P * d->Q; p.chunk = pl.chunk; p.tiles_n = 0; p.tiles = 0; p.splits = 1;
    p.beta = pl.splits > 1 ? 0.f : beta;
    p.div_pq = make_fastdiv(d->P * d->Q);
    p.div_q = make_fastdiv(d->Q);
    const bool va4 = (d->K % 4 == 0), vb4 = (d->C % 4 == 0);
    int rc;
    if (d->x_dtype == SSCG_F32 && d->y_dtype == SSCG_F32) {
        if (va4 && vb4) rc = dispatch_wg<4, 4>(p, pl, d->precision, st);
        else if (va4) rc = dispatch_wg<4, 1>(p, pl, d->precision, st);
        else if (vb4) rc = dispatch_wg<1, 4>(p, pl, d->precision, st);
        else rc = dispatch_wg<1, 1>(p, pl, d->precision, st);
    } else if (d->x_dtype == SSCG_F32) {
        rc = dispatch_wg_types<float, __bf16>(p, pl, va4, vb4, st);
    } else if (d->y_dtype == SSCG_F32) {
        rc = dispatch_wg_types<__bf16, float>(p, pl, va4, vb4, st);
    } else {
        rc = SSCG_ERR_UNSUPPORTED;     // both bf16 with fewer than 32 channels on a side: no layer of the reference's nets
    }
    if (rc) return rc;
    if (pl.splits > 1) return sscg_wgrad_reduce(reinterpret_cast<const float*>(ws), dw, (size_t)d->K * p.Ng, pl.splits, beta, st);
    return SSCG_OK;
}
