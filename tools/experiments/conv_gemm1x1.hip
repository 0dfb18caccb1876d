// 1x1 stride-1 convolutions with a SHORT reduction (<= 256 source channels) on fp32 tensors by the split contraction - the DeepLab
// Bottleneck's expanding projection and its data-gradient twin (reference call sites: arch/generators.py:336,348-363 conv3 256 -> 1024
// forward, conv1 1024 -> 256 data gradient; autograd of model.py:472): D[m][n] = sum_k A[m][k] * B[n][k], K <= 256, N = 4 K.
//
// Why a kernel of its own (round 6).  In conv_split.hip's tiling these launches are 552 tiles of 128 x 128 with EIGHT k-tiles each:
// a workgroup spends as long in its prologue (row decode, tap table, 1000 instructions) and epilogue as in its k-loop, every A
// element is split into its three bf16 pieces by two waves of each of the eight column tiles that meet it (16 times), and the step -
// whose contended resource is the VALU pipe - pays 9+ VALU operations per MFMA for them (46 us forward, 72 us data gradient with its
// fused store phase, against 13 us of matrix-core time and 9 / 29 us of HBM time).  Here the A operand is STATIONARY:
//   * a workgroup (8 waves, 2 x 4 of 32 x 32) owns a contiguous range of (64-row, 128-column) units; consecutive units share their 64
//     rows, whose K <= 256 reduction is loaded ONCE, split ONCE (5.5 VALU operations per element, once per element and launch) and
//     kept in LDS as the three bf16 planes an MFMA reads (96 KB at K = 256; 16-byte slots XOR-swizzled with the row: conflict-free
//     ds_read_b128 fragments);
//   * the weight planes stream through two 24 KB LDS stages by LDS-DMA (buffer_load ... lds, as conv_split.hip), across unit
//     boundaries without a bubble;
//   * the k-loop is ds_read_b128 + MFMA only: 12 fragment reads and 12 MFMAs per wave and k-tile, four accumulators (leading piece
//     product / the five small ones, per k-step) - no VALU work besides a handful of address operations;
//   * the epilogue works on the MFMA layout directly (a wave-instruction covers two 128-byte row segments): forward - bias,
//     activation, the fused normalisation statistics in conv_split.hip's record format (one record per 32-row wave block); data
//     gradient - the fused store phase (fan-in addend, the normalisation layer's backward sums with the mask recomputed from its
//     input or read off the unit's output), its row streams requested at the top of the unit's k-loop and consumed after it.
// One workgroup per CU (144 KB of LDS), at most 256 workgroups, each with units / 256 units of work: no tail.
#include "common.h"
#include "sscg_internal.h"
#include <cstdlib>

namespace {

typedef __bf16 bf16;
constexpr int G1_BM = 64, G1_BN = 128, G1_NT = 512, G1_BK = 32;
constexpr int G1_BSTAGE = 3 * G1_BN * 64;      // bytes of one weight stage: three planes of [128 rows][32 k] bf16

struct G1Params {
    const float* __restrict__ src;       // A [M][Kr] fp32
    const bf16* __restrict__ wgt;        // plane 0 of B: [3][Ng][Kr] bf16, planes `wplane` elements apart
    long wplane;
    const float* __restrict__ bias;      // [Ng] or null
    float* __restrict__ dst;             // [M][Ng]
    int M, Ng, Kr;
    int act;
    float slope;
    int units, nchunks;
    unsigned wgt_bytes;
    double* __restrict__ stats;          // forward: [mtiles * 2][2][Ng][2] or null
    int stat_L;
    const float* __restrict__ bn_x;      // data gradient, fused store phase (conv_split.hip KsParams)
    const float* __restrict__ bn_z;
    const float* __restrict__ addend;
    const float* __restrict__ bn_mean;
    const float* __restrict__ bn_rstd;
    const float* __restrict__ bn_gamma;
    const float* __restrict__ bn_beta;
    double* __restrict__ bn_sums;        // [G][chunks][Ng][2], one record per 32-row block and group
    int bn_L, bn_G, bn_chunks, bn_act;
    float bn_slope;
    FastDiv div_gl, div_nc;
};

template <int N> __device__ __forceinline__ void g1_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void g1_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void g1_pin(bf16x8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float g1_act(float v, int act, float slope) {
    const float neg = act == SSCG_ACT_RELU ? 0.f : (act == SSCG_ACT_LRELU ? v * slope : v);
    return v > 0.f ? v : neg;
}

enum { G1_FWD = 0, G1_DGRAD = 1 };
#ifndef G1_ABLATE
#define G1_ABLATE 0      // timing ablations (WRONG results): 1 no copy wait / barrier, 2 no copies after the first, 4 no fragment reads / MFMAs, 8 no store phase
#endif

// JA / JX / JZ: the data gradient's store phase reads the addend / the normalisation layer's input (= takes its backward sums) / the
// unit's output (mask source).  Compile-time: the number of row loads in flight sets the k-loop's vmcnt immediates.
template <int MODE, int KR, int JA, int JX, int JZ>
__global__ __launch_bounds__(G1_NT, 2) void g1x1_kernel(G1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef __attribute__((address_space(3))) char lds_char;
    constexpr int NL = 4 * (JA + JX + JZ);
    const int tid = (int)threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    constexpr int Kr = KR, nkt = KR / G1_BK;
    constexpr int NA = KR / 32;                                         // 16-byte loads per thread of one m-tile's A rows
    constexpr int a_plane = G1_BM * Kr * 2;                             // bytes of one A piece plane
    const int u0 = (int)(((long)blockIdx.x * p.units) / gridDim.x), u1 = (int)(((long)(blockIdx.x + 1) * p.units) / gridDim.x);
    if (u0 >= u1) return;
    lds_char* const lds0 = (lds_char*)smem_raw;
    constexpr int b0 = 3 * a_plane;                                         // first weight stage
    constexpr int sc0 = b0 + 2 * G1_BSTAGE;                                 // the waves' store-phase scratch: 2 KB each

    // ---- weight copies (conv_split.hip's B side: 4 lanes x 16 B per 64-byte row, slot XOR (row >> 2) & 3 on the SOURCE side)
    const int rb0 = tid >> 2;                                           // row of the 128-row pass
    const int kqb = (tid & 3) ^ ((rb0 >> 2) & 3);
    const unsigned brow = ((unsigned)rb0 * (unsigned)Kr + kqb * 8) * 2u;
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, (short)0, (int)p.wgt_bytes, 0x00020000);
    int ru = u0, rkt = 0, rstage = 0;                                   // the next tile to request: unit, k-tile, stage
    int rn0 = (u0 - fd_div(u0, p.div_nc) * p.nchunks) * G1_BN;
    auto request = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const int so = __builtin_amdgcn_readfirstlane((int)((pl * p.wplane + (long)rn0 * Kr + rkt * G1_BK) * 2));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds0 + b0 + rstage * G1_BSTAGE + pl * (G1_BN * 64) + wave * 1024),
                                                     16, (int)brow, so, 0, 0);
        }
        rstage ^= 1;
        if (++rkt == nkt) {
            rkt = 0;
            ++ru;
            rn0 += G1_BN;
            if (rn0 >= p.Ng) rn0 = 0;
        }
    };

    // ---- the A rows of an m-tile: loaded, split into three bf16 planes, kept in LDS.  Thread t: row t >> 3, the 16-byte slots
    // (t & 7) + 8 i of its row (8 k each).  Slot q of row r lives at slot q ^ (r & 15) (the low four bits: inside its 256-byte half).
    const int ar = tid >> 3, aq0 = tid & 7;
    constexpr int nslot = KR / 64;                                      // slots per thread: 4 at K = 256
    f32x4 araw[NA];
    auto a_issue = [&](int mt) {        // every load of the tile in flight at once (taken slot by slot they cost a memory latency each)
        const int m = mt * G1_BM + ar;
        const float* row = p.src + (size_t)(m < p.M ? m : p.M - 1) * Kr;
#pragma unroll
        for (int i = 0; i < nslot; ++i) {
            araw[2 * i] = *reinterpret_cast<const f32x4*>(row + (aq0 + 8 * i) * 8);
            araw[2 * i + 1] = *reinterpret_cast<const f32x4*>(row + (aq0 + 8 * i) * 8 + 4);
        }
    };
    auto a_commit = [&](int mt) {
        const bool ok = mt * G1_BM + ar < p.M;
        const int key = ar & 15;
#pragma unroll
        for (int i = 0; i < nslot; ++i) {
            const int q = aq0 + 8 * i;
            f32x4 x0 = araw[2 * i], x1 = araw[2 * i + 1];
            if (!ok) { x0 = 0.f; x1 = 0.f; }
            const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            bf16x8 h, mm, l;
            sscg_split8(x, h, mm, l);
            char* a = smem_raw + ar * (Kr * 2) + ((q ^ key) << 4);
            *reinterpret_cast<bf16x8*>(a) = h;
            *reinterpret_cast<bf16x8*>(a + a_plane) = mm;
            *reinterpret_cast<bf16x8*>(a + 2 * a_plane) = l;
        }
    };

    // ---- fragment addresses
    const int arow = wm * 32 + li;
    const int akey = arow & 15;
    const lds_char* const a_base = lds0 + arow * (Kr * 2);
    const int swb = (li >> 2) & 3;
    const int b_base = b0 + (wn * 32 + li) * 64;

    int mt_cur = fd_div(u0, p.div_nc);
    request();                          // tile (u0, 0) -> stage 0
    a_issue(mt_cur);
    a_commit(mt_cur);
    g1_wait_vm<0>();
    __syncthreads();                    // the pieces are visible; tile (u0, 0) has landed for every wave
    int stage = 0;

    for (int u = u0; u < u1; ++u) {
        const int mt = mt_cur;
        const int nc = u - mt * p.nchunks;
        const bool more = u + 1 < u1;
        const int mt_next = (more && nc + 1 == p.nchunks) ? mt + 1 : mt;       // (workgroup-uniform) the next unit's rows: requested under THIS unit's k-loop
        const bool pre_a = mt_next != mt;
        const int m0 = mt * G1_BM, n0 = nc * G1_BN;
        f32x16 acc0, acc1, lo0, lo1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; lo0[e] = 0.f; lo1[e] = 0.f; }

        // Store-phase geometry: the wave's 32 x 32 block leaves through a 2 KB LDS scratch of its own (two halves of 16 rows) so that a
        // lane holds four consecutive channels of four rows - block rows (lane >> 3) + 8 j, channels (lane & 7) * 4 ... + 3 - and every
        // global access of the epilogue is 16 bytes wide (in the MFMA layout a lane owns single elements of 16 rows: 16 four-byte stores
        // per lane, and the store path, not HBM, set the kernel's time: 36 us for a launch whose k-loops take 15).
        const int mw = m0 + wm * 32;                                                // first row of this wave's block
        const int n = n0 + wn * 32 + (lane & 7) * 4;                                // first of the lane's four channels
        const int rl = lane >> 3;
        // row streams of the fused store phase: requested under the k-loop, consumed after it.  Rows past M read row M - 1 (never used).
        f32x4 av[JA ? 4 : 1], xv[JX ? 4 : 1], zv[JZ ? 4 : 1];
        auto row_loads = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int m = mw + rl + 8 * j;
                m = m < p.M ? m : p.M - 1;
                const size_t o = (size_t)m * p.Ng + n;
                if (JA) av[j] = *reinterpret_cast<const f32x4*>(p.addend + o);
                if (JX) xv[j] = *reinterpret_cast<const f32x4*>(p.bn_x + o);
                if (JZ) zv[j] = *reinterpret_cast<const f32x4*>(p.bn_z + o);
            }
        };

        for (int kt = 0; kt < nkt; ++kt) {
            // A fragments of this k-tile do not depend on the copies: requested ahead of the wait
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int s = 0; s < ((G1_ABLATE & 4) ? 0 : 2); ++s) {
                const lds_char* a = a_base + (((kt * 4 + 2 * s + lh) ^ akey) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][0]) : "v"(a));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][1]) : "v"(a + a_plane));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][2]) : "v"(a + 2 * a_plane));
            }
            if (kt > 0 && !(G1_ABLATE & 1)) {
                // tile (u, kt) has landed (kt = 0: waited for before the previous unit's stores); behind tile (u, 1)'s copies sit this unit's row
                // streams and, before an m-tile change, the next tile's A rows: exactly that many younger loads may stay in flight
                if (kt == 1) { if (pre_a) g1_wait_vm<NL + NA>(); else g1_wait_vm<NL>(); } else g1_wait_vm<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (ru < u1 && !((G1_ABLATE & 2) && kt > 0)) request();                 // -> the stage the previous k-tile drained
            if (kt == 0) {                                                          // (pinned behind the request: the k-tile 1 wait counts exactly these younger loads)
                __builtin_amdgcn_sched_barrier(0);
                if (NL > 0) row_loads();
                if (pre_a) a_issue(mt_next);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (G1_ABLATE & 4) { stage ^= 1; continue; }
            const lds_char* b = lds0 + b_base + stage * G1_BSTAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[s][pl]) : "v"(b + (((2 * s + lh) ^ swb) << 4)), "n"(pl * (G1_BN * 64)));
            g1_wait_lgkm<0>();
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { g1_pin(fa[s][pl]); g1_pin(fb[s][pl]); }
            // six piece products per k-step, smallest first; the two k-steps alternate so that consecutive MFMAs write different accumulators
            lo0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][2], fb[0][0], lo0, 0, 0, 0);
            lo1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][2], fb[1][0], lo1, 0, 0, 0);
            lo0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][1], lo0, 0, 0, 0);
            lo1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][1], lo1, 0, 0, 0);
            lo0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][2], lo0, 0, 0, 0);
            lo1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][2], lo1, 0, 0, 0);
            lo0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][1], fb[0][0], lo0, 0, 0, 0);
            lo1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][1], fb[1][0], lo1, 0, 0, 0);
            lo0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][1], lo0, 0, 0, 0);
            lo1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][1], lo1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][0], fb[0][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][0], fb[1][0], acc1, 0, 0, 0);
            stage ^= 1;
        }

        // ---- epilogue.  MFMA C layout: column lane & 31, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> through the wave's scratch ->
        // o[j] = four channels of block row rl + 8 j
        f32x4 o[4];
        if (G1_ABLATE & 8) {
            for (int j = 0; j < 4; ++j) o[j] = f32x4{acc0[j], acc1[j], lo0[j], lo1[j]};
        } else {
            const lds_char* const sw = lds0 + sc0 + wave * 2048 + lh * 512 + li * 4;
            const lds_char* const sr = lds0 + sc0 + wave * 2048 + rl * 128 + (lane & 7) * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = 8 * h + e8;
                    const float t = (lo0[e] + lo1[e]) + (acc0[e] + acc1[e]);
                    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(sw), "v"(t), "n"(((e8 & 3) + 8 * (e8 >> 2)) * 128) : "memory");
                }
                g1_wait_lgkm<0>();
                asm volatile("ds_read_b128 %0, %1" : "=v"(o[2 * h]) : "v"(sr) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(o[2 * h + 1]) : "v"(sr) : "memory");
                g1_wait_lgkm<0>();
                asm volatile("" : "+v"(o[2 * h]), "+v"(o[2 * h + 1]));
            }
        }
        const bool nok = n < p.Ng;
        if (MODE == G1_FWD && p.bias && nok) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += bv;
        }
        // eight lanes (rl = 0..7) hold the rows of one channel quadruple: their partial sums meet in lane rl = 0
        auto lanes8 = [&](double x) {
            x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
            return x;
        };
        if (MODE == G1_FWD && p.stats) {
            // fused statistics of the normalisation layer that follows (conv_split.hip's record format, two wave rows per 64-row tile):
            // four rows in fp32, across the row lanes in fp64 where the tile lies inside one group and inside the tensor
            const int gb = (fd_div(m0, p.div_gl) + 1) * p.stat_L;
            const bool fast = m0 + G1_BM <= gb && m0 + G1_BM <= p.M;
            double* rec = p.stats + ((size_t)(mt * 2 + wm) * 2) * p.Ng * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
                if (fast) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a += o[j][c]; b = fmaf(o[j][c], o[j][c], b); }
                    s0 = (double)a; q0 = (double)b;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mw + rl + 8 * j;
                        if (m < p.M) {
                            const double d = (double)o[j][c];
                            if (m < gb) { s0 += d; q0 += d * d; } else { s1 += d; q1 += d * d; }
                        }
                    }
                }
                s0 = lanes8(s0); q0 = lanes8(q0);
                if (!fast) { s1 = lanes8(s1); q1 = lanes8(q1); }
                if (rl == 0 && nok) {
                    rec[(size_t)(n + c) * 2] = s0; rec[(size_t)(n + c) * 2 + 1] = q0;
                    rec[((size_t)p.Ng + n + c) * 2] = s1; rec[((size_t)p.Ng + n + c) * 2 + 1] = q1;
                }
            }
        }
        if (MODE == G1_FWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) o[j][c] = g1_act(o[j][c], p.act, p.slope);
        }
        if (JA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += av[j];
        }
        if (JX && mw < p.M) {
            // backward sums of the normalisation layer in front (conv_split.hip's store phase): one record per 32-row block and group
            const int bg = fd_div(mw, p.div_gl);
            const int gb = (bg + 1) * p.bn_L;
            const bool two = mw + 32 > gb && bg + 1 < p.bn_G;
            f32x4 mu0 = 0.f, rs0 = 0.f, mu1 = 0.f, rs1 = 0.f, ga = 1.f, be = 0.f;
            if (nok) {
                mu0 = *reinterpret_cast<const f32x4*>(p.bn_mean + (size_t)bg * p.Ng + n);
                rs0 = *reinterpret_cast<const f32x4*>(p.bn_rstd + (size_t)bg * p.Ng + n);
                if (two) {
                    mu1 = *reinterpret_cast<const f32x4*>(p.bn_mean + (size_t)(bg + 1) * p.Ng + n);
                    rs1 = *reinterpret_cast<const f32x4*>(p.bn_rstd + (size_t)(bg + 1) * p.Ng + n);
                }
                if (p.bn_gamma) { ga = *reinterpret_cast<const f32x4*>(p.bn_gamma + n); be = *reinterpret_cast<const f32x4*>(p.bn_beta + n); }
            }
            const int k0 = (mw >> 5) - (int)(((long)bg * p.bn_L) >> 5);
            double* r0 = p.bn_sums + (((size_t)bg * p.bn_chunks + k0) * p.Ng + n) * 2;
            double* r1 = p.bn_sums + ((size_t)(bg + 1) * p.bn_chunks * p.Ng + n) * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float sl = 0.f, ql = 0.f, sh = 0.f, qh = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = mw + rl + 8 * j;
                    if (m < p.M) {
                        const bool lo = m < gb;
                        const float xh = (xv[j][c] - (lo ? mu0[c] : mu1[c])) * (lo ? rs0[c] : rs1[c]);
                        const float ym = JZ ? zv[j][c] : xh * ga[c] + be[c];
                        float gg = o[j][c];
                        if (p.bn_act == SSCG_ACT_RELU) gg = ym > 0.f ? gg : 0.f;
                        else if (p.bn_act == SSCG_ACT_LRELU) gg = ym > 0.f ? gg : gg * p.bn_slope;
                        if (lo) { sl += gg; ql = fmaf(gg, xh, ql); } else { sh += gg; qh = fmaf(gg, xh, qh); }
                    }
                }
                const double a = lanes8((double)sl), b = lanes8((double)ql);
                if (rl == 0 && nok) { r0[c * 2] = a; r0[c * 2 + 1] = b; }
                if (two) {
                    const double cc = lanes8((double)sh), d = lanes8((double)qh);
                    if (rl == 0 && nok) { r1[c * 2] = cc; r1[c * 2 + 1] = d; }
                }
            }
        }
        // the next unit's first weight tile (requested one k-tile ago) must have landed BEFORE this unit's stores go out: a wait
        // behind them would wait for the stores as well
        if (more) {
            if (pre_a) {
                __syncthreads();            // every wave has read its last A fragments of this m-tile
                a_commit(mt_next);
            }
            g1_wait_vm<0>();
        }
        if (nok && (!(G1_ABLATE & 8) || u + 1 == u1)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mw + rl + 8 * j;
                if (m < p.M) *reinterpret_cast<f32x4*>(p.dst + (size_t)m * p.Ng + n) = o[j];
            }
        }
        if (more) {
            if (pre_a) __syncthreads(); else __builtin_amdgcn_s_barrier();      // tile (u + 1, 0) landed for all; the new pieces are visible
            mt_cur = mt_next;
        }
    }
}

bool g1_geometry_ok(long M, int Ng, int Kr, int tuning) {
    if (tuning & 0xffff) return false;              // a forced tile class / split plan names conv_split.hip's tiling
    static const bool off = getenv("SSCG_G1X1_OFF") != nullptr;                      // A/B aid: back to conv_split.hip's tiling
    return !off && (Kr == 128 || Kr == 256) && Ng % G1_BN == 0 && Ng >= 2 * Kr && M >= 4096 && M * (long)Ng < (1L << 31);
}

template <int MODE, int KR, int JA, int JX, int JZ>
int g1_launch_k(G1Params& p, hipStream_t st) {
    const int mtiles = cdiv(p.M, G1_BM);
    p.nchunks = p.Ng / G1_BN;
    p.units = mtiles * p.nchunks;
    p.div_nc = make_fastdiv(p.nchunks);
    const int grid = p.units < 256 ? p.units : 256;
    const size_t smem = (size_t)3 * G1_BM * p.Kr * 2 + 2 * G1_BSTAGE + 8 * 2048;      // K = 256: all 160 KB of the CU
    auto kern = g1x1_kernel<MODE, KR, JA, JX, JZ>;
    SSCG_ENSURE_SMEM((kern), smem);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G1_NT), smem, st, p);
    SSCG_LAUNCH_CHECK();
    return SSCG_OK;
}

template <int MODE, int JA, int JX, int JZ>
int g1_launch(G1Params& p, hipStream_t st) {
    return p.Kr == 256 ? g1_launch_k<MODE, 256, JA, JX, JZ>(p, st) : g1_launch_k<MODE, 128, JA, JX, JZ>(p, st);
}

}  // namespace

static bool g1_desc_ok(const sscg_conv_desc* d) { return d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->pad_mode == 0; }

bool sscg_g1x1_fwd_applies(const sscg_conv_desc* d) {
    return g1_desc_ok(d) && d->act != SSCG_ACT_TANH && g1_geometry_ok((long)d->N * d->P * d->Q, d->K, d->C, d->tuning);
}

bool sscg_g1x1_dgrad_applies(const sscg_conv_desc* d) { return g1_desc_ok(d) && g1_geometry_ok((long)d->N * d->H * d->W, d->C, d->K, d->tuning); }

int sscg_g1x1_fwd(const sscg_conv_desc* d, const void* x, const void* w, long wplane, const float* bias, void* y, double* stats, long stat_L,
                  hipStream_t st) {
    G1Params p = {};
    p.src = reinterpret_cast<const float*>(x); p.wgt = reinterpret_cast<const bf16*>(w); p.wplane = wplane;
    p.bias = bias; p.dst = reinterpret_cast<float*>(y);
    p.M = d->N * d->P * d->Q; p.Ng = d->K; p.Kr = d->C;
    p.act = d->act; p.slope = d->slope;
    p.wgt_bytes = (unsigned)(((size_t)2 * wplane + (size_t)d->K * d->C) * sizeof(bf16));
    p.stats = stats; p.stat_L = (int)stat_L;
    p.div_gl = make_fastdiv(stat_L > 0 ? (int)stat_L : 1);
    return g1_launch<G1_FWD, 0, 0, 0>(p, st);
}

int sscg_g1x1_dgrad(const sscg_conv_desc* d, const void* dy, const void* wt, long wplane, void* dx, hipStream_t st, const sscg_bsums* bs,
                    int chunks, const void* addend) {
    G1Params p = {};
    p.src = reinterpret_cast<const float*>(dy); p.wgt = reinterpret_cast<const bf16*>(wt); p.wplane = wplane;
    p.dst = reinterpret_cast<float*>(dx);
    p.M = d->N * d->H * d->W; p.Ng = d->C; p.Kr = d->K;
    p.act = SSCG_ACT_NONE;
    p.wgt_bytes = (unsigned)(((size_t)2 * wplane + (size_t)d->K * d->C) * sizeof(bf16));
    p.addend = reinterpret_cast<const float*>(addend);
    p.div_gl = make_fastdiv(1);
    if (bs) {
        p.bn_x = reinterpret_cast<const float*>(bs->nx); p.bn_z = reinterpret_cast<const float*>(bs->nz);
        p.bn_mean = bs->mean; p.bn_rstd = bs->rstd; p.bn_gamma = bs->gamma; p.bn_beta = bs->beta;
        p.bn_sums = reinterpret_cast<double*>(bs->sums); p.bn_L = (int)bs->L; p.bn_G = bs->G; p.bn_chunks = chunks;
        p.bn_act = bs->act; p.bn_slope = bs->slope;
        p.div_gl = make_fastdiv((int)bs->L);
    }
    const int sel = (addend ? 1 : 0) | (bs ? 2 : 0) | ((bs && bs->nz) ? 4 : 0);
    switch (sel) {
        case 0: return g1_launch<G1_DGRAD, 0, 0, 0>(p, st);
        case 1: return g1_launch<G1_DGRAD, 1, 0, 0>(p, st);
        case 2: return g1_launch<G1_DGRAD, 0, 1, 0>(p, st);
        case 3: return g1_launch<G1_DGRAD, 1, 1, 0>(p, st);
        case 6: return g1_launch<G1_DGRAD, 0, 1, 1>(p, st);
        case 7: return g1_launch<G1_DGRAD, 1, 1, 1>(p, st);
        default: return SSCG_ERR_BAD_ARG;
    }
}
