// Shared device/host helpers for the sscg HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// Dry runs (sscg_set_dry_run): every kernel launch of the library becomes a no-op while the host side of each entry point - argument
// checks, planning, workspace carving, descriptor set-up - runs as usual.  bench.py uses it to time the host's issue cost of a step
// with no back-pressure from the device ("host_issue_unblocked_ms").  One flag per process, read at every launch site.
extern int g_sscg_dry_run;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                   \
    do {                                                                                                                     \
        if (!g_sscg_dry_run) { (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__); }        \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = h + m + l exactly to 2^-24 |x|: three bfloat16 pieces of an fp32 value (round to nearest each time; the residuals are
// exact in fp32).  The "split" contraction mode multiplies the pieces on the bf16 matrix cores: a0*b0 + a0*b1 + a1*b0 + a0*b2 +
// a1*b1 + a2*b0 (products of 8-bit mantissas are exact, the accumulation is fp32) - the terms left out are below 2^-24 |a*b|.
struct sscg_bf3 { __bf16 h, m, l; };
__device__ __forceinline__ sscg_bf3 sscg_split3(float x) {
    sscg_bf3 t;
    t.h = (__bf16)x;
    const float r1 = x - (float)t.h;
    t.m = (__bf16)r1;
    const float r2 = r1 - (float)t.m;
    t.l = (__bf16)r2;
    return t;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// The same split for eight values at once, in packed form: three bf16x8 MFMA operands from eight fp32 (v_cvt_pk_bf16_f32 rounds a
// pair to nearest; the fp32 value of a piece is its 16 bits shifted up).  11 VALU operations per pair of elements.
typedef uint32_t sscg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t sscg_cvt_pk_bf16(float a, float b) {      // low half = bf16(a), high half = bf16(b)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void sscg_split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    sscg_u32x4 ph, pm, pl;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float a = x[2 * w], b = x[2 * w + 1];
        const uint32_t hh = sscg_cvt_pk_bf16(a, b);
        const float ra = a - __builtin_bit_cast(float, hh << 16), rb = b - __builtin_bit_cast(float, hh & 0xffff0000u);
        const uint32_t mm = sscg_cvt_pk_bf16(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, mm << 16), sb = rb - __builtin_bit_cast(float, mm & 0xffff0000u);
        ph[w] = hh; pm[w] = mm; pl[w] = sscg_cvt_pk_bf16(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, ph);
    m = __builtin_bit_cast(bf16x8, pm);
    l = __builtin_bit_cast(bf16x8, pl);
}

// ---- element access for tensors that are fp32 or bfloat16 in HBM (arithmetic is always fp32)
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }
// four consecutive elements (16 bytes of fp32 / 8 bytes of bf16; the address must be aligned to that)
template <typename T> __device__ __forceinline__ void ld4(const T* p, float v[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float v[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void ld4<__bf16>(const __bf16* p, float v[4]) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float v[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float v[4]) {
    const f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
}
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, const float v[4]) {
    const bf16x4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
}
// eight consecutive elements (bf16: one 16-byte access)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ void ld8(const T* p, float v[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float v[8]) { ld4<float>(p, v); ld4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void ld8<__bf16>(const __bf16* p, float v[8]) {
    const bf16x8_t t = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float v[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float v[8]) { st4<float>(p, v); st4<float>(p + 4, v + 4); }
template <> __device__ __forceinline__ void st8<__bf16>(__bf16* p, const float v[8]) {
    bf16x8_t t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8_t*>(p) = t;
}

#define SSCG_OK 0
#define SSCG_ERR_BAD_ARG (-1)
#define SSCG_ERR_UNSUPPORTED (-2)
#define SSCG_ERR_WORKSPACE (-3)

// Every entry point returns 0 or a negative library code / positive hipError_t.
#define SSCG_LAUNCH_CHECK()                        \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Activation codes shared by conv epilogues and norm kernels.
enum { SSCG_ACT_NONE = 0, SSCG_ACT_RELU = 1, SSCG_ACT_LRELU = 2, SSCG_ACT_TANH = 3 };

__device__ __forceinline__ float sscg_act(float v, int act, float slope) {
    if (act == SSCG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == SSCG_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == SSCG_ACT_TANH) return tanhf(v);
    return v;
}

// Division by a runtime-invariant divisor (dividend < 2^31).
struct FastDiv {
    uint32_t mul;
    uint32_t shift;
    int32_t d;
};

static inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1u << l) < (uint32_t)d) ++l;
    f.shift = l;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - (uint64_t)d)) / (uint64_t)d + 1);
    return f;
}

__device__ __forceinline__ int fd_div(int n, const FastDiv& f) {
    return (int)((__umulhi((uint32_t)n, f.mul) + (uint32_t)n) >> f.shift);
}

// XCD-aware workgroup remap (bijective): the dispatcher places block b on XCD b % 8 (observed, speed only);
// give every XCD a contiguous chunk of the logical tile space so that neighbouring tiles share its private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// wave64 all-lane sum (double) through ds_bpermute-free shuffles
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel that wants more than 64 KB of dynamic LDS: set when the request grows, not on
// every launch (the call costs the host 1-2 us; the step issues ~1000 such launches).  One static table per call site = per template
// instantiation, one entry per DEVICE (the attribute is per device: a process that launches on a second GPU - a test, a --gpu_ids
// switch - must set it there too); two threads racing here set the same value twice.
#define SSCG_ENSURE_SMEM(kern, smem)                                                                                                   \
    do {                                                                                                                               \
        static size_t sscg_attr_smem_[16] = {0};                                                                                       \
        int sscg_dev_ = 0;                                                                                                             \
        if (hipGetDevice(&sscg_dev_) != hipSuccess) sscg_dev_ = 0;                                                                     \
        size_t& sscg_cur_ = sscg_attr_smem_[sscg_dev_ & 15];                                                                           \
        if (sscg_cur_ == 0) sscg_cur_ = 64 * 1024;                                                                                     \
        if ((size_t)(smem) > sscg_cur_) {                                                                                              \
            hipError_t sscg_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)(smem));                                                                     \
            if (sscg_e_ != hipSuccess) return (int)sscg_e_;                                                                            \
            sscg_cur_ = (size_t)(smem);                                                                                                \
        }                                                                                                                              \
    } while (0)

// A BM x BN fp32 result tile held in MFMA accumulators (wave (wm, wn) owns TM x TN blocks of 32 x 32; a lane owns single elements of
// 16 rows per block) leaves through LDS: staged as [row][BN + 4] floats - `ot`, BM * (BN + 4) * 4 bytes, the k-loop's LDS is dead -
// every thread then writes 16 bytes = four consecutive columns of a row: 4x fewer store instructions than the four-byte stores of the
// MFMA layout, each a full row segment.  Ng % 4 == 0.  beta != 0: out = tile + beta * out.  Rows >= Mlim / columns >= Ng are dropped.
template <int BM, int BN, int NT, int TM, int TN>
__device__ __forceinline__ void sscg_stage_store_tile(const f32x16 (&acc)[TM][TN], float* ot, float* __restrict__ out, int m0, int n0, int Mlim,
                                                      int Ng, int row_w, int col_w, int li, int lh, int tid, float beta) {
    constexpr int OLD = BN + 4;
    __syncthreads();                        // every wave has read its last fragments
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                ot[(row_w + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * OLD + col_w + j * 32 + li] = acc[i][j][e];
    __syncthreads();
    constexpr int TPR = BN / 4;             // threads per row
    constexpr int RPP = NT / TPR;           // rows per pass
    const int c4 = (tid % TPR) * 4;
    const int n = n0 + c4;
    if (n >= Ng) return;
#pragma unroll
    for (int ps = 0; ps < BM / RPP; ++ps) {
        const int r = tid / TPR + ps * RPP;
        const int m = m0 + r;
        if (m >= Mlim) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(ot + r * OLD + c4);
        float* o = out + (size_t)m * Ng + n;
        if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
}
