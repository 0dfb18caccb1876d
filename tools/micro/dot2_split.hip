// Micro-check on the MI355X (not part of the product): (1) the dot2-based residuals of common.h sscg_split8 equal the shift / mask /
// subtract form bit for bit over the whole fp32 range; (2) issue rate of v_dot2c_f32_bf16 against v_sub_f32.
// build: hipcc --offload-arch=gfx950 -O3 -I semi-supervised-segmentation-cyclegan_amd/csrc tools/micro/dot2_split.hip -o /tmp/dot2_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }

__global__ void split_both(const float* x, uint32_t* out, size_t n) {      // out[6][n/2]: h m l (dot2), h m l (arith)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    uint32_t sel_lo = 0x0000bf80u, sel_hi = 0xbf800000u;
    asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
    const bf16x2 nl = __builtin_bit_cast(bf16x2, sel_lo), nh = __builtin_bit_cast(bf16x2, sel_hi);
    const uint32_t hh = cvt_pk(a, b);
    const float ra = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, hh), nl, a, false);
    const float rb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, hh), nh, b, false);
    const uint32_t mm = cvt_pk(ra, rb);
    const float sa = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, mm), nl, ra, false);
    const float sb = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, mm), nh, rb, false);
    const uint32_t ll = cvt_pk(sa, sb);
    const size_t h = n / 2;
    out[i] = hh; out[h + i] = mm; out[2 * h + i] = ll;
    const float ra2 = a - __builtin_bit_cast(float, hh << 16), rb2 = b - __builtin_bit_cast(float, hh & 0xffff0000u);
    const uint32_t mm2 = cvt_pk(ra2, rb2);
    const float sa2 = ra2 - __builtin_bit_cast(float, mm2 << 16), sb2 = rb2 - __builtin_bit_cast(float, mm2 & 0xffff0000u);
    out[3 * h + i] = hh; out[4 * h + i] = mm2; out[5 * h + i] = cvt_pk(sa2, sb2);
}

template <int KIND> __global__ void rate(float* y, int iters) {
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = y[threadIdx.x + 64 * e];
    uint32_t sel = 0x0000bf80u;
    asm volatile("" : "+s"(sel));
    const bf16x2 nl = __builtin_bit_cast(bf16x2, sel);
    uint32_t hh = 0x3f803f80u;
    asm volatile("" : "+v"(hh));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (KIND == 0) v[e] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, hh), nl, v[e], false);
                else if (KIND == 1) v[e] = v[e] - __builtin_bit_cast(float, hh);
                else { const f32x2 t = {v[e], v[(e + 1) & 7]}; hh ^= __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2)); }
                asm volatile("" : "+v"(v[e]));
            }
    }
    if (KIND == 2) v[0] += __builtin_bit_cast(float, hh);
    for (int e = 0; e < 8; ++e) y[threadIdx.x + 64 * e] = v[e];
}

int main() {
    const size_t n = 1 << 24;
    std::vector<float> x(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) {
        uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand() ^ ((uint32_t)rand() << 31);
        if (i < n / 2) {                      // every exponent incl. denormals, no inf / nan
            if (((u >> 23) & 0xff) == 0xff) u &= ~(1u << 23);
        } else {                              // the range activations and gradients live in: 2^-40 .. 2^15
            u = (u & 0x807fffffu) | ((87u + (u >> 23) % 56u) << 23);
        }
        memcpy(&x[i], &u, 4);
    }
    float* dx; uint32_t* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 3 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    split_both<<<(unsigned)((n / 2 + 255) / 256), 256>>>(dx, dout, n);
    std::vector<uint32_t> o(3 * n);
    hipMemcpy(o.data(), dout, 3 * n * 4, hipMemcpyDeviceToHost);
    const size_t h = n / 2;
    size_t bad_all = 0, bad_rng = 0, bad_tiny = 0;
    for (size_t i = 0; i < h; ++i) {
        const bool bad = o[h + i] != o[4 * h + i] || o[2 * h + i] != o[5 * h + i];
        if (!bad) continue;
        const float a = x[2 * i], b = x[2 * i + 1];
        const bool tiny = fabsf(a) < 1e-30f || fabsf(b) < 1e-30f;
        if (2 * i >= n / 2) { ++bad_rng; if (bad_rng < 5) printf("  in-range mismatch: a %g b %g m %08x/%08x l %08x/%08x\n", a, b, o[h + i], o[4 * h + i], o[2 * h + i], o[5 * h + i]); }
        else if (tiny) ++bad_tiny;
        else { ++bad_all; if (bad_all < 5) printf("  full-range mismatch: a %g b %g m %08x/%08x l %08x/%08x\n", a, b, o[h + i], o[4 * h + i], o[2 * h + i], o[5 * h + i]); }
    }
    printf("pairs %zu: mismatching pairs - in [2^-40, 2^15]: %zu; full range, |x| >= 1e-30: %zu; |x| < 1e-30 (denormal residuals): %zu\n", h, bad_rng, bad_all, bad_tiny);

    float* y; hipMalloc(&y, 64 * 8 * 4 * 1024); hipMemset(y, 0, 64 * 8 * 4 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) rate<0><<<1024, 64>>>(y, iters); else if (kind == 1) rate<1><<<1024, 64>>>(y, iters); else rate<2><<<1024, 64>>>(y, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // 1024 waves on 1024 SIMDs: one wave per SIMD, 64 ops per iteration
            if (rep) printf("%s: %.3f ms for %d x 64 ops per wave = %.2f ns per op (4 cycles at 2.4 GHz = 1.67 ns)\n",
                            kind == 0 ? "v_dot2c_f32_bf16" : kind == 1 ? "v_sub_f32" : "v_cvt_pk_bf16_f32 (+xor)", ms, iters, ms * 1e6 / (iters * 64.0));
        }
    }
    return (bad_rng || bad_all) ? 1 : 0;
}
