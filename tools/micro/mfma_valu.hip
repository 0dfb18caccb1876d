// Micro-check (not product): how many VALU operations of a kind issue under a v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD?
// kinds: 0 v_sub_f32, 1 v_dot2c_f32_bf16, 2 v_cvt_pk_bf16_f32, 3 v_and_b32, 4 v_perm_b32, 5 v_pk_add_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int NV> __global__ __launch_bounds__(64) void k(float* y, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = y[threadIdx.x + 64 * e];
    uint32_t sel = 0x0000bf80u; asm volatile("" : "+s"(sel));
    uint32_t hh = 0x3f803f80u; asm volatile("" : "+v"(hh));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int e = (r * NV + q) & 7;
                if (KIND == 0) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[e]) : "v"(hh));
                else if (KIND == 1) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[e]) : "s"(sel), "v"(hh));
                else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[e]) : "v"(hh));
                else if (KIND == 3) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[e]) : "v"(hh));
                else if (KIND == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(hh), "s"(sel));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
    for (int e = 0; e < 8; ++e) y[threadIdx.x + 64 * e] = v[e] + t;
}

template <int KIND, int NV> void run(float* y, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<KIND, NV><<<1024, 64>>>(y, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-20s %d per MFMA: %.1f ns per MFMA\n", name, NV, ms * 1e6 / (iters * 8.0));
}
template <int KIND> void sweep(float* y, const char* name) {
    run<KIND, 0>(y, name); run<KIND, 2>(y, name); run<KIND, 4>(y, name); run<KIND, 5>(y, name); run<KIND, 6>(y, name); run<KIND, 7>(y, name); run<KIND, 8>(y, name);
}
int main() {
    float* y; hipMalloc(&y, 64 * 8 * 4 * 1024); hipMemset(y, 0, 64 * 8 * 4 * 1024);
    sweep<0>(y, "v_sub_f32"); sweep<1>(y, "v_dot2c_f32_bf16"); sweep<2>(y, "v_cvt_pk_bf16_f32"); sweep<3>(y, "v_and_b32"); sweep<4>(y, "v_perm_b32");
    return 0;
}
