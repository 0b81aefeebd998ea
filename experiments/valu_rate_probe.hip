// Probe (round 3): issue cost of the VALU instructions a softmax can be built from, on gfx950, one wave per SIMD,
// 8 independent chains each (cycles per wave-instruction by s_memtime ticks scaled with a v_add_f32 reference).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int OP>
__global__ __launch_bounds__(256, 1) void rate(float* out, unsigned long long* cyc, int iters) {
    float x[8];
    f32x2_t y[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = (f32x2_t){x[i], -x[i]}; u[i] = 0x3f803f80u + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
                if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 1) & 7]));
                if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
                if (OP == 7) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
                if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(y[(i + 1) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i] + y[i][0] + y[i][1] + __uint_as_float(u[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static double run(const char* name, float* out, unsigned long long* cyc, double ref) {
    const int iters = 20000, grid = 256;
    hipLaunchKernelGGL((rate<OP>), dim3(grid), dim3(256), 0, 0, out, cyc, 200);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<OP>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)iters * 32);
    printf("%-22s %7.3f ns per wave-instruction  (%.2f x v_add_f32)\n", name, ns, ref > 0 ? ns / ref : 1.0);
    return ns;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    double ref = run<0>("v_add_f32", out, cyc, 0);
    run<0>("v_add_f32", out, cyc, ref);
    run<6>("v_fma_f32", out, cyc, ref);
    run<1>("v_pk_add_f32", out, cyc, ref);
    run<5>("v_pk_mul_f32", out, cyc, ref);
    run<9>("v_pk_fma_f32", out, cyc, ref);
    run<2>("v_dot2c_f32_bf16", out, cyc, ref);
    run<7>("v_dot2_f32_bf16", out, cyc, ref);
    run<3>("v_exp_f32", out, cyc, ref);
    run<4>("v_cvt_pk_bf16_f32", out, cyc, ref);
    run<8>("v_max3_f32", out, cyc, ref);
    return 0;
}
