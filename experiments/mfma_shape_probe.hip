// Probe (round 3): which MFMA shape carries an attention-like instruction mix furthest under the MI355X power cap when
// ONE wave owns a SIMD?  32x32x16 (32 cycles, 16 accumulator registers per lane) vs 16x16x32 (16 cycles, 4 registers:
// 8x less accumulator traffic per FLOP; a bare stream of it was ~10 % more power-efficient with 8 waves per CU, round 1).
// Per 65536 FLOP of matrix work (2 big / 4 small MFMAs) the mix issues FILL x {2 v_exp, 2 v_add, 1 v_cvt_pk} + 1
// ds_read_b128 — FILL = 1 is the zero-reference softmax of attn_hd128_w64 (2.5 VALU per 32x32x16-equivalent).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int SHAPE, int FILL, int WAVES, int RDS = 1>
__global__ __launch_bounds__(WAVES * 64, 1) void mix(const u32x4_t* __restrict__ in, float* __restrict__ out, int iters,
                                                     unsigned long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) char smem[96 * 1024];      // one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += WAVES * 64) ((u32x4_t*)smem)[i] = in[i];
    __syncthreads();
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i + 1777) & 4095]);
    }
    f32x16_t A32[4];
    f32x4_t A16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) A32[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) A16[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float x[4] = {0.001f * lane, -0.002f * lane, 0.5f, -0.25f}, y[4] = {0, 0, 0, 0}, s0 = 0.f, s1 = 0.f;
    unsigned pk = 0;
    const char* lp = smem + lane * 16;
    auto fill_a = [&](int k) __attribute__((always_inline)) {     // exp, add
        asm volatile("v_exp_f32 %0, %1" : "=v"(y[k & 3]) : "v"(x[k & 3]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(y[(k + 2) & 3]));
    };
    auto fill_b = [&](int k) __attribute__((always_inline)) {     // exp, add, cvt
        asm volatile("v_exp_f32 %0, %1" : "=v"(y[(k + 1) & 3]) : "v"(x[(k + 1) & 3]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(y[(k + 3) & 3]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(y[k & 3]), "v"(y[(k + 2) & 3]));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {         // 8 groups of 65536 FLOP
            if (SHAPE == 0) {
                A32[(2 * j) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[j & 3], A32[(2 * j) & 3], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f) fill_a(j + f);
                SB();
                A32[(2 * j + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j + 1) & 3], A32[(2 * j + 1) & 3], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f) fill_b(j + f);
            } else {
                A16[(4 * j) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j & 3], A16[(4 * j) & 15], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f) asm volatile("v_exp_f32 %0, %1" : "=v"(y[(j + f) & 3]) : "v"(x[(j + f) & 3]));
                SB();
                A16[(4 * j + 1) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j + 1) & 3], A16[(4 * j + 1) & 15], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(s0) : "v"(y[(j + f + 2) & 3]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1) : "v"(y[(j + f + 3) & 3]));
                }
                SB();
                A16[(4 * j + 2) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j + 2) & 3], A16[(4 * j + 2) & 15], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f) asm volatile("v_exp_f32 %0, %1" : "=v"(y[(j + f + 1) & 3]) : "v"(x[(j + f + 1) & 3]));
                SB();
                A16[(4 * j + 3) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j + 3) & 3], A16[(4 * j + 3) & 15], 0, 0, 0);
                SB();
#pragma unroll
                for (int f = 0; f < FILL; ++f)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(y[(j + f) & 3]), "v"(y[(j + f + 2) & 3]));
            }
            if (FILL >= 0) a[(j + 2) & 3] = *(const bf16x8_t*)(lp + ((j * 1024 + it * 8192) & 0xfc00));
            if (RDS >= 2) b[(j + 2) & 3] = *(const bf16x8_t*)(lp + ((j * 1024 + it * 8192 + 32768) & 0xfc00));   // 32-query waves: twice the fragment reads per MFMA
            SB();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = s0 + s1 + __uint_as_float(pk);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += A32[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += A16[i][0] + A16[i][1] + A16[i][2] + A16[i][3];
    out[blockIdx.x * WAVES * 64 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int FILL, int WAVES, int RDS = 1>
static void run(const u32x4_t* in, float* out, unsigned long long* cyc, int rounds) {
    const int iters = 20000, grid = 1024;
    for (int r = 0; r < rounds; ++r) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL((mix<SHAPE, FILL, WAVES, RDS>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, 2000, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix<SHAPE, FILL, WAVES, RDS>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, iters, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid);
        CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
        double cs = 0;
        for (auto v : h) cs += (double)v;
        cs /= grid;
        const double flops = (double)grid * WAVES * iters * 8 * 65536.0;
        printf("%s  waves/CU %d  fill x%d (VALU per 32x32x16-equivalent: %.1f) + %.1f ds_read: %8.1f TF/s  %6.1f cyc per 32768 FLOP  (clock ~%.2f GHz)\n",
               SHAPE ? "16x16x32" : "32x32x16", WAVES, FILL, FILL * 2.5, 0.5 * RDS, flops / (ms * 1e-3) / 1e12, cs / (iters * 16.0),
               cs * (grid / 256.0) / (ms * 1e-3) / 1e9 / 1.0);
        fflush(stdout);
    }
}

int main() {
    u32x4_t* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 65536)); CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    std::vector<unsigned short> h(32768);
    srand(1);
    for (auto& v : h) {
        float f = ((rand() % 2001) - 1000) / 1000.0f;
        unsigned u; memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    CK(hipMemcpy(in, h.data(), 65536, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {
        run<0, 0, 4>(in, out, cyc, 1);
        run<1, 0, 4>(in, out, cyc, 1);
        run<0, 1, 4>(in, out, cyc, 1);
        run<1, 1, 4>(in, out, cyc, 1);
        run<0, 2, 4>(in, out, cyc, 1);
        run<1, 2, 4>(in, out, cyc, 1);
        run<0, 1, 8>(in, out, cyc, 1);
        run<1, 1, 8>(in, out, cyc, 1);
        run<1, 1, 8, 2>(in, out, cyc, 1);
        run<1, 1, 4, 2>(in, out, cyc, 1);
    }
    return 0;
}
