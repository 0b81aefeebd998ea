// Issue-model probe for gfx950 (not product code): what does ONE SIMD sustain when two co-resident
// waves stream v_mfma_f32_32x32x16_bf16 with N VALU / transcendental / ds_read_b128 fillers per MFMA?
// Reports TFLOP/s (wall), shader cycles per MFMA per wave (s_memtime) and the effective clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/mfma_probe.hip -o moviigen1.1_amd/lib/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int NACC, int FILL, int TRANS, int LDSR, int WAVES, int LDSKB = 64>
__global__ __launch_bounds__(WAVES * 64, 2) void probe(const u32x4_t* __restrict__ in, float* __restrict__ out,
                                                      int iters, unsigned long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) char smem[LDSKB * 1024];   // 96 KiB: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += WAVES * 64) ((u32x4_t*)smem)[i] = in[i];
    __syncthreads();
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i + 1777) & 4095]);
    }
    f32x16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x[8], y[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = -0.01f * (lane + i);
    const float c = 0.999f;
    const char* lp = smem + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc[j % NACC], 0, 0, 0);
            if (LDSR) a[(j + 3) & 3] = *(const bf16x8_t*)(lp + ((j * 1024 + it * 16384) & 0xfc00));
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f]) : "v"(c));
#pragma unroll
            for (int t = 0; t < TRANS; ++t) asm volatile("v_exp_f32 %0, %0" : "+v"(y[t]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += y[i];
    out[blockIdx.x * (WAVES * 64) + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// bare v_mfma_f32_16x16x32_bf16 stream (same FLOPs per instruction as 32x32x16: 2*16*16*32 = 16384... half): which
// shape gives more TFLOP/s under the power cap?
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__global__ __launch_bounds__(512, 2) void probe_16(const u32x4_t* __restrict__ in, float* __restrict__ out, int iters,
                                                   unsigned long long* __restrict__ cyc) {
    const int tid = threadIdx.x;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i + 1777) & 4095]);
    }
    f32x4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j)
            acc[j & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[(j >> 2) & 3], acc[j & 7], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// one wave per SIMD, GEMM-like mix: SHAPE 0 = 16 x v_mfma 32x32x16 + 8 ds_read_b128 per iteration,
// SHAPE 1 = 32 x v_mfma 16x16x32 + 8 ds_read_b128 (same FLOPs, same LDS bytes)
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe_mix(const u32x4_t* __restrict__ in, float* __restrict__ out, int iters,
                                                    unsigned long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) char smem[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) ((u32x4_t*)smem)[i] = in[i];
    __syncthreads();
    bf16x8_t a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 8 + i) & 4095]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i + 1777) & 4095]);
    f32x16_t acc32[4];
    f32x4_t acc16[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* lp = smem + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SHAPE == 0) {
                acc32[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[j & 3], acc32[j & 3], 0, 0, 0);
                acc32[(j + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + 1) & 3], acc32[(j + 1) & 3], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc16[(j * 4 + u) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[u], acc16[(j * 4 + u) & 15], 0, 0, 0);
            }
            a[(j + 5) & 7] = *(const bf16x8_t*)(lp + ((j * 1024 + it * 8192) & 0xfc00));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc32[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    out[blockIdx.x * 256 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe_stage(const u32x4_t* __restrict__ in, float* __restrict__ out, int iters,
                                                      unsigned long long* __restrict__ cyc) {
    __shared__ __attribute__((aligned(16))) char smem[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) ((u32x4_t*)smem)[i] = in[i];
    __syncthreads();
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8_t, in[(tid * 4 + i + 1777) & 4095]);
    }
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = 0.001f * (lane + i);
    const float c = 0.999f;
    const char* lp = smem + lane * 16;
    const u32x4_t* gsrc = in + ((blockIdx.x * 64 + lane) & 2047);
    char* ldst = smem + 65536 + wave * 4096;
    u32x4_t stage = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[(j >> 2) & 3], acc[j & 3], 0, 0, 0);
            a[(j + 3) & 3] = *(const bf16x8_t*)(lp + ((j * 1024 + it * 16384) & 0xfc00));
#pragma unroll
            for (int f = 0; f < 4; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[f]) : "v"(c));
            if (MODE == 1 && (j & 7) == 3)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + ((it * 2 + (j >> 3)) & 31) * 64), (lptr_t)(ldst + (j >> 3) * 1024), 16, 0, 0);
            if (MODE == 2 && (j & 7) == 3) stage = gsrc[((it * 2 + (j >> 3)) & 31) * 64];
            if (MODE == 2 && (j & 7) == 7) *(u32x4_t*)(ldst + (j >> 3) * 1024 + lane * 16) = stage;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[i][e];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += x[i];
    r += smem[65536 + tid];
    out[blockIdx.x * 256 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC, int FILL, int TRANS, int LDSR, int WAVES, int LDSKB = 64>
void run(const u32x4_t* in, float* out, unsigned long long* cyc, int zero) {
    const int iters = 20000, grid = 256 * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<NACC, FILL, TRANS, LDSR, WAVES, LDSKB>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, 2000, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<NACC, FILL, TRANS, LDSR, WAVES, LDSKB>), dim3(grid), dim3(WAVES * 64), 0, 0, in, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double cs = 0;
    for (auto v : h) cs += (double)v;
    cs /= grid;
    const double flops = (double)grid * WAVES * iters * 16 * 32768.0;
    // each CU runs grid/256 workgroups back to back (one resident at 64 KiB LDS + launch bounds... two may fit)
    printf("acc=%d fill=%d trans=%d lds=%d waves=%d/CU %s: %8.1f TF/s  %6.1f cyc/MFMA/wave (memtime)  wall %.2f ms\n", NACC, FILL, TRANS,
           LDSR, LDSKB > 64 ? WAVES : 2 * WAVES > 8 ? 8 : 2 * WAVES, zero ? "zeros " : "random", flops / (ms * 1e-3) / 1e12, cs / (iters * 16.0), ms);
}

template <int MODE>
void run_stage(const u32x4_t* in, float* out, unsigned long long* cyc) {
    const int iters = 20000, grid = 256 * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe_stage<MODE>), dim3(grid), dim3(256), 0, 0, in, out, 2000, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_stage<MODE>), dim3(grid), dim3(256), 0, 0, in, out, iters, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
    double cs = 0;
    for (auto v : h) cs += (double)v;
    cs /= grid;
    printf("stage mode %d (0 none, 1 LDS-DMA, 2 global_load+ds_write; 1 KiB per 8 MFMAs): %8.1f TF/s  %6.1f cyc/MFMA  -> %.0f cyc per staged KiB\n",
           MODE, (double)grid * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12, cs / (iters * 16.0), MODE ? 0.0 : 0.0);
}

int main() {
    u32x4_t* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 65536)); CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    std::vector<unsigned short> h(32768);
    for (int z = 0; z < 2; ++z) {
        srand(1);
        if (z == 1) break;
        for (auto& v : h) {
            float f = z ? 0.f : ((rand() % 2001) - 1000) / 1000.0f;
            unsigned u; memcpy(&u, &f, 4);
            v = (unsigned short)(u >> 16);
        }
        CK(hipMemcpy(in, h.data(), 65536, hipMemcpyHostToDevice));
        CK(hipMemcpy(in, h.data(), 65536, hipMemcpyHostToDevice));
        {
            const int iters = 20000, grid = 1024;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(probe_16, dim3(grid), dim3(512), 0, 0, in, out, 2000, cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe_16, dim3(grid), dim3(512), 0, 0, in, out, iters, cyc);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("bare 16x16x32 bf16 MFMA, 8 waves/CU, random: %8.1f TF/s\n", (double)grid * 8 * iters * 32 * 16384.0 / (ms * 1e-3) / 1e12);
        }
        for (int shape = 0; shape < 2; ++shape) {
            const int iters = 20000, grid = 1024;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&](int n) {
                if (shape == 0) hipLaunchKernelGGL(probe_mix<0>, dim3(grid), dim3(256), 0, 0, in, out, n, cyc);
                else hipLaunchKernelGGL(probe_mix<1>, dim3(grid), dim3(256), 0, 0, in, out, n, cyc);
            };
            launch(2000);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(grid);
            CK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
            double cs = 0;
            for (auto v : h) cs += (double)v;
            cs /= grid;
            printf("GEMM-like mix, one wave per SIMD, %s + 8 ds_read_b128 per 524288 FLOP/wave: %8.1f TF/s  %.0f cycles per iteration\n",
                   shape ? "32 x mfma 16x16x32" : "16 x mfma 32x32x16", (double)grid * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12, cs / iters);
        }
        run<4, 0, 0, 0, 8>(in, out, cyc, 0);
        run_stage<0>(in, out, cyc);
        run_stage<1>(in, out, cyc);
        run_stage<2>(in, out, cyc);
        break;
        run<1, 0, 0, 0, 8>(in, out, cyc, z);
        run<2, 0, 0, 0, 8>(in, out, cyc, z);
        run<4, 0, 0, 1, 8>(in, out, cyc, z);
        run<4, 2, 0, 1, 8>(in, out, cyc, z);
        run<4, 4, 0, 1, 8>(in, out, cyc, z);
        run<4, 6, 0, 1, 8>(in, out, cyc, z);
        run<4, 8, 0, 1, 8>(in, out, cyc, z);
        run<4, 3, 1, 1, 8>(in, out, cyc, z);
        run<4, 4, 2, 1, 8>(in, out, cyc, z);
        run<2, 4, 2, 1, 8>(in, out, cyc, z);
        run<1, 4, 2, 1, 8>(in, out, cyc, z);
        run<4, 4, 2, 1, 4>(in, out, cyc, z);
        run<4, 8, 4, 1, 4>(in, out, cyc, z);
        // ONE wave per SIMD (4 waves per CU, 96 KiB LDS keeps a second workgroup out)
        run<4, 0, 0, 0, 4, 96>(in, out, cyc, z);
        run<4, 0, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 2, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 4, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 5, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 6, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 8, 0, 1, 4, 96>(in, out, cyc, z);
        run<4, 3, 1, 1, 4, 96>(in, out, cyc, z);
        run<4, 4, 2, 1, 4, 96>(in, out, cyc, z);
        run<4, 2, 2, 0, 4, 96>(in, out, cyc, z);
        run<1, 4, 0, 1, 4, 96>(in, out, cyc, z);
        run<2, 4, 0, 1, 4, 96>(in, out, cyc, z);
    }
    return 0;
}
