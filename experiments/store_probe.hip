// Which store / load-modify-store shape does the memory pipe of a gfx950 CU like?  The epilogue of the 256 x 256 GEMM tile costs
// ~16 k of a tile's 238 k cycles (profiles/r04p_gemm_v11.log); this probe writes the SAME output tiles (4 waves x 128 x 128 per
// workgroup, one workgroup per CU, 40 tiles per workgroup, rows `ld` bytes apart) with nothing else going on, in the lane -> address
// shapes an epilogue can produce:
//   bf16 (256 bytes per wave row):  0 = 8 B per lane, 16 rows x 32 B per instruction (the MFMA's own layout)
//                                   1 = 16 B per lane, 16 rows x 64 B (two feature blocks paired)
//                                   2 = 16 B per lane, 4 rows x 256 B (whole row segments, as after a transposition)
//   fp32 (512 bytes per wave row):  3 = 16 B per lane, 16 rows x 64 B (the MFMA's own layout)
//                                   4 = 2 x 16 B per lane at 32 B stride, 16 rows x 128 B per instruction pair
//                                   5 = 16 B per lane, 2 rows x 512 B (whole row segments)
//   6 / 7 / 8 = 3 / 4 / 5 with the residual load in front of every store (x + y)
// build: hipcc --offload-arch=gfx950 -O3 experiments/store_probe.hip -o moviigen1.1_amd/lib/store_probe
// run:   store_probe [workgroups=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int P>
__global__ __launch_bounds__(256, 1) void probe(char* __restrict__ out, long ld, int tiles_m, int total) {
    extern __shared__ char lds_pad[];      // 128 KiB: one workgroup per CU, as the GEMM
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int ES = P <= 2 ? 2 : 4;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int tm = t % tiles_m, tn = t / tiles_m;
        char* base = out + (long)(tm * 256 + wm * 128) * ld + (long)(tn * 256 + wn * 128) * ES;
        const unsigned val = (unsigned)(t * 64 + lane);
        if (P == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) *(uint2*)(base + (long)(j * 16 + r16) * ld + c * 32 + G * 8) = make_uint2(val, val + c);
        } else if (P == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *(uint4*)(base + (long)(j * 16 + r16) * ld + p * 64 + G * 16) = make_uint4(val, val + p, val + j, val);
        } else if (P == 2) {
#pragma unroll
            for (int q = 0; q < 32; ++q)
                *(uint4*)(base + (long)(q * 4 + (lane >> 4)) * ld + (lane & 15) * 16) = make_uint4(val, val + q, val, val);
        } else if (P == 3 || P == 6) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                u4 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[j] = (u4){val, val + c, val + j, val};
                    if (P == 6) x[j] = *(const u4*)(base + (long)(j * 16 + r16) * ld + c * 64 + G * 16);
                }
                if (P == 6) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) :: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[j].x += val;
                    *(u4*)(base + (long)(j * 16 + r16) * ld + c * 64 + G * 16) = x[j];
                }
            }
        } else if (P == 4 || P == 7) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    u4 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = jh * 4 + (u >> 1), h = u & 1;
                        x[u] = (u4){val, val + p, val + u, val};
                        if (P == 7) x[u] = *(const u4*)(base + (long)(j * 16 + r16) * ld + p * 128 + G * 32 + h * 16);
                    }
                    if (P == 7) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) :: "memory");
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = jh * 4 + (u >> 1), h = u & 1;
                        x[u].x += val;
                        *(u4*)(base + (long)(j * 16 + r16) * ld + p * 128 + G * 32 + h * 16) = x[u];
                    }
                }
        } else {
#pragma unroll
            for (int qb = 0; qb < 64; qb += 8) {
                u4 x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u] = (u4){val, val + qb, val + u, val};
                    if (P == 8) x[u] = *(const u4*)(base + (long)((qb + u) * 2 + (lane >> 5)) * ld + (lane & 31) * 16);
                }
                if (P == 8) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) :: "memory");
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    x[u].x += val;
                    *(u4*)(base + (long)((qb + u) * 2 + (lane >> 5)) * ld + (lane & 31) * 16) = x[u];
                }
            }
        }
    }
}

template <int P>
static void run(char* buf, int nwg) {
    constexpr int ES = P <= 2 ? 2 : 4;
    const int tiles_m = 512, tiles_n = 20;          // 131 072 x 5120
    const long ld = 5120L * ES;
    CK(hipFuncSetAttribute((const void*)probe<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // the same tiles per workgroup whatever the grid: 40 each
    const int total = nwg * 40;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(probe<P>, dim3(nwg), dim3(256), 128 * 1024, 0, buf, ld, tiles_m, total);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<P>, dim3(nwg), dim3(256), 128 * 1024, 0, buf, ld, tiles_m, total);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)total * 256 * 256 * ES * (P >= 6 ? 2 : 1);
    printf("shape %d  %3d workgroups: %8.3f ms  %7.1f GB/s  %6.2f us per tile  %5.1f bytes/ns per CU\n", P, nwg, ms, bytes / ms * 1e-6,
           ms * 1e3 / 40, bytes / nwg / (ms * 1e6));
}

int main(int argc, char** argv) {
    char* buf;
    CK(hipMalloc(&buf, 131072L * 5120 * 4));
    CK(hipMemset(buf, 0, 131072L * 5120 * 4));
    for (int nwg : {256, 64, 8}) {
        if (argc > 1 && atoi(argv[1]) != nwg) continue;
        run<0>(buf, nwg); run<1>(buf, nwg); run<2>(buf, nwg); run<3>(buf, nwg); run<4>(buf, nwg); run<5>(buf, nwg);
        run<6>(buf, nwg); run<7>(buf, nwg); run<8>(buf, nwg);
    }
    return 0;
}
