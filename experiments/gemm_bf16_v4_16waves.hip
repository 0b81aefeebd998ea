// NOT BUILT (archived experiment, round 1; measured 985-1010 TFLOP/s vs 1040-1078 for variant 3).
// bf16 GEMM, variant 4 (experiment): the 256 x 256 x 64 tile of variant 3 with SIXTEEN waves (4x4, 64 x 64
// each, 128 VGPRs) = four waves per SIMD: more waves to cover each other's LDS-DMA issue stalls, at
// the price of one fragment read per MFMA instead of 0.75.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define V4_BM 256
#define V4_BN 256
#define V4_BK 64
#define V4_THREADS 1024
#define V4_A_BYTES (V4_BM * V4_BK * 2)  // 32 KiB
#define V4_W_BYTES (V4_BN * V4_BK * 2)  // 32 KiB
#define V4_STAGE (V4_A_BYTES + V4_W_BYTES)
#define V4_NSTAGE 2
#ifndef V4_PPS
#define V4_PPS 1   // LDS-DMA pieces per slot: with 8 pieces per wave they all go out in the first half of the k-tile, the second half is their time to land
#endif

typedef const __attribute__((address_space(1))) void* v4_gptr_t;
typedef __attribute__((address_space(3))) void* v4_lptr_t;
MG_DEV void v4_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v4_gptr_t)g, (v4_lptr_t)l, 16, 0, 0); }

static unsigned long long* g_gemm4_prof = nullptr;
extern "C" void mg_gemm4_debug_profile(unsigned long long* dev_buf) { g_gemm4_prof = dev_buf; }

template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V4_THREADS, 4) void gemm_bf16_v4_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, unsigned long long* __restrict__ prof) {
    unsigned long long pt[4] = {0, 0, 0, 0}, pc = 0;
    auto tick = [&](int i) __attribute__((always_inline)) {
        if (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (i >= 0) pt[i] += now - pc;
            pc = now;
        }
    };
    __shared__ __attribute__((aligned(16))) char smem[V4_NSTAGE * V4_STAGE];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GM = 4;  // 4 x 256 = the same 1024-token band as variant 1
    const int per_group = GM * tiles_n;
    const int group = swz / per_group;
    const int first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = swz - group * per_group;
    const int tm = first_m + in_g % gsz;
    const int tn = in_g / gsz;
    const int64_t m0 = (int64_t)tm * V4_BM;
    const int n0 = tn * V4_BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;     // 4 (token) x 4 (feature) waves, 64 x 64 each

    // DMA sources: wave w stages A rows [32w, 32w+32) and W rows [32w, 32w+32): 4 + 4 pieces of 8 rows
    const int srow = lane >> 3;
    // LDS-DMA duty: wave w stages A rows [16w, 16w+16) and W rows [16w, 16w+16): 2 + 2 pieces per k-tile
    constexpr int NPMAX = 4;
    const int np = 4;
    const uint16_t* gp[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
        const int row = wave * 16 + (i & 1) * 8 + srow;
        if (i < 2) {
            int64_t am = m0 + row;
            if (am > M - 1) am = M - 1;
            gp[i] = A + am * lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        } else {
            int wr = n0 + row;
            if (wr > N - 1) wr = N - 1;
            gp[i] = Wt + (int64_t)wr * ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        }
    }
    auto piece_lds = [&](int q) __attribute__((always_inline)) {
        return (q < 2 ? 0 : V4_A_BYTES) + (wave * 16 + (q & 1) * 8) * 128;
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* base = smem + (kt % V4_NSTAGE) * V4_STAGE;
        const int koff = kt * V4_BK;
#pragma unroll
        for (int i = 0; i < NPMAX; ++i) v4_glds16(gp[i] + koff, base + piece_lds(i));
    };

    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;
    const int a_row_off = (wm * 64 + l31) * 128;
    const int w_row_off = V4_A_BYTES + (wn * 64 + l31) * 128;

    f32x16_t acc[2][2];      // [feature block][token block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / V4_BK;
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        tick(-1);
        // tile kt landed (every piece of it: two stages), and everyone is past compute(kt-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tick(0);
        const bool refill = kt + 1 < nk;
        char* lnext = smem + ((kt + 1) % V4_NSTAGE) * V4_STAGE;
        const int koff2 = (kt + 1) * V4_BK;
        tick(1);
        const char* ls = smem + (kt % V4_NSTAGE) * V4_STAGE;
        // fragment reads run ONE k-step ahead of the MFMAs that consume them (register double buffer)
        bf16x8_t fa[2][2], fw[2][2];
        {
            const int coff = t3 << 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[0][j] = *(const bf16x8_t*)(ls + a_row_off + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[0][i] = *(const bf16x8_t*)(ls + w_row_off + i * 32 * 128 + coff);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int coff = (t3 ^ ((kk + 1) << 1)) << 4;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fa[(kk + 1) & 1][j] = *(const bf16x8_t*)(ls + a_row_off + j * 32 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fw[(kk + 1) & 1][i] = *(const bf16x8_t*)(ls + w_row_off + i * 32 * 128 + coff);
            }
            __builtin_amdgcn_sched_barrier(0);  // reads of step kk+1 issue BEFORE the MFMAs of step kk
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
                const int q = kk * 2 + i;           // one LDS-DMA piece behind every second MFMA of the first half
                if (refill && q < np) v4_glds16(gp[q] + koff2, lnext + piece_lds(q));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        tick(2);
        if (PROF) pt[3] += 1;
    }
    if (PROF && lane == 0 && prof) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(prof + wave * 4 + i, pt[i]);
    }

    // ---- epilogue (identical to gemm_bf16.hip): lane owns token row m, 4 features per quad -----------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t m = m0 + wm * 64 + j * 32 + l31;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int n = n0 + wn * 64 + i * 32 + rq * 8 + g * 4;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rq * 4 + e];
                const bool full = (n + 3 < N);
                if (bias) {
                    if (full) {
                        const float4 b4 = *(const float4*)(bias + n);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < N) v[e] += bias[n + e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);
                if (EPI == MG_EPI_BIAS_GELU_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
                }
                if (EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16) {
                    uint16_t* o = (uint16_t*)out + m * ldo + n;
                    if (full) {
                        uint2 p;
                        p.x = pack_bf2(v[0], v[1]);
                        p.y = pack_bf2(v[2], v[3]);
                        *(uint2*)o = p;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < N) o[e] = f2bf(v[e]);
                    }
                } else {
                    float* o = (float*)out + m * ldo + n;
                    if (EPI == MG_EPI_GATE_RESID_F32) {
                        if (full) {
                            float4 gg = gate ? *(const float4*)(gate + n) : make_float4(1.f, 1.f, 1.f, 1.f);
                            float4 x4 = *(float4*)o;
                            x4.x += v[0] * gg.x; x4.y += v[1] * gg.y; x4.z += v[2] * gg.z; x4.w += v[3] * gg.w;
                            *(float4*)o = x4;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < N) o[e] += v[e] * (gate ? gate[n + e] : 1.f);
                        }
                    } else {
                        if (full) {
                            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < N) o[e] = v[e];
                        }
                    }
                }
            }
        }
    }
}

int mg_gemm_v4_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    const int64_t tiles_m64 = (M + V4_BM - 1) / V4_BM;
    const int tiles_n = (N + V4_BN - 1) / V4_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(V4_THREADS);
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16_v4_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, nullptr)
    if (g_gemm4_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v4_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm4_prof);
        return mg_check_launch();
    }
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
