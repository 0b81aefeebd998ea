// bf16 GEMM, variant 3: 256(token) x 256(feature) x 64(k) tile, 8 waves (2x4, 128 tokens x 64
// features each), TWO LDS stages of 64 KiB, LDS-DMA pieces issued one at a time between the MFMAs.
//
// Why (s_memtime breakdown of variant 2, `mg_selftest gemmprof`): per 64-deep k-tile the 8 waves of
// the 256x128 kernel issue 48 LDS-DMA pieces (48 KiB) for 1024 matrix-pipe cycles per SIMD; the
// CU's one vector-memory path moves ~64 B/clk, so the pieces alone cost ~770 cycles per k-tile and
// each costs the issuing wave 70-150 cycles.  A 256x256 tile moves 64 KiB per 2048 pipe cycles —
// 1/3 fewer bytes and DMA instructions per MFMA — and a wave tile of 128x64 needs 6 ds_read_b128
// per 8 MFMAs instead of 8.  LDS image, swizzle, MFMA roles and epilogues are those of gemm_bf16.hip.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define V3_BM 256
#define V3_BN 256
#define V3_BK 64
#define V3_THREADS 512
#define V3_A_BYTES (V3_BM * V3_BK * 2)  // 32 KiB
#define V3_W_BYTES (V3_BN * V3_BK * 2)  // 32 KiB
#define V3_STAGE (V3_A_BYTES + V3_W_BYTES)
#define V3_NSTAGE 2

typedef const __attribute__((address_space(1))) void* v3_gptr_t;
typedef __attribute__((address_space(3))) void* v3_lptr_t;
MG_DEV void v3_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v3_gptr_t)g, (v3_lptr_t)l, 16, 0, 0); }

static unsigned long long* g_gemm3_prof = nullptr;
extern "C" void mg_gemm3_debug_profile(unsigned long long* dev_buf) { g_gemm3_prof = dev_buf; }

template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V3_THREADS, 2) void gemm_bf16_v3_kernel(
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
    __shared__ __attribute__((aligned(16))) char smem[V3_NSTAGE * V3_STAGE];

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
    const int64_t m0 = (int64_t)tm * V3_BM;
    const int n0 = tn * V3_BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;     // 2 (token) x 4 (feature) waves, 128 x 64 each

    // DMA sources: wave w stages A rows [32w, 32w+32) and W rows [32w, 32w+32): 4 + 4 pieces of 8 rows
    const int srow = lane >> 3;
    const uint16_t* ga[4];
    const uint16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        int64_t am = m0 + row;
        if (am > M - 1) am = M - 1;
        ga[i] = A + am * lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        int wr = n0 + row;
        if (wr > N - 1) wr = N - 1;
        gw[i] = Wt + (int64_t)wr * ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* la = smem + (kt % V3_NSTAGE) * V3_STAGE + wave * 32 * 128;
        char* lw = smem + (kt % V3_NSTAGE) * V3_STAGE + V3_A_BYTES + wave * 32 * 128;
        const int koff = kt * V3_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) v3_glds16(ga[i] + koff, la + i * 8 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) v3_glds16(gw[i] + koff, lw + i * 8 * 128);
    };

    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;
    const int a_row_off = (wm * 128 + l31) * 128;
    const int w_row_off = V3_A_BYTES + (wn * 64 + l31) * 128;

    f32x16_t acc[2][4];      // [feature block][token block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / V3_BK;
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        tick(-1);
        // tile kt landed (every piece of it: two stages), and everyone is past compute(kt-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tick(0);
        const bool refill = kt + 1 < nk;
        char* la = smem + ((kt + 1) % V3_NSTAGE) * V3_STAGE + wave * 32 * 128;
        char* lw = smem + ((kt + 1) % V3_NSTAGE) * V3_STAGE + V3_A_BYTES + wave * 32 * 128;
        const int koff2 = (kt + 1) * V3_BK;
        tick(1);
        const char* ls = smem + (kt % V3_NSTAGE) * V3_STAGE;
        // fragment reads run ONE k-step ahead of the MFMAs that consume them (register double buffer)
        bf16x8_t fa[2][4], fw[2][2];
        {
            const int coff = t3 << 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[0][j] = *(const bf16x8_t*)(ls + a_row_off + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[0][i] = *(const bf16x8_t*)(ls + w_row_off + i * 32 * 128 + coff);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk < 3) {
                const int coff = (t3 ^ ((kk + 1) << 1)) << 4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    fa[(kk + 1) & 1][j] = *(const bf16x8_t*)(ls + a_row_off + j * 32 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fw[(kk + 1) & 1][i] = *(const bf16x8_t*)(ls + w_row_off + i * 32 * 128 + coff);
            }
            __builtin_amdgcn_sched_barrier(0);  // reads of step kk+1 issue BEFORE the MFMAs of step kk
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int j = 2 * h; j < 2 * h + 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
                    // the 8 LDS-DMA pieces of tile kt+1 go out behind every second MFMA of the FIRST half
                    // of this k-tile, so they have the second half (>1000 cycles) to land before the barrier
                    const int piece = kk * 4 + i * 2 + h;
                    if (piece < 8 && refill) {
                        if (piece < 4) v3_glds16(ga[piece] + koff2, la + piece * 8 * 128);
                        else v3_glds16(gw[piece - 4] + koff2, lw + (piece - 4) * 8 * 128);
                    }
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
    for (int j = 0; j < 4; ++j) {
        const int64_t m = m0 + wm * 128 + j * 32 + l31;
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

int mg_gemm_v3_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    const int64_t tiles_m64 = (M + V3_BM - 1) / V3_BM;
    const int tiles_n = (N + V3_BN - 1) / V3_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(V3_THREADS);
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16_v3_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, nullptr)
    if (g_gemm3_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v3_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm3_prof);
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
