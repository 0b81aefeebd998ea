// bf16 GEMM, variant 2: 256(token) x 128(feature) x 64(k) tile, 8 waves (4x2, 64x64 each),
// THREE LDS stages (3 x 48 KiB = 144 KiB, one workgroup per CU) with counted vmcnt + raw barriers.
//
// Why (profiles/r01_pmc_gemm.txt, variant 1 = gemm_bf16.hip): matrix pipe 47.6 % busy, 52 % of
// wave time waiting on dependencies / the pipe — with two stages the LDS-DMA of tile t+1 must land
// within ONE k-tile of MFMA time (~1000 cycles/SIMD), about the L2 round trip.  Here tile t+2 is
// issued right after the barrier that publishes tile t, so every DMA has two k-tiles to land and
// the wait in front of the barrier is `s_waitcnt vmcnt(6)` (this wave's 6 pieces of tile t+1 may
// stay in flight), never a full drain.  __syncthreads() would re-insert vmcnt(0) (hipcc drains
// LDS-DMA at its fences), hence __builtin_amdgcn_s_barrier().
// LDS image, swizzle, MFMA roles and epilogues are those of gemm_bf16.hip.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V2_BM 256
#define V2_BN 128
#define V2_BK 64
#define V2_THREADS 512
#define V2_A_BYTES (V2_BM * V2_BK * 2)  // 32 KiB
#define V2_W_BYTES (V2_BN * V2_BK * 2)  // 16 KiB
#define V2_STAGE (V2_A_BYTES + V2_W_BYTES)
#define V2_NSTAGE 3

typedef const __attribute__((address_space(1))) void* v2_gptr_t;
typedef __attribute__((address_space(3))) void* v2_lptr_t;
MG_DEV void v2_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v2_gptr_t)g, (v2_lptr_t)l, 16, 0, 0); }

static unsigned long long* g_gemm_prof = nullptr;
#ifdef MG_AB_BUILD
// s_memtime hook (A/B library only): 8 waves x 4 sums {wait+barrier, stage issue, MFMA segment, k-tiles}
extern "C" void mg_gemm_debug_profile(unsigned long long* dev_buf) { g_gemm_prof = dev_buf; }
#endif

template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V2_THREADS, 2) void gemm_bf16_v2_kernel(
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
    __shared__ __attribute__((aligned(16))) char smem[V2_NSTAGE * V2_STAGE];

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
    const int64_t m0 = (int64_t)tm * V2_BM;
    const int n0 = tn * V2_BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // DMA sources: wave w stages A rows [32w, 32w+32) (4 instructions of 8 rows) and W rows [16w, 16w+16) (2)
    const int srow = lane >> 3;
    const uint16_t* ga[4];
    const uint16_t* gw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        int64_t am = m0 + row;
        if (am > M - 1) am = M - 1;
        ga[i] = A + am * lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + srow;
        int wr = n0 + row;
        if (wr > N - 1) wr = N - 1;
        gw[i] = Wt + (int64_t)wr * ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* la = smem + (kt % V2_NSTAGE) * V2_STAGE + wave * 32 * 128;
        char* lw = smem + (kt % V2_NSTAGE) * V2_STAGE + V2_A_BYTES + wave * 16 * 128;
        const int koff = kt * V2_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) v2_glds16(ga[i] + koff, la + i * 8 * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) v2_glds16(gw[i] + koff, lw + i * 8 * 128);
    };

    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;
    const int a_row_off = (wm * 64 + l31) * 128;
    const int w_row_off = V2_A_BYTES + (wn * 64 + l31) * 128;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / V2_BK;
    stage(0);
    if (nk > 1) stage(1);
    for (int kt = 0; kt < nk; ++kt) {
        tick(-1);
        // tile kt landed (this wave's 6 pieces of tile kt+1 may still be in flight)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tick(0);
        // everyone is past compute(kt-1): its stage is free for tile kt+2.  The 6 LDS-DMA pieces of
        // that tile are NOT issued here in a block: one global_load_lds costs the issuing wave
        // 70-150 cycles (48 of them at once from 8 waves: 430-930 cycles with the matrix pipe idle,
        // measured with s_memtime) — they are issued one at a time between the MFMAs below.
        const bool refill = kt + 2 < nk;
        char* la = smem + ((kt + 2) % V2_NSTAGE) * V2_STAGE + wave * 32 * 128;
        char* lw = smem + ((kt + 2) % V2_NSTAGE) * V2_STAGE + V2_A_BYTES + wave * 16 * 128;
        const int koff2 = (kt + 2) * V2_BK;
        tick(1);
        const char* ls = smem + (kt % V2_NSTAGE) * V2_STAGE;
        // fragment reads run ONE k-step ahead of the MFMAs that consume them (register double
        // buffer): the 4 ds_read_b128 of step kk+1 are in flight under the 4 MFMAs (128 pipe cycles)
        // of step kk instead of being waited for with lgkmcnt(0) right before them.
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
                const int piece = kk * 2 + i;      // one DMA piece behind every second MFMA (6 pieces, steps 0..2)
                if (piece < 6 && refill) {
                    if (piece < 4) v2_glds16(ga[piece] + koff2, la + piece * 8 * 128);
                    else v2_glds16(gw[piece - 4] + koff2, lw + (piece - 4) * 8 * 128);
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

    // ---- epilogue (gemm_epilogue.h): lane owns token row m, 4 features per accumulator quad; batched loads ----
    mg_gemm_epilogue<EPI, 2, 2>(acc, m0 + wm * 64, n0 + wn * 64, l31, g, M, N, bias, gate, out, ldo);
}

int mg_gemm_v2_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    const int64_t tiles_m64 = (M + V2_BM - 1) / V2_BM;
    const int tiles_n = (N + V2_BN - 1) / V2_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(V2_THREADS);
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16_v2_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, nullptr)
    if (g_gemm_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v2_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm_prof);
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
