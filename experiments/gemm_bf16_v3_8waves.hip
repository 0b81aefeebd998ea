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
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V3_BM 256
#define V3_BN 256
#define V3_BK 64
#define V3_THREADS 512
#define V3_A_BYTES (V3_BM * V3_BK * 2)  // 32 KiB
#define V3_W_BYTES (V3_BN * V3_BK * 2)  // 32 KiB
#define V3_STAGE (V3_A_BYTES + V3_W_BYTES)
#define V3_NSTAGE 2
#ifndef V3_PPS
#define V3_PPS 1   // LDS-DMA pieces per slot: with 8 pieces per wave they all go out in the first half of the k-tile, the second half is their time to land
#endif

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
    // LDS-DMA duty.  The two waves of a SIMD (w and w+4) together stage rows [64*(w&3), +64) of the A
    // tile and of the W tile = 16 pieces of 1 KiB per k-tile; the older wave (0-3) takes the first
    // V3_NPA of them, the younger the rest.  Measured splits (`mg_selftest gemmprof 3`, cycles per k-tile
    // of the older / younger wave incl. barrier wait): 8/8 2960, 10/6 3190, 12/4 3030, 14/2 3050, 16/0
    // 2890-3270 — the pair's MFMA (2048) + issue-stall (~1000) total is conserved, an uneven split only
    // moves the idle time from one wave to the other.  8/8 stays.
#ifndef V3_NPA
#define V3_NPA 8
#endif
    constexpr int NPMAX = V3_NPA > 16 - V3_NPA ? V3_NPA : 16 - V3_NPA;
    const bool older = wave < 4;
    const int p0 = older ? 0 : V3_NPA;                  // first combined piece (0-7: A rows, 8-15: W rows)
    const int np = older ? V3_NPA : 16 - V3_NPA;
    const int prow0 = (wave & 3) * 64;
    const uint16_t* gp[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
        const int piece = p0 + i;                       // may run past this wave's share: guarded at issue
        const int row = prow0 + (piece & 7) * 8 + srow;
        if (piece < 8) {
            int64_t am = m0 + row;
            if (am > M - 1) am = M - 1;
            gp[i] = A + am * lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        } else {
            int wr = n0 + row;
            if (wr > N - 1) wr = N - 1;
            gp[i] = Wt + (int64_t)wr * ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
        }
    }
    // LDS byte offset of combined piece q inside a stage
    auto piece_lds = [&](int q) __attribute__((always_inline)) {
        return (q < 8 ? 0 : V3_A_BYTES) + (prow0 + (q & 7) * 8) * 128;
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* base = smem + (kt % V3_NSTAGE) * V3_STAGE;
        const int koff = kt * V3_BK;
#pragma unroll
        for (int i = 0; i < NPMAX; ++i)
            if (i < np) v3_glds16(gp[i] + koff, base + piece_lds(p0 + i));
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
        char* lnext = smem + ((kt + 1) % V3_NSTAGE) * V3_STAGE;
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
                    const int slot = kk * 4 + i * 2 + h;    // 16 slots per k-tile, one behind every second MFMA
                    if (refill) {
#pragma unroll
                        for (int q = slot * V3_PPS; q < (slot + 1) * V3_PPS && q < NPMAX; ++q)
                            if (q < np) v3_glds16(gp[q] + koff2, lnext + piece_lds(p0 + q));
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
    mg_gemm_epilogue<EPI, 2, 4>(acc, m0 + wm * 128, n0 + wn * 64, l31, g, M, N, bias, gate, out, ldo);
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
