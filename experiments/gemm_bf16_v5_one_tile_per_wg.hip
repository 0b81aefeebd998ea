// ARCHIVED (round 3): GEMM variant 5 — the 256x256x64 tile with one wave per SIMD, ONE tile per workgroup, 32x32x16 MFMA.
// It was the default for the gated-residual epilogue in round 2; variant 7 (persistent loop on 16x16x32 MFMAs) replaced it
// for every epilogue and variant 6 (its persistent form) stays in the library as the A/B partner.  Not built.
// bf16 GEMM, variant 5: the 256 x 256 x 64 tile with FOUR waves (2x2, 128 tokens x 128 features each),
// ONE wave per SIMD, accumulators in AGPRs.  Same reasoning as attn_hd128_w64.hip: a wave alone on
// its SIMD hides its fragment reads (8 ds_read_b128 per 16 MFMAs) behind its own MFMAs, and an
// unguarded LDS-DMA piece costs it ~30 cycles instead of the 60-100 measured with two waves per SIMD.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V5_BM 256
#define V5_BN 256
#define V5_BK 64
#define V5_THREADS 256
#define V5_A_BYTES (V5_BM * V5_BK * 2)  // 32 KiB
#define V5_W_BYTES (V5_BN * V5_BK * 2)  // 32 KiB
#define V5_STAGE (V5_A_BYTES + V5_W_BYTES)
#define V5_NSTAGE 2
#ifndef V5_PPS
#define V5_PPS 1   // LDS-DMA pieces per slot: with 8 pieces per wave they all go out in the first half of the k-tile, the second half is their time to land
#endif

typedef const __attribute__((address_space(1))) void* v5_gptr_t;
typedef __attribute__((address_space(3))) void* v5_lptr_t;
MG_DEV void v5_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v5_gptr_t)g, (v5_lptr_t)l, 16, 0, 0); }

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile (shared by the 256x256 kernels)

// hand-issued fragment reads (base VGPR + immediate) with counted waits: hipcc guards a register ring of plain
// LDS loads with `s_waitcnt lgkmcnt(0)` at every k-step boundary, i.e. it waits for the read it issued last
template <int OFF>
MG_DEV void v5_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void v5_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// fragment `slot` of a k-step in the order the MFMAs need them: fw0 fa0 fa1 fa2 fa3 fw1 fw2 fw3
MG_DEV void v5_rd_slot(int slot, bf16x8_t (&fa)[4], bf16x8_t (&fw)[4], unsigned abase, unsigned wbase) {
    switch (slot) {   // compile-time after unrolling
        case 0: v5_rd<0>(fw[0], wbase); break;
        case 1: v5_rd<0>(fa[0], abase); break;
        case 2: v5_rd<4096>(fa[1], abase); break;
        case 3: v5_rd<8192>(fa[2], abase); break;
        case 4: v5_rd<12288>(fa[3], abase); break;
        case 5: v5_rd<4096>(fw[1], wbase); break;
        case 6: v5_rd<8192>(fw[2], wbase); break;
        default: v5_rd<12288>(fw[3], wbase); break;
    }
}

template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V5_THREADS, 1) void gemm_bf16_v5_kernel(
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
    __shared__ __attribute__((aligned(16))) char smem[V5_NSTAGE * V5_STAGE];

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
    const int64_t m0 = (int64_t)tm * V5_BM;
    const int n0 = tn * V5_BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;     // 2 (token) x 2 (feature) waves, 128 x 128 each

    // DMA sources: wave w stages A rows [32w, 32w+32) and W rows [32w, 32w+32): 4 + 4 pieces of 8 rows
    const int srow = lane >> 3;
    // LDS-DMA duty: wave w stages rows [64w, 64w+64) of the A tile (pieces 0-7) and of the W tile (8-15)
    constexpr int NPMAX = 16;
    const int prow0 = wave * 64;
    const uint16_t* gp[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
        const int row = prow0 + (i & 7) * 8 + srow;
        if (i < 8) {
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
        return (q < 8 ? 0 : V5_A_BYTES) + (prow0 + (q & 7) * 8) * 128;
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* base = smem + (kt % V5_NSTAGE) * V5_STAGE;
        const int koff = kt * V5_BK;
#pragma unroll
        for (int i = 0; i < NPMAX; ++i) v5_glds16(gp[i] + koff, base + piece_lds(i));
    };

    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;
    const unsigned lds0 = (unsigned)(uintptr_t)(v5_lptr_t)smem;
    const int a_row_off = (wm * 128 + l31) * 128;
    const int w_row_off = V5_A_BYTES + (wn * 128 + l31) * 128;

    f32x16_t acc[4][4];      // [feature block][token block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / V5_BK;
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        tick(-1);
        // tile kt landed (every piece of it: two stages), and everyone is past compute(kt-1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tick(0);
        // the refill of the last k-tile re-loads that tile into the free stage: no guard around the pieces
        char* lnext = smem + ((kt + 1) % V5_NSTAGE) * V5_STAGE;
        const int koff2 = (kt + 1 < nk ? kt + 1 : kt) * V5_BK;
        tick(1);
        // fragment reads run ONE k-step ahead of the MFMAs that consume them (register double buffer), one
        // read behind every second MFMA, in need-order, waits counted (LDS returns in order)
        const unsigned lsb = lds0 + (kt % V5_NSTAGE) * V5_STAGE;
        bf16x8_t fa[2][4], fw[2][4];
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) v5_rd_slot(slot, fa[0], fw[0], lsb + a_row_off + (t3 << 4), lsb + w_row_off + (t3 << 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned coff = (unsigned)((t3 ^ ((kk + 1) << 1)) << 4);
            const unsigned abase = lsb + a_row_off + coff, wbase = lsb + w_row_off + coff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int slot = i * 2 + h;                 // 8 slots per k-step, one behind every second MFMA
                    // reads still allowed in flight: the younger ones of the previous k-step + this k-step's so far
                    const int fresh = kk < 3 ? slot : 0;
                    if (slot == 0) v5_wait<5>();                // fw0 fa0 fa1 landed
                    else if (slot == 1) { if (kk < 3) v5_wait<3 + 1>(); else v5_wait<3>(); }      // fa2 fa3
                    else if (slot == 2) { if (kk < 3) v5_wait<2 + 2>(); else v5_wait<2>(); }      // fw1
                    else if (slot == 4) { if (kk < 3) v5_wait<1 + 4>(); else v5_wait<1>(); }      // fw2
                    else if (slot == 6) { if (kk < 3) v5_wait<0 + 6>(); else v5_wait<0>(); }      // fw3
                    (void)fresh;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 2 * h; j < 2 * h + 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk < 3) v5_rd_slot(slot, fa[(kk + 1) & 1], fw[(kk + 1) & 1], abase, wbase);
                    if (kk < 2) {                               // 16 LDS-DMA pieces in the first half of the k-tile
                        const int q = kk * 8 + slot;
                        v5_glds16(gp[q] + koff2, lnext + piece_lds(q));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
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
    mg_gemm_epilogue<EPI, 4, 4>(acc, m0 + wm * 128, n0 + wn * 128, l31, g, M, N, bias, gate, out, ldo);
}

int mg_gemm_v5_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    const int64_t tiles_m64 = (M + V5_BM - 1) / V5_BM;
    const int tiles_n = (N + V5_BN - 1) / V5_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(V5_THREADS);
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16_v5_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, nullptr)
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v5_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm5_prof);
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
