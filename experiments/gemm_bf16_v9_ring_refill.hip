// ARCHIVED EXPERIMENT (round 3), built and measured, not shipped: variant 8 with the refill spread evenly over the four
// phases (ring of 3 activation + 2 weight panels = all 160 KiB of LDS).  Bit-identical to the other variants.  M = 131 040,
// same box, TFLOP/s (profiles/r03j_gemm_v7_v8_v9.log):      q|k|v   self-o(resid)  cross-q   ffn.0   ffn.2(resid)
//     variant 7 (one wave per SIMD)                           1209      1042         1175     1145      1200
//     variant 8 (ping-pong, refill in phases 0-1)             1256      1057         1215     1185      1214
//     variant 9 (this file)                                   1281      1019         1231     1214      1178
// +2 % where the epilogue only stores, -3 % on the residual epilogue (16 refill pointers: 6-16 spilled VGPRs at the
// 256-register limit of a 2-waves-per-SIMD kernel).  Net +0.3 % of a denoising step for a third 256x256 kernel: left out.
// bf16 GEMM, variant 9: variant 8 (eight waves in two ping-pong groups, 256 x 256 x 64 tile, 16x16x32 MFMA) with the
// LDS-DMA refill spread EVENLY over the four phases of a k-tile: 2 pieces per wave and phase instead of 4 + 4 + 0 + 0.
//
// Why: in variant 8 the load parts of phases 0 and 1 carry all 64 pieces of the next k-tile — 16 KiB per load part at the
// CU's ~64 B/clk vector-memory path = 256 cycles, as long as the partner group's 16 MFMAs — while phases 2 and 3 load
// nothing: the loading group is late at the barrier in half of the intervals (s_memtime, group X: load 185, wait 72, MFMA
// 256, wait 125 cycles per phase: the second wait is the partner's DMA phase).  With 8 KiB per load part every load part
// fits under the partner's MFMAs.  The price is LDS: a piece issued in phase 2 or 3 cannot be for the NEXT k-tile (its
// first fragment reads are one barrier away), so the activation panel runs TWO k-tiles ahead in a ring of three 32 KiB
// slots, the weight panel one k-tile ahead in a ring of two: 5 x 32 KiB = all 160 KiB of the CU's LDS.
//     phases 0, 1 of k-tile g : W(g+1) -> slot (g+1) % 2        phases 2, 3 : A(g+2) -> slot (g+2) % 3
//     phase 3, load part      : s_waitcnt vmcnt(4) — everything but the four youngest pieces (= A(g+2)) has landed,
//                               i.e. W(g+1) and A(g+1); then the barrier after which the first wave reads k-tile g+1.
// The ring runs across output tiles: near the end of a tile the refill addresses switch to the next tile (A two k-tiles
// early, W one), so a new tile starts with its first k-tile (and A of its second) resident.  Needs K >= 128.
// Everything else — LDS image of a panel, swizzle, phase order, fragment reads, barriers, epilogue — is variant 8.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V9_BM 256
#define V9_BN 256
#define V9_BK 64
#define V9_THREADS 512
#define V9_A_BYTES (V9_BM * V9_BK * 2)  // 32 KiB
#define V9_W_BYTES (V9_BN * V9_BK * 2)  // 32 KiB
#define V9_STAGE (V9_A_BYTES + V9_W_BYTES)

typedef const __attribute__((address_space(1))) void* v9_gptr_t;
typedef __attribute__((address_space(3))) void* v9_lptr_t;
MG_DEV void v9_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v9_gptr_t)g, (v9_lptr_t)l, 16, 0, 0); }

template <int OFF>
MG_DEV void v9_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// four fragment reads of consecutive 16-row blocks (2 KiB apart) starting at block b0
template <int B0>
MG_DEV void v9_rd4(bf16x8_t (&f)[4], unsigned base) {
    v9_rd<(B0 + 0) * 2048>(f[0], base);
    v9_rd<(B0 + 1) * 2048>(f[1], base);
    v9_rd<(B0 + 2) * 2048>(f[2], base);
    v9_rd<(B0 + 3) * 2048>(f[3], base);
}

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile

// Wave priorities (s_setprio around the MFMA part, around the load part, static for the second-dispatched group) were
// measured and make no difference here (profiles/r03i_gemm_v9_prio.log: all within 0.5 %): none is used.
template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V9_THREADS, 2) void gemm_bf16_v9_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, unsigned long long* __restrict__ prof) {
    unsigned long long pt[5] = {0, 0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) char smem[5 * V9_A_BYTES];      // A ring: slots 0-2, W ring: slots 3-4

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;        // host guarantees nwg % 8 == 0
    const int GM = 4;                     // 4 x 256 = a 1024-token band
    const int per_group = GM * tiles_n;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;     // group X = wm 0 = waves 0-3, group Y = wm 1 = waves 4-7
    const int srow = lane >> 3;
    constexpr int NP = 8;                        // LDS-DMA duty: wave w stages rows [32w, 32w+32) of A (pieces 0-3) and of W (4-7)
    const int prow0 = wave * 32;

    auto tile_of = [&](int pos, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int swz = xcd_first + pos;
        const int group = swz / per_group;
        const int first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = swz - group * per_group;
        m0 = (int64_t)(first_m + in_g % gsz) * V9_BM;
        n0 = (in_g / gsz) * V9_BN;
    };
    const uint16_t *gp[NP], *gn[NP];             // refill pointers of this output tile and of the workgroup's next one
    auto set_pointers = [&](const uint16_t* (&g)[NP], int64_t m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = prow0 + (i & 3) * 8 + srow;
            if (i < 4) {
                int64_t am = m0 + row;
                if (am > M - 1) am = M - 1;
                g[i] = A + am * lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
            } else {
                int wr = n0 + row;
                if (wr > N - 1) wr = N - 1;
                g[i] = Wt + (int64_t)wr * ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
            }
        }
    };
    auto piece_lds = [&](int p) __attribute__((always_inline)) {      // byte offset inside a 32 KiB panel slot
        return (prow0 + (p & 3) * 8) * 128;
    };

    const int sw = (r16 >> 1) & 7;            // (row >> 1) & 7 of the lane's row in every 16-row block
    const int t3 = G ^ sw;                    // chunk of k-step 0; k-step 1: t3 ^ 4
    const unsigned lds0 = (unsigned)(uintptr_t)(v9_lptr_t)smem;
    const int a_row_off = (wm * 128 + r16) * 128;
    const int w_row_off = 3 * V9_A_BYTES + (wn * 64 + r16) * 128;
    const int nk = K / V9_BK;

    int pos = bid >> 3;
    if (pos >= xcd_count) return;             // whole workgroup: no barrier is left waiting
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_pointers(gp, m0, n0);
#pragma unroll
    for (int i = 0; i < NP; ++i) gn[i] = gp[i];
    // cold start of the FIRST tile only: A(0) -> A slot 0, W(0) -> W slot 0, A(1) -> A slot 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v9_glds16(gp[i], smem + piece_lds(i));
        v9_glds16(gp[4 + i], smem + 3 * V9_A_BYTES + piece_lds(i));
        v9_glds16(gp[i] + V9_BK, smem + V9_A_BYTES + piece_lds(i));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // group Y runs one barrier behind group X from here on
    int gk = 0;                                   // k-tiles consumed so far by this workgroup: stage = gk & 1
    for (;;) {
        f32x4_t acc[4][8];       // [feature block of 16][token block of 16]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int next_pos = pos + per_iter;
        const bool has_next = next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        if (has_next) {
            tile_of(next_pos, m0n, n0n);
            set_pointers(gn, m0n, n0n);
        }
        int sa = gk % 3;                          // A ring slot of this k-tile (kept incrementally: no division in the loop)
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            // refill targets: W of k-tile kt+1 and A of k-tile kt+2 of this tile — or, past its end, of the NEXT tile
            // (without one: a redundant re-load of valid addresses into the free slots)
            const bool wn_ = kt + 1 >= nk, an_ = kt + 2 >= nk;
            const int kw = wn_ ? (has_next ? 0 : kt) : kt + 1;
            const int ka = an_ ? (has_next ? kt + 2 - nk : nk - 1) : kt + 2;
            // the refill pointers move on to the next tile exactly once each (they ARE the next tile's pointers from then
            // on): a wave-uniform branch with 8 / 8 moves, not a select per piece
            if (kt + 2 == nk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gp[i] = gn[i];
            }
            if (kt + 1 == nk) {
#pragma unroll
                for (int i = 4; i < 8; ++i) gp[i] = gn[i];
            }
            const int sa2 = sa == 0 ? 2 : sa - 1;                     // (gk + 2) % 3
            char* lw = smem + (3 + ((gk + 1) & 1)) * V9_A_BYTES;
            char* la = smem + sa2 * V9_A_BYTES;
            const unsigned lsa = lds0 + sa * V9_A_BYTES, lsw = lds0 + (gk & 1) * V9_A_BYTES;
            const unsigned ab0 = lsa + a_row_off + (t3 << 4), wb0 = lsw + w_row_off + (t3 << 4);                 // k-step 0
            const unsigned ab1 = lsa + a_row_off + ((t3 ^ 4) << 4), wb1 = lsw + w_row_off + ((t3 ^ 4) << 4);     // k-step 1
            bf16x8_t fa[4], fw[4];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int tq = (ph == 0 || ph == 3) ? 0 : 1;      // token quarter of this phase; k-step = ph >> 1
                const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                // ---- load part -------------------------------------------------------------------------------
                if (ph == 0) { v9_rd4<0>(fw, wb0); v9_rd4<0>(fa, ab0); }
                else if (ph == 1) v9_rd4<4>(fa, ab0);
                else if (ph == 2) { v9_rd4<0>(fw, wb1); v9_rd4<4>(fa, ab1); }
                else v9_rd4<0>(fa, ab1);
                if (ph < 2) {       // two pieces of W(kt+1)
#pragma unroll
                    for (int p = 2 * ph; p < 2 * ph + 2; ++p)
                        v9_glds16(gp[4 + p] + kw * V9_BK, lw + piece_lds(p));
                } else {            // two pieces of A(kt+2)
#pragma unroll
                    for (int p = 2 * (ph - 2); p < 2 * (ph - 2) + 2; ++p)
                        v9_glds16(gp[p] + ka * V9_BK, la + piece_lds(p));
                }
                // everything but my four youngest pieces (= A(kt+2)) has landed: W(kt+1), and A(kt+1) from the k-tile before
                if (ph == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // my fragment reads are retired
                const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long c2 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                // ---- MFMA part -------------------------------------------------------------------------------
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        acc[i][4 * tq + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[jj], acc[i][4 * tq + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long c3 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                __builtin_amdgcn_s_barrier();
                if (PROF) {
                    const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                    pt[0] += c1 - c0, pt[1] += c2 - c1, pt[2] += c3 - c2, pt[3] += c4 - c3, pt[4] += 1;
                }
            }
            sa = sa == 2 ? 0 : sa + 1;
        }
        // ---- epilogue (gemm_epilogue.h) of THIS tile; the next tile's first k-tile is already staged ----
        mg_gemm_epilogue16<EPI, 4, 8>(acc, m0 + wm * 128, n0 + wn * 64, r16, G, M, N, bias, gate, out, ldo);
        if (!has_next) break;
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // group X's partner of group Y's last barrier
    if (PROF && lane == 0 && prof) {      // per wave: {load part, wait at barrier 1, MFMA part, wait at barrier 2}, phases -> prof[wave * 5 ..]
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(prof + wave * 5 + i, pt[i]);
    }
}

int mg_gemm_v9_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + V9_BM - 1) / V9_BM;
    const int tiles_n = (N + V9_BN - 1) / V9_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    const dim3 grid((unsigned)nwg), block(V9_THREADS);
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v9_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm5_prof);
        return mg_check_launch();
    }
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL((gemm_bf16_v9_kernel<E, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, nullptr)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
