// bf16 GEMM, variant 7: variant 6 (256 x 256 x 64 tile, four waves = one per SIMD, persistent tile loop, LDS-DMA staging
// with the XOR swizzle on the source address) on v_mfma_f32_16x16x32_bf16.
//
// Why the other MFMA shape: on random data the 256x256 kernels run against the chip's POWER budget, not against issue
// cycles (experiments/mfma_shape_probe.hip): 16x16x32 does the same FLOPs with 8x less accumulator
// traffic per instruction and sustains ~10 % more under the cap.  A wave still owns 128 tokens x 128 features: 8 x 8
// accumulators of 4 registers (256 AGPRs, as before), a k-step is 32 deep = 8 feature + 8 token fragments of one
// ds_read_b128 each for 64 MFMAs (0.25 reads per MFMA = the 0.5 per 32x32x16 of variant 6), fragments double-buffered
// one k-step ahead.  The LDS image is unchanged: lane (r = lane & 15, G = lane >> 4) of an operand fragment reads row
// r of its 16-row block at 16-byte chunk (4*ks + G) ^ ((row >> 1) & 7) — the hardware's 16-lane read groups
// ({0-3, 12-15, 20-27}, ...) then cover 16 distinct slots of the 256-byte bank window (rows 2p / 2p+1 are the two
// halves of a window, the XOR spreads the row pairs over its 8 chunks): conflict-free.
// MFMA roles as in every variant: weights = A operand (feature rows), activations = B operand (token columns), so lane
// (tok, G) owns token tok of block j and features 16*i + 4G .. +3: row-local epilogue stores of 8 / 16 bytes.
// MFMA order inside a k-step: token half h (j = 4h .. 4h+3) outer, feature block i inner — the first group needs five
// fragments (fw0, fa0..3), every later group one more (fw1..fw7, then fa4..fa7 in one go): the 16 reads of k-step 0
// of a k-tile (which cannot be issued before the tile's barrier) are exposed for ~5 reads' latency only.
// Everything else is variant 6: one workgroup per
// CU walks the tile list, and the refill slot of a tile's LAST k-tile — which variant 5 spends on a redundant re-load —
// fetches the FIRST k-tile of the workgroup's next tile.  The operands of tile T+1 are therefore in flight while the
// epilogue of tile T runs (variant 5 starts every tile cold: dispatch, 16 pointer set-ups, a full DMA round trip
// with the matrix pipe idle), and one launch of <= 256 workgroups replaces 10-30 thousand workgroup dispatches.
// Tile order = variant 5's XCD-contiguous raster evaluated on the virtual workgroup id (iteration * grid + block):
// in iteration i the 32 workgroups of an XCD work on 32 consecutive tiles of that XCD's range (4 token bands x 8
// feature panels), streaming through K together, so an A or W k-slice is fetched from the fabric once per XCD.
// Same arithmetic and accumulation order (k ascending, fp32) as variants 1/2/5/6: identical bits.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V7_BM 256
#define V7_BN 256
#define V7_BK 64
#define V7_THREADS 256
#define V7_A_BYTES (V7_BM * V7_BK * 2)  // 32 KiB
#define V7_W_BYTES (V7_BN * V7_BK * 2)  // 32 KiB
#define V7_STAGE (V7_A_BYTES + V7_W_BYTES)

typedef const __attribute__((address_space(1))) void* v7_gptr_t;
typedef __attribute__((address_space(3))) void* v7_lptr_t;
MG_DEV void v7_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v7_gptr_t)g, (v7_lptr_t)l, 16, 0, 0); }

template <int OFF>
MG_DEV void v7_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void v7_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// fragment read `slot` (0..15) of a k-step, in the order the MFMAs need them: fw0 fa0 fa1 fa2 fa3 fw1 .. fw7 fa4 .. fa7
// (block b of an operand = rows 16b .. 16b+15 of the wave's 128: 2 KiB further in the LDS image)
MG_DEV void v7_rd_slot(int slot, bf16x8_t (&fa)[8], bf16x8_t (&fw)[8], unsigned abase, unsigned wbase) {
    switch (slot) {   // compile-time after unrolling
        case 0: v7_rd<0>(fw[0], wbase); break;
        case 1: v7_rd<0>(fa[0], abase); break;
        case 2: v7_rd<2048>(fa[1], abase); break;
        case 3: v7_rd<4096>(fa[2], abase); break;
        case 4: v7_rd<6144>(fa[3], abase); break;
        case 5: v7_rd<2048>(fw[1], wbase); break;
        case 6: v7_rd<4096>(fw[2], wbase); break;
        case 7: v7_rd<6144>(fw[3], wbase); break;
        case 8: v7_rd<8192>(fw[4], wbase); break;
        case 9: v7_rd<10240>(fw[5], wbase); break;
        case 10: v7_rd<12288>(fw[6], wbase); break;
        case 11: v7_rd<14336>(fw[7], wbase); break;
        case 12: v7_rd<8192>(fa[4], abase); break;
        case 13: v7_rd<10240>(fa[5], abase); break;
        case 14: v7_rd<12288>(fa[6], abase); break;
        default: v7_rd<14336>(fa[7], abase); break;
    }
}

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile

// The wave's 16 LDS-DMA pieces of the next k-tile ride behind the 16 MFMA groups of k-step 0.  s_memtime per k-tile
// (M = 131 040, N = K = 5120, profiles/r03e_gemm_dma_schedules.txt): barrier wait 119 + 1774 (k-step 0) + 1036 (k-step 1
// = the MFMA floor) cycles.  Moving pieces into k-step 1 shortens k-step 0 by ~35 cycles per piece but they then land
// after the next barrier wants them (8 + 8: wait 653 + 1491 + 1164; alternating groups: 899 + 1475 + 1136): every
// other schedule tried lost 2-11 % of wall.  (That experiment indexed gp[] with a computed constant — which hipcc folded
// in one epilogue instantiation and turned into 464 v_cndmask per k-tile in another: keep the index a plain loop variable.)
template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V7_THREADS, 1) void gemm_bf16_v7_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, unsigned long long* __restrict__ prof) {
    unsigned long long pt[4] = {0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) char smem[2 * V7_STAGE];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    // XCD-contiguous raster of variant 5 on the virtual id v: workgroup b of XCD b & 7 takes, in iteration i, position
    // i * (nwg / 8) + (b >> 3) of its XCD's range [x * q + min(x, r), ...), q = total / 8, r = total % 8
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;        // host guarantees nwg % 8 == 0
    const int GM = 4;                     // 4 x 256 = a 1024-token band
    const int per_group = GM * tiles_n;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;     // 2 (token) x 2 (feature) waves, 128 x 128 each
    const int srow = lane >> 3;
    constexpr int NP = 16;                       // LDS-DMA duty: wave w stages rows [64w, 64w+64) of A (pieces 0-7) and of W (8-15)
    const int prow0 = wave * 64;
    const int colsw[2] = {0, 0};
    (void)colsw;

    auto tile_of = [&](int pos, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int swz = xcd_first + pos;
        const int group = swz / per_group;
        const int first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = swz - group * per_group;
        m0 = (int64_t)(first_m + in_g % gsz) * V7_BM;
        n0 = (in_g / gsz) * V7_BN;
    };
    const uint16_t* gp[NP];
    auto set_pointers = [&](int64_t m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
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
    };
    auto piece_lds = [&](int p) __attribute__((always_inline)) {
        return (p < 8 ? 0 : V7_A_BYTES) + (prow0 + (p & 7) * 8) * 128;
    };

    const int sw = (r16 >> 1) & 7;            // (row >> 1) & 7 of the lane's row in every 16-row block
    const int t3 = G ^ sw;                    // chunk of k-step 0; k-step 1: t3 ^ 4
    const unsigned lds0 = (unsigned)(uintptr_t)(v7_lptr_t)smem;
    const int a_row_off = (wm * 128 + r16) * 128;
    const int w_row_off = V7_A_BYTES + (wn * 128 + r16) * 128;
    const int nk = K / V7_BK;

    int pos = bid >> 3;
    if (pos >= xcd_count) return;
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_pointers(m0, n0);
    {   // cold start of the FIRST tile only
#pragma unroll
        for (int i = 0; i < NP; ++i) v7_glds16(gp[i], smem + piece_lds(i));
    }
    int gk = 0;                                   // k-tiles consumed so far by this workgroup: stage = gk & 1
    for (;;) {
        f32x4_t acc[8][8];       // [feature block of 16][token block of 16]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int next_pos = pos + per_iter;
        const bool has_next = next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            // k-tile kt landed (every piece of it), and everyone is past the compute of the previous one
            const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            unsigned long long c2 = 0;
            int koff2 = (kt + 1) * V7_BK;
            if (kt == nk - 1) {   // refill slot of the last k-tile: the first k-tile of the NEXT tile (or a redundant re-load)
                koff2 = has_next ? 0 : kt * V7_BK;
                if (has_next) {
                    tile_of(next_pos, m0n, n0n);
                    set_pointers(m0n, n0n);
                }
            }
            char* lnext = smem + ((gk + 1) & 1) * V7_STAGE;
            const unsigned lsb = lds0 + (gk & 1) * V7_STAGE;
            bf16x8_t fa[2][8], fw[2][8];
            {   // k-step 0 of this k-tile: its fragments can only be read now (the barrier above published the tile)
                const unsigned ab0 = lsb + a_row_off + (t3 << 4), wb0 = lsb + w_row_off + (t3 << 4);
#pragma unroll
                for (int slot = 0; slot < 16; ++slot) v7_rd_slot(slot, fa[0], fw[0], ab0, wb0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned coff = (unsigned)((t3 ^ 4) << 4);
                const unsigned abase = lsb + a_row_off + coff, wbase = lsb + w_row_off + coff;     // k-step 1 (read during k-step 0)
#pragma unroll
                for (int grp = 0; grp < 16; ++grp) {          // group = 4 MFMAs: token half h = grp >> 3, feature block i = grp & 7
                    const int h = grp >> 3, i = grp & 7;
                    // fragments this group needs are the oldest outstanding reads; younger ones: the rest of this k-step's
                    // set (ks == 0: all 16 were issued up front) and, in k-step 0, the grp reads already issued for k-step 1
                    // ks 0, group g: needs slots <= need(g); outstanding allowed = (15 - need) + g
                    // ks 1, group g: its set was issued during ks 0 and nothing is issued behind it: allowed = 15 - need
                    const int need = grp == 0 ? 4 : (grp < 8 ? 4 + grp : 15);
                    const int allowed = (15 - need) + (ks == 0 ? grp : 0);
                    if (grp <= 8) {
                        switch (allowed) {      // compile-time; lgkmcnt is a 4-bit field
                            case 0: v7_wait<0>(); break;
                            case 1: v7_wait<1>(); break;
                            case 2: v7_wait<2>(); break;
                            case 3: v7_wait<3>(); break;
                            case 4: v7_wait<4>(); break;
                            case 5: v7_wait<5>(); break;
                            case 6: v7_wait<6>(); break;
                            case 7: v7_wait<7>(); break;
                            case 8: v7_wait<8>(); break;
                            case 9: v7_wait<9>(); break;
                            case 10: v7_wait<10>(); break;
                            case 11: v7_wait<11>(); break;
                            case 12: v7_wait<12>(); break;
                            case 13: v7_wait<13>(); break;
                            default: v7_wait<14>(); break;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 4 * h; j < 4 * h + 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) {
                        v7_rd_slot(grp, fa[1], fw[1], abase, wbase);                       // one read of k-step 1 ...
                        v7_glds16(gp[grp] + koff2, lnext + piece_lds(grp));                // ... and one LDS-DMA piece of the next k-tile
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (PROF && ks == 0) c2 = __builtin_amdgcn_s_memtime();
            }
            if (PROF) {
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                pt[0] += c1 - c0, pt[1] += c2 - c1, pt[2] += c3 - c2, pt[3] += 1;
            }
        }
        // ---- epilogue (gemm_epilogue.h) of THIS tile; the next tile's first k-tile is already on its way ----
        mg_gemm_epilogue16<EPI, 8, 8>(acc, m0 + wm * 128, n0 + wn * 128, r16, G, M, N, bias, gate, out, ldo);
        if (!has_next) break;
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
    }
    if (PROF && lane == 0 && prof) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(prof + wave * 4 + i, pt[i]);
    }
}

int mg_gemm_v7_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + V7_BM - 1) / V7_BM;
    const int tiles_n = (N + V7_BN - 1) / V7_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    const dim3 grid((unsigned)nwg), block(V7_THREADS);
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v7_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, g_gemm5_prof);
        return mg_check_launch();
    }
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL((gemm_bf16_v7_kernel<E, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
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
