// bf16 GEMM, variant 10: the ping-pong structure of variant 8 (256 x 256 output tile, eight waves in two groups, one group
// computes while the other loads) on 32-DEEP k-tiles in FOUR 32 KiB LDS stages — a phase is a whole k-tile:
//     { 12 fragment reads + 4 LDS-DMA pieces of k-tile t+3 ; vmcnt(8) ; lgkmcnt(0) ; s_barrier ; 32 MFMAs ; s_barrier }
// i.e. two barriers per 32 MFMAs instead of variant 8's two per 16.
//
// Why a new LDS image instead of "variant 8 with two phases" (experiments/gemm_bf16_v8_two_phase.hip): with two 64 KiB
// stages a refill has four barrier intervals between "both groups have read the stage" and "the first group reads it
// again", LDS-DMA issue is only free in load parts, and no placement gives the pieces enough time to land (-3.5 % / -9 %).
// Four stages of a 32-deep k-tile make the window SIX intervals with the same 128 KiB of LDS: the pieces of k-tile t+3 go
// out in the load part of k-tile t (the stage of t-1: last read by the other group one interval earlier) and are awaited
// two k-tiles later — `vmcnt(8)` in the load part of k-tile t+2 retires everything but the 8 pieces issued since — for both
// groups alike, no skew, no issue between MFMAs.
//
// LDS image of a stage: A rows 0..255, then W rows 0..255, 64 bytes (32 k) per row = 4 chunks of 16 bytes; chunk c of row r
// sits in slot c ^ f(r), f(r) = (-(r >> 2)) & 3.  A fragment read (lane = (r16, G): row r16 of its 16-row block, k = 8G ..
// 8G+7) then touches, in each of the hardware's four ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...), 16 distinct
// 16-byte slots of the 256-byte bank window (brute-forced: f in {0,3,2,1} over (r >> 2) & 3 is one of eight such maps).
// An LDS-DMA piece = 64 lanes x 16 bytes = 16 rows: lane l -> row l >> 2, slot l & 3, source chunk (l & 3) ^ f(row): the
// swizzle is applied to the SOURCE address, the LDS side of global_load_lds is linear.
//
// MEASURED AND NOT ADOPTED (round 3, profiles/r03q_gemm_v10.log): correct on the whole selftest shape set at the first run
// (no spills in any instantiation), and in cycles it does what it was built for — load part 435-470, MFMA part 510 = gap-free,
// 1213 ticks per 32 MFMAs against variant 8's 657 per 16 — but it runs 2-4.5 % SLOWER than variant 8 on four of the five
// block shapes at M = 131040 (q|k|v 1246 vs 1286, cross-q 1220 vs 1278, ffn.0 1177 vs 1199, ffn.2 1236 vs 1246; gated
// residual o 1111 vs 1101): the GEMM sits at the chip's power cap (profiles/r03n_pmc_gemm_v8.txt) and the extra L2 requests of
// 64-byte rows cost more than the saved barriers return.  It was wired as mg_gemm_set_variant(10) / mg_gemm_v10_launch.
//
// Same persistent XCD-aware tile loop and rasters, next-tile prefetch (three k-tiles ahead, across output tiles),
// epilogue and accumulation order as variant 8: bit-identical results.  K >= 128 (four k-tiles), else variant 8 runs.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define VA_BM 256
#define VA_BN 256
#define VA_BK 32
#define VA_THREADS 512
#define VA_A_BYTES (VA_BM * VA_BK * 2)  // 16 KiB
#define VA_STAGE (2 * VA_A_BYTES)       // 32 KiB
#define VA_NSTAGE 4

typedef const __attribute__((address_space(1))) void* va_gptr_t;
typedef __attribute__((address_space(3))) void* va_lptr_t;
MG_DEV void va_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((va_gptr_t)g, (va_lptr_t)l, 16, 0, 0); }

template <int OFF>
MG_DEV void va_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// four fragment reads of consecutive 16-row blocks (1 KiB apart) starting at block B0
template <int B0>
MG_DEV void va_rd4(bf16x8_t (&f)[4], unsigned base) {
    va_rd<(B0 + 0) * 1024>(f[0], base);
    va_rd<(B0 + 1) * 1024>(f[1], base);
    va_rd<(B0 + 2) * 1024>(f[2], base);
    va_rd<(B0 + 3) * 1024>(f[3], base);
}

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile

template <int EPI, bool PROF = false>
__global__ __launch_bounds__(VA_THREADS, 2) void gemm_bf16_v10_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, int raster, unsigned long long* __restrict__ prof) {
    unsigned long long pt[5] = {0, 0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) char smem[VA_NSTAGE * VA_STAGE];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;        // host guarantees nwg % 8 == 0
    const int GM = raster ? 16 : 4;       // rasters of variant 8 (gemm_bf16_v8.hip)
    const int per_group = GM * tiles_n;
    const int slot = bid >> 3;
    const int p256 = (8 * (xcd >> 2) + (slot >> 2)) * 16 + 4 * (xcd & 3) + (slot & 3);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;     // group X = wm 0 = waves 0-3 (tokens 0-127), group Y = wm 1 = waves 4-7
    constexpr int NP = 4;                        // LDS-DMA duty: wave w stages rows [32w, 32w+32) of A (pieces 0-1) and of W (2-3)
    const int prow0 = wave * 32;
    const int prow = lane >> 2;                  // my row inside a 16-row piece
    const int psrc = ((lane & 3) ^ ((-(prow >> 2)) & 3)) << 3;     // source chunk (in elements) of my LDS slot

    auto tile_of = [&](int pos, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int swz = raster ? pos * 256 + p256 : xcd_first + pos;
        const int group = swz / per_group;
        const int first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = swz - group * per_group;
        m0 = (int64_t)(first_m + in_g % gsz) * VA_BM;
        n0 = (in_g / gsz) * VA_BN;
    };
    const uint16_t* gp[NP];
    auto set_pointers = [&](int64_t m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = prow0 + (i & 1) * 16 + prow;
            if (i < 2) {
                int64_t am = m0 + row;
                if (am > M - 1) am = M - 1;
                gp[i] = A + am * lda + psrc;
            } else {
                int wr = n0 + row;
                if (wr > N - 1) wr = N - 1;
                gp[i] = Wt + (int64_t)wr * ldw + psrc;
            }
        }
    };
    auto piece_lds = [&](int p) __attribute__((always_inline)) {
        return (p < 2 ? 0 : VA_A_BYTES) + (prow0 + (p & 1) * 16) * 64;
    };

    const unsigned lds0 = (unsigned)(uintptr_t)(va_lptr_t)smem;
    const int fsw = (G ^ ((-(r16 >> 2)) & 3)) << 4;          // byte offset of my fragment chunk inside its row
    const int a_off = (wm * 128 + r16) * 64 + fsw;
    const int w_off = VA_A_BYTES + (wn * 64 + r16) * 64 + fsw;
    const int nk = K / VA_BK;

    int pos = raster ? 0 : bid >> 3;
    if (raster ? p256 >= total : pos >= xcd_count) return;             // whole workgroup: no barrier is left waiting
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_pointers(m0, n0);
    // cold start of the FIRST tile: k-tiles 0, 1, 2 -> stages 0, 1, 2 (host guarantees nk >= 4)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < NP; ++i) va_glds16(gp[i] + t * VA_BK, smem + t * VA_STAGE + piece_lds(i));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // group Y runs one barrier behind group X from here on
    int gk = 0;                                   // k-tiles consumed so far by this workgroup: stage = gk & 3
    for (;;) {
        f32x4_t acc[4][8];       // [feature block of 16][token block of 16]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int next_pos = raster ? pos + 1 : pos + per_iter;
        const bool has_next = raster ? next_pos * 256 + p256 < total : next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            // refill: k-tile kt + 3 of this tile, or k-tile kt + 3 - nk of the NEXT tile (pointers switch at kt == nk - 3:
            // every piece of the current tile has been issued by then), or nothing when there is no next tile
            int kn = kt + 3;
            bool issue = true;
            if (kn >= nk) {
                kn -= nk;
                issue = has_next;
                if (has_next && kt == nk - 3) {
                    tile_of(next_pos, m0n, n0n);
                    set_pointers(m0n, n0n);
                }
            }
            const int koff = kn * VA_BK;
            char* lfill = smem + ((gk + 3) & 3) * VA_STAGE;
            const unsigned lsb = lds0 + (gk & 3) * VA_STAGE;
            bf16x8_t fw[4], fa[8];
            const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // ---- load part -----------------------------------------------------------------------------------
            {
                bf16x8_t (&falo)[4] = *(bf16x8_t (*)[4])&fa[0];
                bf16x8_t (&fahi)[4] = *(bf16x8_t (*)[4])&fa[4];
                va_rd4<0>(fw, lsb + w_off);
                va_rd4<0>(falo, lsb + a_off);
                va_rd4<4>(fahi, lsb + a_off);
            }
            if (issue) {
#pragma unroll
                for (int p = 0; p < NP; ++p) va_glds16(gp[p] + koff, lfill + piece_lds(p));
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // all but the 8 newest: my pieces of k-tile kt+1 have landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tail of the last tile: nothing new in flight
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my fragment reads are retired
            const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long c2 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // ---- MFMA part -----------------------------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long c3 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            __builtin_amdgcn_s_barrier();
            if (PROF) {
                const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                pt[0] += c1 - c0, pt[1] += c2 - c1, pt[2] += c3 - c2, pt[3] += c4 - c3, pt[4] += 1;
            }
        }
        // ---- epilogue (gemm_epilogue.h) of THIS tile; the next tile's first three k-tiles are staged / in flight ----
        mg_gemm_epilogue16<EPI, 4, 8>(acc, m0 + wm * 128, n0 + wn * 64, r16, G, M, N, bias, gate, out, ldo);
        if (!has_next) break;
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // group X's partner of group Y's last barrier
    if (PROF && lane == 0 && prof) {      // per wave: {load part, wait at barrier 1, MFMA part, wait at barrier 2, phases} -> prof[wave * 5 ..]
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(prof + wave * 5 + i, pt[i]);
    }
}

int mg_gemm_v10_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                       int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + VA_BM - 1) / VA_BM;
    const int tiles_n = (N + VA_BN - 1) / VA_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    const int raster = (nwg == 256 && tiles_n < 32) ? 1 : 0;      // as variant 8 (profiles/r03o_gemm_raster.log)
    const dim3 grid((unsigned)nwg), block(VA_THREADS);
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v10_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, raster, g_gemm5_prof);
        return mg_check_launch();
    }
#define LAUNCH(E)                                                                                                         \
    hipLaunchKernelGGL((gemm_bf16_v10_kernel<E, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, gate, \
                       tiles_m, tiles_n, raster, nullptr)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
