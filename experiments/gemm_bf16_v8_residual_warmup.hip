// ARCHIVED EXPERIMENT (round 4, not built into the library): GEMM variant 8 with a residual warm-up — `warm_kt` k-tiles before
// the end of a tile every wave touches the 128-byte lines of its 128 x 64 fp32 block of x with four LDS-DMA dword loads into a
// sink, so that the gated-residual epilogue's four read-modify-write round trips hit the L2.  Measured same-box, alternating
// (profiles/r04c_gemm_residual_warmup.log, M = 131 040): o-proj 1111-1114 TFLOP/s without vs 1070-1078 with (1 / 2 / 3 k-tiles
// ahead), ffn.2 1219-1226 vs 1199-1207, the store epilogue unchanged (1273-1279): the warm-up costs 1.5-4 % instead of
// gaining — the epilogue is bound by the CU's 64 B/clk vector-memory path (512 KiB of x in and out per tile), not by the
// latency of its first loads, and the extra requests compete with the operand stream.
// bf16 GEMM, variant 8: the 256 x 256 x 64 tile of variant 7 with EIGHT waves in two ping-pong groups — the structure that
// hides a wave's LDS-DMA issue and fragment reads behind its SIMD partner's MFMAs (cdna guide 5: the 8-phase idea, built
// here on this library's LDS image and refill protocol).
//
// Why: variant 7 (one wave per SIMD) is issue-limited, not power-limited: 58 % matrix-pipe occupancy at 2.08 GHz; per
// k-tile 1774 cycles for the 64 MFMAs of the k-step that carries the wave's 16 LDS-DMA pieces vs 1036 without
// — a wave that issues a piece or waits for fragments blocks its own MFMAs and nothing else runs on that SIMD.
//
// Structure: waves 0-3 (group X: tokens 0-127 of the tile) and 4-7 (group Y: tokens 128-255); wave w and w + 4 share a
// SIMD.  A wave owns 128 tokens x 64 features (4 x 8 accumulators of 4 registers = 128 AGPRs; <= 256 registers per wave).
// A k-tile is FOUR phases per wave; a phase = { load part: fragment reads (8 / 4 / 8 / 4 ds_read_b128) + 4 LDS-DMA pieces
// of the next k-tile in phases 0 and 1; lgkmcnt(0) ; s_barrier ; 16 MFMAs (4 feature x 4 token blocks, one 32-deep k-step)
// ; s_barrier }.  Group Y runs ONE barrier behind group X (an extra barrier at its start, one at X's end), so in every
// interval between two barriers one group computes while the other loads: X: L0 | M0 | L1 | M1 ...  Y: -- | L0 | M0 | L1 ...
// Phase order (token quarter, k-step): (0,0) (1,0) (1,1) (0,1): consecutive phases share the feature fragments or the
// k-step, so the reads are 8, 4, 8, 4.  LDS hazards: every read is retired (lgkmcnt(0)) BEFORE the barrier that ends its
// load part, a wave's own pieces are waited for (vmcnt(0)) in the load part of phase 3, i.e. before the barrier after
// which the first wave reads the next k-tile; the stage refilled in phases 0-1 of k-tile t+1 was last read in phase 3 of
// k-tile t, whose reads retired before the barrier in between — for either group.
// Same LDS image, swizzle, persistent XCD-contiguous tile loop, next-tile prefetch and epilogue as variant 7; same
// arithmetic and accumulation order: identical bits.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V8_BM 256
#define V8_BN 256
#define V8_BK 64
#define V8_THREADS 512
#define V8_A_BYTES (V8_BM * V8_BK * 2)  // 32 KiB
#define V8_W_BYTES (V8_BN * V8_BK * 2)  // 32 KiB
#define V8_STAGE (V8_A_BYTES + V8_W_BYTES)

typedef const __attribute__((address_space(1))) void* v8_gptr_t;
typedef __attribute__((address_space(3))) void* v8_lptr_t;
MG_DEV void v8_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v8_gptr_t)g, (v8_lptr_t)l, 16, 0, 0); }
MG_DEV void v8_glds4(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v8_gptr_t)g, (v8_lptr_t)l, 4, 0, 0); }

template <int OFF>
MG_DEV void v8_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// four fragment reads of consecutive 16-row blocks (2 KiB apart) starting at block b0
template <int B0>
MG_DEV void v8_rd4(bf16x8_t (&f)[4], unsigned base) {
    v8_rd<(B0 + 0) * 2048>(f[0], base);
    v8_rd<(B0 + 1) * 2048>(f[1], base);
    v8_rd<(B0 + 2) * 2048>(f[2], base);
    v8_rd<(B0 + 3) * 2048>(f[3], base);
}

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile

// Wave priorities (s_setprio around the MFMA part, around the load part, static for the second-dispatched group) were
// measured and make no difference here (profiles/r03i_gemm_v8_prio.log: all within 0.5 %): none is used.
template <int EPI, bool PROF = false>
__global__ __launch_bounds__(V8_THREADS, 2) void gemm_bf16_v8_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, int raster, int warm_kt, unsigned long long* __restrict__ prof) {
    unsigned long long pt[5] = {0, 0, 0, 0, 0};
    __shared__ __attribute__((aligned(16))) char smem[2 * V8_STAGE];
    __shared__ __attribute__((aligned(16))) char warm_sink[256];     // where the residual warm-up loads land (never read)

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;        // host guarantees nwg % 8 == 0
    // raster 0: XCD x owns a contiguous range of tile positions (bands of 4 row tiles, its 32 workgroups = 4 x 8 tiles).
    // rasters 1 / 2 / 3 (nwg == 256): the WHOLE chip works on one super-tile of 16 x 16 / 32 x 8 / 8 x 32 output tiles per
    // iteration — XCD x still owns a 4 x 8 block of it (an XR x 8/XR grid of XCDs), so what one L2 sees is unchanged, but
    // the A band of an XCD is also the band of 8/XR - 1 others and its 8 weight panels those of XR - 1 others: the re-reads
    // meet in the Infinity Cache instead of going to HBM, and consecutive iterations walk along N inside one band.
    const int XR = raster == 2 ? 8 : raster == 3 ? 2 : 4;      // XCD grid XR x (8 / XR) inside a super-tile of 4 XR rows x 64 / XR columns
    const int SR = 4 * XR;
    const int GM = raster ? SR : 4;
    const int per_group = GM * tiles_n;
    const int slot = bid >> 3;
    const int p256 = (8 * (xcd / XR) + (slot >> 2)) * SR + 4 * (xcd % XR) + (slot & 3);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;     // group X = wm 0 = waves 0-3, group Y = wm 1 = waves 4-7
    const int srow = lane >> 3;
    constexpr int NP = 8;                        // LDS-DMA duty: wave w stages rows [32w, 32w+32) of A (pieces 0-3) and of W (4-7)
    const int prow0 = wave * 32;

    auto tile_of = [&](int pos, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int swz = raster ? pos * 256 + p256 : xcd_first + pos;
        const int group = swz / per_group;
        const int first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = swz - group * per_group;
        m0 = (int64_t)(first_m + in_g % gsz) * V8_BM;
        n0 = (in_g / gsz) * V8_BN;
    };
    const uint16_t* gp[NP];
    auto set_pointers = [&](int64_t m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = prow0 + (i & 3) * 8 + srow;
            if (i < 4) {
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
        return (p < 4 ? 0 : V8_A_BYTES) + (prow0 + (p & 3) * 8) * 128;
    };

    const int sw = (r16 >> 1) & 7;            // (row >> 1) & 7 of the lane's row in every 16-row block
    const int t3 = G ^ sw;                    // chunk of k-step 0; k-step 1: t3 ^ 4
    const unsigned lds0 = (unsigned)(uintptr_t)(v8_lptr_t)smem;
    const int a_row_off = (wm * 128 + r16) * 128;
    const int w_row_off = V8_A_BYTES + (wn * 64 + r16) * 128;
    const int nk = K / V8_BK;

    int pos = raster ? 0 : bid >> 3;
    if (raster ? p256 >= total : pos >= xcd_count) return;             // whole workgroup: no barrier is left waiting
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_pointers(m0, n0);
#pragma unroll
    for (int i = 0; i < NP; ++i) v8_glds16(gp[i], smem + piece_lds(i));      // cold start of the FIRST tile only
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
        const int next_pos = raster ? pos + 1 : pos + per_iter;
        const bool has_next = raster ? next_pos * 256 + p256 < total : next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            int koff2 = (kt + 1) * V8_BK;
            if (kt == nk - 1) {   // refill of the last k-tile: the first k-tile of the NEXT tile (or a redundant re-load)
                koff2 = has_next ? 0 : kt * V8_BK;
                if (has_next) {
                    tile_of(next_pos, m0n, n0n);
                    set_pointers(m0n, n0n);
                }
            }
            char* lnext = smem + ((gk + 1) & 1) * V8_STAGE;
            const unsigned lsb = lds0 + (gk & 1) * V8_STAGE;
            const unsigned ab0 = lsb + a_row_off + (t3 << 4), wb0 = lsb + w_row_off + (t3 << 4);                 // k-step 0
            const unsigned ab1 = lsb + a_row_off + ((t3 ^ 4) << 4), wb1 = lsb + w_row_off + ((t3 ^ 4) << 4);     // k-step 1
            bf16x8_t fa[4], fw[4];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int tq = (ph == 0 || ph == 3) ? 0 : 1;      // token quarter of this phase; k-step = ph >> 1
                const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
                // ---- load part -------------------------------------------------------------------------------
                if (ph == 0) { v8_rd4<0>(fw, wb0); v8_rd4<0>(fa, ab0); }
                else if (ph == 1) v8_rd4<4>(fa, ab0);
                else if (ph == 2) { v8_rd4<0>(fw, wb1); v8_rd4<4>(fa, ab1); }
                else v8_rd4<0>(fa, ab1);
                if (ph < 2) {
#pragma unroll
                    for (int p = 4 * ph; p < 4 * ph + 4; ++p) v8_glds16(gp[p] + koff2, lnext + piece_lds(p));
                }
                // Gated-residual epilogue: its four read-modify-write round trips of x (one per column group; the 128 arch
                // VGPRs hold one group's eight quads at a time) run with the matrix pipe idle.  `warm_kt` k-tiles before the
                // end, right behind this k-tile's last piece, the wave touches every 128-byte line of its 128 x 64 fp32 block
                // of x — four LDS-DMA dword loads into a sink, no VGPR destination, nothing to keep alive — so the
                // epilogue's loads find their lines in the L2 instead of starting four HBM round trips.  vmcnt retires
                // loads in order: in this k-tile phase 3 waits for all but the four youngest (= the pieces).
                const bool warm_now = EPI == MG_EPI_GATE_RESID_F32 && warm_kt > 0 && kt == nk - warm_kt;
                if (EPI == MG_EPI_GATE_RESID_F32 && ph == 1 && warm_now) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int id = i * 64 + lane;                       // line (row id >> 1, half id & 1) of the wave's block
                        int64_t xm = m0 + wm * 128 + (id >> 1);
                        if (xm > M - 1) xm = M - 1;
                        int xn = n0 + wn * 64 + (id & 1) * 32;
                        if (xn > N - 4) xn = N - 4;
                        v8_glds4((const float*)out + xm * ldo + xn, warm_sink);
                    }
                }
                if (ph == 3) {      // my pieces of the next k-tile have landed
                    if (warm_now) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
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

// k-tiles before the end of a tile at which the gated-residual kernel warms the L2 with its block of x (0 = never);
// mg_gemm_set_variant(80 + n) selects n for A/B runs (gemm_bf16.hip), 8 = the default below
static int g_v8_warm_kt = 1;
void mg_gemm_v8_set_warm(int kt) { g_v8_warm_kt = kt < 0 ? 0 : kt; }

int mg_gemm_v8_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + V8_BM - 1) / V8_BM;
    const int tiles_n = (N + V8_BN - 1) / V8_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    // Raster by shape (profiles/r03o_gemm_raster.log, r03r_gemm_rasters.log; M = 131040, TFLOP/s for rasters 0 / 1 / 2 / 3):
    //   q|k|v 1295 / 1267 / 1299 / 1231, ffn.0 1214 / 1186 / 1219 / 1184            -> wide outputs stay on raster 0
    //   o (K 5120, gated residual) 1076 / 1129 / 1063 / 1153, cross-q 1236 / 1286 / 1261 / 1284   -> raster 3 (8 x 32)
    //   ffn.2 (K 13824) 1250 / 1250 / 1245 / 1238                                   -> raster 1 (16 x 16)
    const int raster = (nwg == 256 && tiles_n < 32) ? (K > 8192 ? 1 : 3) : 0;
    const dim3 grid((unsigned)nwg), block(V8_THREADS);
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        hipLaunchKernelGGL((gemm_bf16_v8_kernel<MG_EPI_BIAS_BF16, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K,
                           out, ldo, gate, tiles_m, tiles_n, raster, 0, g_gemm5_prof);
        return mg_check_launch();
    }
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL((gemm_bf16_v8_kernel<E, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n, raster, (K / V8_BK) > g_v8_warm_kt ? g_v8_warm_kt : 0, nullptr)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
