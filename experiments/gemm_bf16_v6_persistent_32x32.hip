// ARCHIVED (round 3): GEMM variant 6 — the persistent 256x256x64 loop with one wave per SIMD on 32x32x16 MFMAs (the
// round-2 default).  Variant 7 is the same loop on 16x16x32 MFMAs (+2-4 %), variant 8 its 8-wave ping-pong form.  Not built.
// bf16 GEMM, variant 6: variant 5's tile and k-loop (256 x 256 x 64, four waves = one per SIMD, AGPR accumulators,
// LDS-DMA pieces and hand-issued fragment reads between the MFMAs) inside a PERSISTENT tile loop: one workgroup per
// CU walks the tile list, and the refill slot of a tile's LAST k-tile — which variant 5 spends on a redundant re-load —
// fetches the FIRST k-tile of the workgroup's next tile.  The operands of tile T+1 are therefore in flight while the
// epilogue of tile T runs (variant 5 starts every tile cold: dispatch, 16 pointer set-ups, a full DMA round trip
// with the matrix pipe idle), and one launch of <= 256 workgroups replaces 10-30 thousand workgroup dispatches.
// Tile order = variant 5's XCD-contiguous raster evaluated on the virtual workgroup id (iteration * grid + block):
// in iteration i the 32 workgroups of an XCD work on 32 consecutive tiles of that XCD's range (4 token bands x 8
// feature panels), streaming through K together, so an A or W k-slice is fetched from the fabric once per XCD.
// Same arithmetic, same accumulation order, same epilogue as variants 1/2/5: identical bits.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V6_BM 256
#define V6_BN 256
#define V6_BK 64
#define V6_THREADS 256
#define V6_A_BYTES (V6_BM * V6_BK * 2)  // 32 KiB
#define V6_W_BYTES (V6_BN * V6_BK * 2)  // 32 KiB
#define V6_STAGE (V6_A_BYTES + V6_W_BYTES)

typedef const __attribute__((address_space(1))) void* v6_gptr_t;
typedef __attribute__((address_space(3))) void* v6_lptr_t;
MG_DEV void v6_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((v6_gptr_t)g, (v6_lptr_t)l, 16, 0, 0); }

template <int OFF>
MG_DEV void v6_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void v6_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// fragment `slot` of a k-step in the order the MFMAs need them: fw0 fa0 fa1 fa2 fa3 fw1 fw2 fw3
MG_DEV void v6_rd_slot(int slot, bf16x8_t (&fa)[4], bf16x8_t (&fw)[4], unsigned abase, unsigned wbase) {
    switch (slot) {   // compile-time after unrolling
        case 0: v6_rd<0>(fw[0], wbase); break;
        case 1: v6_rd<0>(fa[0], abase); break;
        case 2: v6_rd<4096>(fa[1], abase); break;
        case 3: v6_rd<8192>(fa[2], abase); break;
        case 4: v6_rd<12288>(fa[3], abase); break;
        case 5: v6_rd<4096>(fw[1], wbase); break;
        case 6: v6_rd<8192>(fw[2], wbase); break;
        default: v6_rd<12288>(fw[3], wbase); break;
    }
}

template <int EPI>
__global__ __launch_bounds__(V6_THREADS, 1) void gemm_bf16_v6_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[2 * V6_STAGE];

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
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
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
        m0 = (int64_t)(first_m + in_g % gsz) * V6_BM;
        n0 = (in_g / gsz) * V6_BN;
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
        return (p < 8 ? 0 : V6_A_BYTES) + (prow0 + (p & 7) * 8) * 128;
    };

    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;
    const unsigned lds0 = (unsigned)(uintptr_t)(v6_lptr_t)smem;
    const int a_row_off = (wm * 128 + l31) * 128;
    const int w_row_off = V6_A_BYTES + (wn * 128 + l31) * 128;
    const int nk = K / V6_BK;

    int pos = bid >> 3;
    if (pos >= xcd_count) return;
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_pointers(m0, n0);
    {   // cold start of the FIRST tile only
#pragma unroll
        for (int i = 0; i < NP; ++i) v6_glds16(gp[i], smem + piece_lds(i));
    }
    int gk = 0;                                   // k-tiles consumed so far by this workgroup: stage = gk & 1
    for (;;) {
        f32x16_t acc[4][4];      // [feature block][token block]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int next_pos = pos + per_iter;
        const bool has_next = next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            // k-tile kt landed (every piece of it), and everyone is past the compute of the previous one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int koff2 = (kt + 1) * V6_BK;
            if (kt == nk - 1) {   // refill slot of the last k-tile: the first k-tile of the NEXT tile (or a redundant re-load)
                koff2 = has_next ? 0 : kt * V6_BK;
                if (has_next) {
                    tile_of(next_pos, m0n, n0n);
                    set_pointers(m0n, n0n);
                }
            }
            char* lnext = smem + ((gk + 1) & 1) * V6_STAGE;
            const unsigned lsb = lds0 + (gk & 1) * V6_STAGE;
            bf16x8_t fa[2][4], fw[2][4];
#pragma unroll
            for (int slot = 0; slot < 8; ++slot) v6_rd_slot(slot, fa[0], fw[0], lsb + a_row_off + (t3 << 4), lsb + w_row_off + (t3 << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const unsigned coff = (unsigned)((t3 ^ ((kk + 1) << 1)) << 4);
                const unsigned abase = lsb + a_row_off + coff, wbase = lsb + w_row_off + coff;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int slot = i * 2 + h;                 // 8 slots per k-step, one behind every second MFMA
                        if (slot == 0) v6_wait<5>();                // fw0 fa0 fa1 landed
                        else if (slot == 1) { if (kk < 3) v6_wait<3 + 1>(); else v6_wait<3>(); }      // fa2 fa3
                        else if (slot == 2) { if (kk < 3) v6_wait<2 + 2>(); else v6_wait<2>(); }      // fw1
                        else if (slot == 4) { if (kk < 3) v6_wait<1 + 4>(); else v6_wait<1>(); }      // fw2
                        else if (slot == 6) { if (kk < 3) v6_wait<0 + 6>(); else v6_wait<0>(); }      // fw3
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 2 * h; j < 2 * h + 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kk & 1][i], fa[kk & 1][j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (kk < 3) v6_rd_slot(slot, fa[(kk + 1) & 1], fw[(kk + 1) & 1], abase, wbase);
                        if (kk < 2) {                               // 16 LDS-DMA pieces in the first half of the k-tile
                            const int p = kk * 8 + slot;
                            v6_glds16(gp[p] + koff2, lnext + piece_lds(p));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // ---- epilogue (gemm_epilogue.h) of THIS tile; the next tile's first k-tile is already on its way ----
        mg_gemm_epilogue<EPI, 4, 4>(acc, m0 + wm * 128, n0 + wn * 128, l31, g, M, N, bias, gate, out, ldo);
        if (!has_next) break;
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
    }
}

int mg_gemm_v6_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + V6_BM - 1) / V6_BM;
    const int tiles_n = (N + V6_BN - 1) / V6_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    const dim3 grid((unsigned)nwg), block(V6_THREADS);
#define LAUNCH(E)                                                                                          \
    hipLaunchKernelGGL(gemm_bf16_v6_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
