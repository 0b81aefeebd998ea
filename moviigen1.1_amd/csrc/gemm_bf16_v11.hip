// bf16 GEMM, variant 11: variant 7's structure (256 x 256 x 64 tile on v_mfma_f32_16x16x32_bf16, four waves = one per SIMD, a wave
// owns 128 tokens x 128 features, two 64 KiB LDS stages, persistent XCD-contiguous tile loop with the next tile's first k-tile
// fetched in the last refill slot) with the instruction discipline of the vendor library's assembly kernel of the same
// structure, which runs 8-19 % faster than variants 7 / 8 on this chip (profiles/r04g_lib_gemm.log):
//   * variant 7 issues its non-MFMA work in CLUMPS — behind every four MFMAs `s_add m0, v_lshl_add_u64, s_nop, ds_read,
//     global_load_lds, s_waitcnt` — and a clump outlasts the 16 cycles the fourth MFMA keeps the matrix pipe busy: ~23 idle
//     cycles x 32 groups per k-tile (2810 cycles for 2048 of MFMAs).  Here every ds_read / LDS-DMA load stands ALONE in the gap
//     behind an MFMA; the order of the whole k-tile is generated (tools/gen_gemm_v11_schedule.py -> gemm_bf16_v11_ktile_s*.inc),
//     including the minimal counted lgkmcnt waits; the last 32 MFMAs of a k-tile run BEHIND the next k-tile's barrier, in the
//     shadow of its first fragment reads (s_memtime, N = K = 5120: 2528 cycles per k-tile; variant 7: 2929);
//   * the LDS-DMA loads are BUFFER loads (`buffer_load_dwordx4 v_off, s[rsrc], s_koff offen lds`): one 32-bit VGPR offset per
//     piece, set once per output tile (rows clamped there), the k advance in a scalar offset, the tile's base row in the
//     resource — no 64-bit per-lane pointers, no VALU address arithmetic and no s_nop in the loop.
//   * epilogues shaped for the CU's memory pipe, which takes one request per LANE when adjacent lanes are different output rows
//     (experiments/store_probe.hip): the bf16 outputs store 16 bytes per lane (W rows staged in a permuted order, v11_epilogue_pair),
//     the fp32 outputs go through LDS and out as whole row segments, the residual read the same way (v11_epilogue_rows).
// LDS image, XOR swizzle (on the source offset), fragment addressing, MFMA roles and order and tile raster are variant 7's, the
// epilogue arithmetic is gemm_epilogue.h's element for element: same accumulation order, identical bits
// (test_gemm_tile_variants_agree, test_gemm_default_epilogues_match_direct_ones).
#include "gemm_v11_common.h"

extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile

// SCHED = gap stride of the LDS-DMA loads in the generated k-tile (tools/gen_gemm_v11_schedule.py; chosen in the launcher)
template <int EPI, int SCHED, bool PROF = false>
__global__ __launch_bounds__(V11_THREADS, 1) void gemm_bf16_v11_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, int raster, int flags, unsigned long long* __restrict__ prof) {
    __shared__ __attribute__((aligned(16))) char smem[2 * V11_STAGE];
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, tt[4] = {0, 0, 0, 0};      // PROF: s_memtime per k-tile {barrier, to first MFMA, k-step 0, k-step 1, k-tiles}
    unsigned long long pe[3] = {0, 0, 0};                                  // PROF: {whole kernel, tail + epilogue, tiles} of this wave
    const unsigned long long t_start = PROF ? __builtin_amdgcn_s_memtime() : 0;

    // bf16 outputs: W rows staged in the permuted order of v11_epilogue_pair (the launcher sends unaligned outputs to variant 8)
    constexpr bool PAIRED = EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    // XCD-contiguous raster (variant 7): workgroup b of XCD b & 7 takes, in iteration i, position i * (nwg / 8) + (b >> 3) of
    // its XCD's range [x * q + min(x, r), ...), q = total / 8, r = total % 8
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;        // host guarantees nwg % 8 == 0
    // raster 0: XCD x owns a contiguous range of tile positions (bands of 4 row tiles); rasters 1 / 2 / 3 (nwg == 256): the whole
    // chip works on one super-tile of 16 x 16 / 32 x 8 / 8 x 32 output tiles per iteration (variant 8's rasters)
    const int XR = raster == 2 ? 8 : raster == 3 ? 2 : 4;
    const int SR = 4 * XR;
    const int GM = raster ? SR : 4;
    const int per_group = GM * tiles_n;
    const int slot = bid >> 3;
    const int p256 = (8 * (xcd / XR) + (slot >> 2)) * SR + 4 * (xcd % XR) + (slot & 3);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;     // 2 (token) x 2 (feature) waves, 128 x 128 each
    const int srow = lane >> 3;
    constexpr int NP = 16;                       // LDS-DMA duty: wave w stages rows [64w, 64w+64) of A (pieces 0-7) and of W (8-15)
    const int prow0 = wave * 64;

    auto tile_of = [&](int pos, int64_t& m0, int& n0) __attribute__((always_inline)) {
        const int swz = raster ? pos * 256 + p256 : xcd_first + pos;
        const int group = swz / per_group;
        const int first_m = group * GM;
        const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
        const int in_g = swz - group * per_group;
        m0 = (int64_t)(first_m + in_g % gsz) * V11_BM;
        n0 = (in_g / gsz) * V11_BN;
    };
    // Per piece ONE 32-bit byte offset relative to the tile's first row (A + m0 * lda resp. Wt + n0 * ldw live in the buffer
    // resources): row * ld * 2 + (chunk ^ swizzle) * 16, rows past M / N clamped to the last one (never stored).
    int voff[NP];
    u32x4_t rs_a, rs_w;
    auto set_offsets = [&](int64_t m0, int n0) __attribute__((always_inline)) {
        const int64_t rows_a = M - m0;       // >= 1
        const int rows_w = N - n0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = prow0 + (i & 7) * 8 + srow;
            const int chunk = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
            if (i < 8) {
                const int r = row < rows_a ? row : (int)(rows_a - 1);
                voff[i] = r * (int)(lda * 2) + chunk;
            } else {
                const int f = PAIRED ? v11_feature_of_row(row) : row;      // which W row (feature) this LDS row holds
                const int r = f < rows_w ? f : rows_w - 1;
                voff[i] = r * (int)(ldw * 2) + chunk;
            }
        }
        rs_a = v11_rsrc(A + m0 * lda);
        rs_w = v11_rsrc(Wt + (int64_t)n0 * ldw);
    };
    auto piece_lds = [&](int p) __attribute__((always_inline)) {
        return (p < 8 ? 0 : V11_A_BYTES) + (prow0 + (p & 7) * 8) * 128;
    };

    const int sw = (r16 >> 1) & 7;            // (row >> 1) & 7 of the lane's row in every 16-row block
    const int t3 = G ^ sw;                    // chunk of k-step 0; k-step 1: t3 ^ 4
    const unsigned lds0 = (unsigned)(uintptr_t)(v11_lptr_t)smem;
    const int a_row_off = (wm * 128 + r16) * 128;
    const int w_row_off = V11_A_BYTES + (wn * 128 + r16) * 128;
    const int nk = K / V11_BK;

    int pos = raster ? 0 : bid >> 3;
    if (raster ? p256 >= total : pos >= xcd_count) return;             // whole workgroup: no barrier is left waiting
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_offsets(m0, n0);
    // LDS byte address of piece p in stage s: lds0 + s * STAGE + prow0 * 128 + (p < 8 ? 0 : A_BYTES) + (p & 7) * 1024
    const unsigned lds_pieces = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + prow0 * 128));
#define V11_PIECE_IMM(p) (((p) < 8 ? 0 : V11_A_BYTES) + ((p) & 7) * 1024)
    {   // cold start of the FIRST tile only.  The resources were just written by v_readfirstlane and the loads are inline assembly: the
        // compiler does not see a VMEM instruction reading those SGPRs (5 wait states) — keep the distance by hand.
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 1" ::"s"(lds_pieces), "s"(V11_PIECE_IMM(i)) : "scc", "memory");
            v11_dma(voff[i], i < 8 ? rs_a : rs_w, 0);
        }
    }
    if (flags & 8) {      // measurement: skew the workgroups of an XCD over one tile period (29 x 8128 cycles ~ 237 k), so that their epilogues do not coincide
        for (int i = 0; i < (bid >> 3); ++i) __builtin_amdgcn_s_sleep(127);
    }
    int gk = 0;                                   // k-tiles consumed so far by this workgroup: stage = gk & 1
    bf16x8_t f0a[8], f0w[8], f1a[8], f1w[8];      // fragments of k-step 0 / 1; f1* carry the TAIL of a k-tile's k-step 1 across the next barrier
#define V11_SB __builtin_amdgcn_sched_barrier(0)
#define V11_T(k) if (PROF) tt[k] = __builtin_amdgcn_s_memtime()
    for (;;) {
        f32x4_t acc[8][8];       // [feature block of 16][token block of 16]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // what the tail of "k-tile -1" multiplies: zeros (a few MFMAs per output tile that add nothing)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f1w[i] = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
            f1a[i] = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
        }
        const int next_pos = raster ? pos + 1 : pos + per_iter;
        const bool has_next = raster ? next_pos * 256 + p256 < total : next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        for (int kt = 0; kt < nk; ++kt, ++gk) {
            // k-tile kt landed (every piece of it), and everyone is past the compute of the previous one
            const unsigned long long tq = PROF ? __builtin_amdgcn_s_memtime() : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (PROF) tt[0] = __builtin_amdgcn_s_memtime();
            if (flags & 1) {      // measurement: de-phase the four waves by 16 cycles each, so that their loads / reads do not collide
                if (wave & 1) asm volatile("s_nop 15");
                if (wave & 2) asm volatile("s_nop 15\n\ts_nop 15");
            }
            int kbytes2 = (kt + 1) * (V11_BK * 2);
            if (kt == nk - 1) {   // refill slot of the last k-tile: the first k-tile of the NEXT tile (or a redundant re-load)
                kbytes2 = has_next ? 0 : kt * (V11_BK * 2);
                if (has_next) {
                    tile_of(next_pos, m0n, n0n);
                    set_offsets(m0n, n0n);
                }
            }
            const unsigned lnext = lds_pieces + ((gk + 1) & 1) * V11_STAGE;       // scalar: where this wave's pieces of the next k-tile go
            const unsigned lsb = lds0 + (gk & 1) * V11_STAGE;
            const unsigned ab0 = lsb + a_row_off + (t3 << 4), wb0 = lsb + w_row_off + (t3 << 4);                  // k-step 0
            const unsigned ab1 = lsb + a_row_off + ((t3 ^ 4) << 4), wb1 = lsb + w_row_off + ((t3 ^ 4) << 4);      // k-step 1
#define V11_M0(p) v11_set_m0<V11_PIECE_IMM(p)>(lnext)
#define V11_G(p) v11_dma(voff[p], (p) < 8 ? rs_a : rs_w, kbytes2)
            V11_SB;
            // the last MFMAs of k-tile kt-1 run first, in the shadow of this k-tile's first fragment reads (kt == 0: they multiply
            // the zeroed fragments below — one code path, no branch around 250 instructions)
            if constexpr (SCHED == 6) {
#include "gemm_bf16_v11_ktile_s6.inc"
            } else {
#include "gemm_bf16_v11_ktile_s4.inc"
            }
#undef V11_G
#undef V11_M0
            // The MFMAs above are inline asm: the compiler's hazard recognizer does not know that their results are still in the
            // matrix pipe.  Inside the loop nothing reads an accumulator; behind the LAST k-tile the register allocator may
            // shuffle accumulators on the loop's exit edge (v_accvgpr_mov right behind the last MFMA read stale values): pad there.
            if (kt == nk - 1) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            if (PROF) {
                tt[3] = __builtin_amdgcn_s_memtime();
                pt[0] += tt[0] - tq, pt[1] += tt[1] - tt[0], pt[2] += tt[2] - tt[1], pt[3] += tt[3] - tt[2], pt[4] += 1;
            }
        }
        // the tail of the LAST k-tile (builtin MFMAs: the compiler orders the epilogue's accumulator reads behind them)
        const unsigned long long te = PROF ? __builtin_amdgcn_s_memtime() : 0;
        V11_SB;
#include "gemm_bf16_v11_tail.inc"
        // ---- epilogue (gemm_epilogue.h) of THIS tile; the next tile's first k-tile is already on its way ----
        if constexpr (PAIRED)
            v11_epilogue_pair<EPI>(acc, m0 + wm * 128, n0 + wn * 128, r16, G, (flags & 4) ? 0 : M, N, bias, out, ldo);      // flags & 4: measurement without the stores
        else {
            const int64_t m_wave = m0 + wm * 128;
            const int n_wave = n0 + wn * 128;
            // everyone is done with the fragments of the last k-tile: its stage is the transposition buffer (the OTHER stage is
            // receiving the next tile's first k-tile).  With the memory clobber: the builtin barrier is no fence to the compiler, which
            // moved the first LDS writes of the transposition in front of it.
            asm volatile("s_barrier" ::: "memory");
            if (!(flags & 16) && m_wave + 128 <= M && n_wave + 128 <= N)      // the wave's whole 128 x 128 block exists (wave-uniform)
                v11_epilogue_rows<EPI>(acc, smem + ((gk + 1) & 1) * V11_STAGE + wave * 16384, lane, r16, G, m_wave, n_wave, bias, gate, out, ldo);
            else
                mg_gemm_epilogue16<EPI, 8, 8>(acc, m_wave, n_wave, r16, G, (flags & 4) ? 0 : M, N, bias, gate, out, ldo);
        }
        if (PROF) pe[1] += __builtin_amdgcn_s_memtime() - te, pe[2] += 1;
        if (!has_next) break;
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
    }
#undef V11_PIECE_IMM
#undef V11_T
#undef V11_SB
    if (PROF && lane == 0 && prof) {
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(prof + wave * 5 + i, pt[i]);
        pe[0] = __builtin_amdgcn_s_memtime() - t_start;
#pragma unroll
        for (int i = 0; i < 3; ++i) atomicAdd(prof + 32 + wave * 3 + i, pe[i]);
    }
}

int mg_gemm_v8_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M, int N, int K,
                      int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);      // gemm_bf16_v8.hip

static int g_v11_flags = 0;     // measurement bits (mg_gemm_set_variant(110 + flags)): 1 = de-phase the waves, 2 = raster 0 always, 4 = no stores (timing only), 8 = skewed start, 16 = fp32 outputs: direct epilogue, 32 = the every-4th-gap schedule whatever K
void mg_gemm_v11_set_flags(int f) { g_v11_flags = f; }

int mg_gemm_v11_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                       int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (128 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    // 32-bit byte offsets inside a tile (255 rows x ld x 2 bytes + 128; the fp32 epilogue's 128 rows x ldo x 4) and bf16 pitches that only
    // allow 8-byte stores: variant 8 takes those shapes, as it did when it was the default
    const bool bf16_out = epilogue == MG_EPI_BIAS_BF16 || epilogue == MG_EPI_BIAS_GELU_BF16;
    if (lda * 2 * 256 > 0x7fffffffLL || ldw * 2 * 256 > 0x7fffffffLL || ldo * (bf16_out ? 2 : 4) * 128 > 0x7fffffffLL || (bf16_out && (ldo & 7)))
        return mg_gemm_v8_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, st);
    const int64_t tiles_m64 = (M + V11_BM - 1) / V11_BM;
    const int tiles_n = (N + V11_BN - 1) / V11_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;          // few tiles: one iteration, still a multiple of 8 (idle ones return)
    // raster by shape: variant 8's rule (profiles/r03r_gemm_rasters.log)
    const int raster = (nwg == 256 && tiles_n < 32 && !(g_v11_flags & 2)) ? (K > 8192 ? 1 : 3) : 0;
    const dim3 grid((unsigned)nwg), block(V11_THREADS);
    // which generated k-tile: loads every 4th gap — K > 8192, and every shape on raster 0 (wide outputs: the XCDs stream different
    // bands, the loads take longer to land; q|k|v 1332 vs 1281, ffn.0 1234 vs 1187 TFLOP/s, profiles/r04u_gemm_wide_s4.log) — or
    // every 6th (the chip-wide rasters at K <= 8192: 2528 vs 2611 cycles per k-tile)
    const bool sched4 = K > 8192 || raster == 0 || (g_v11_flags & 32);
    if (g_gemm5_prof && epilogue == MG_EPI_BIAS_BF16) {
        if (sched4)
            hipLaunchKernelGGL((gemm_bf16_v11_kernel<MG_EPI_BIAS_BF16, 4, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, gate,
                               tiles_m, tiles_n, raster, g_v11_flags, g_gemm5_prof);
        else
            hipLaunchKernelGGL((gemm_bf16_v11_kernel<MG_EPI_BIAS_BF16, 6, true>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, gate,
                               tiles_m, tiles_n, raster, g_v11_flags, g_gemm5_prof);
        return mg_check_launch();
    }
#define LAUNCH(E)                                                                                                          \
    do {                                                                                                                   \
        if (sched4)                                                                                \
            hipLaunchKernelGGL((gemm_bf16_v11_kernel<E, 4, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                               gate, tiles_m, tiles_n, raster, g_v11_flags, nullptr);                                      \
        else                                                                                                               \
            hipLaunchKernelGGL((gemm_bf16_v11_kernel<E, 6, false>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                               gate, tiles_m, tiles_n, raster, g_v11_flags, nullptr);                                      \
    } while (0)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
