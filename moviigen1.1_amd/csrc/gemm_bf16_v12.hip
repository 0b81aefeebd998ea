// bf16 GEMM, variant 12: variant 11's tile, LDS image, fragment addressing, MFMA order, rasters and epilogues (gemm_v11_common.h) — identical
// bits — on a k-loop in which a stage is REFILLED WHILE IT IS CONSUMED, for the k-tile two ahead.
//
// Why (profiles/r05a_pmc_lib_gemm.txt, one box, same operands): variant 11 needs 23-26 shader cycles per MFMA and SIMD, the vendor
// library's assembly kernel of the same 256 x 256 x 64 tile 19.2; both run against the package-power limit (PPT residency 0.7,
// profiles/r05a_telemetry.log), so the vendor kernel does the same work at 1.72 instead of 2.05 GHz — at a lower voltage — and finishes
// 13 % earlier.  Variant 11's k-tile has one barrier behind `vmcnt(0)`: what a wave loads during k-tile t must have LANDED at the top of
// k-tile t+1, so its 16 LDS-DMA loads are packed into the first half of the k-tile, among the fragment reads, and the barrier waits for the
// stragglers.  With the same two 64 KiB stages a load can be given ~1.4 k-tiles instead:
//   body of stream position g (stage s = g & 1; k-step-0 fragments already in registers)             tools/gen_gemm_v12_schedule.py
//     64 MFMAs of k-step 0; in their gaps the 8 A-fragment reads of k-step 1, then
//        BAR1 (lgkmcnt(0) + s_barrier): nobody reads stage s's A rows any more -> the wave's 8 A loads of k-tile g+2 go INTO stage s,
//        between them the 8 W-fragment reads of k-step 1;  BAR2 (same): the W rows are free -> W loads of k-tile g+2
//     64 MFMAs of k-step 1; more loads;  BAR3 (vmcnt(13) + s_barrier): k-tile g+1 — issued one body ago — has landed in stage s^1 -> its 16
//        k-step-0 fragment reads, the last 3 loads.  No wait at the end: the next body's counted lgkmcnt waits follow the issue order.
// Three barriers per k-tile, none behind a drained memory pipe.  The k-tile STREAM runs across output tiles: the body of a tile's last-but-one
// k-tile loads the next tile's k-tile 0; the LAST k-tile of a tile is a body without loads, reads-ahead and barriers (gemm_bf16_v12_last.inc):
// its stage is the fp32 epilogues' transposition buffer, and the registers of the fragments read ahead are the epilogue's.  Behind the
// epilogue a short tile prologue issues the next tile's k-tile 1 and reads its k-step-0 fragments of k-tile 0 (landed: vmcnt(0) + barrier
// closed the last body).  Needs K >= 128 (two k-tiles); the launcher sends anything else to the general 256 x 128 kernel (gemm_bf16_v2.hip).
#include "gemm_v11_common.h"

#ifdef MG_AB_BUILD
extern unsigned long long* g_gemm5_prof;    // gemm_bf16.hip: mg_gemm5_debug_profile (A/B library only)
#endif

// SCHED = which generated body (tools/gen_gemm_v12_schedule.py: SCHEDULES); chosen in the launcher
// the lane index, re-derived where it is called: two VALU instructions the compiler can neither hoist out of the tile loop nor merge (see set_offsets)
MG_DEV int v12_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int EPI, int SCHED, bool PROF = false>
__global__ __launch_bounds__(V11_THREADS, 1) void gemm_bf16_v12_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n, int raster, int flags, unsigned long long* __restrict__ prof) {
    __shared__ __attribute__((aligned(16))) char smem[2 * V11_STAGE];
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, ta = 0;                    // PROF: s_memtime {wait at the lgkmcnt barriers, -, wait at the vmcnt barrier, whole body, bodies}
    unsigned long long ps[4] = {0, 0, 0, 0}, tsg = 0;                      // PROF: {last k-tile, vmcnt(0) + barrier, epilogue, tile prologue} at [44 + 4 wave]
    unsigned long long pe[3] = {0, 0, 0};                                  // PROF: {whole kernel, last body's close + epilogue + tile prologue, tiles}
    const unsigned long long t_start = PROF ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long r_start = PROF ? __builtin_amdgcn_s_memrealtime() : 0;      // the 100 MHz counter all workgroups share: [64 + 2 bid] = {start, end} of the LAST launch

    constexpr bool PAIRED = EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int total = tiles_m * tiles_n;
    // rasters: variant 11's (gemm_bf16_v11.hip)
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xcd_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcd_count = q8 + (xcd < r8 ? 1 : 0);
    const int per_iter = nwg >> 3;
    const int XR = raster == 2 ? 8 : raster == 3 ? 2 : 4;
    const int SR = 4 * XR;
    const int GM = raster ? SR : 4;
    const int per_group = GM * tiles_n;
    const int slot = bid >> 3;
    const int p256 = (8 * (xcd / XR) + (slot >> 2)) * SR + 4 * (xcd % XR) + (slot & 3);

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, r16 = lane & 15, G = lane >> 4;
    // FRESH: the lane index re-derived in front of each tile's offsets and epilogue (v12_lane).  Not for the fp32-store instantiation: with the registers that frees the
    // allocator parks accumulators in scratch INSIDE the last k-tile (scratch_store of a[128:131] between MFMAs that reuse them) and the tile comes out wrong
    // (test_gemm_epilogues[3-*], round 6) — that instantiation keeps the code it had (the DiT's fp32-store GEMM, the patch embedding, has K = 64: another kernel).
    constexpr bool FRESH = EPI != MG_EPI_BIAS_F32;
    const int lane0 = lane;
    const int wm = wave >> 1, wn = wave & 1;
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
    int voff[NP];
    u32x4_t rs_a, rs_w;
    auto set_offsets = [&](int64_t m0, int n0) __attribute__((always_inline)) {
        // (round 6) from a FRESH lane index, and the row clamp in 32 bits: the rows and chunks of the 16 pieces depend on the lane only, so derived from
        // threadIdx the compiler computed them once at kernel entry (48 registers), kept them in scratch and reloaded them value by value — 26 to 30
        // scratch_load / s_waitcnt vmcnt(0) pairs in a row at the top of EVERY tile, the matrix pipe empty
        const int lane = FRESH ? v12_lane() : lane0;
        const int srow = lane >> 3;
        const int64_t rows_a64 = M - m0;
        const int rows_a = rows_a64 < 0x40000000 ? (int)rows_a64 : 0x40000000;
        const int rows_w = N - n0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = prow0 + (i & 7) * 8 + srow;
            const int chunk = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
            if (i < 8) {
                const int r = FRESH ? (row < rows_a ? row : rows_a - 1) : (row < rows_a64 ? row : (int)(rows_a64 - 1));
                voff[i] = r * (int)(lda * 2) + chunk;
            } else {
                const int f = PAIRED ? v11_feature_of_row(row) : row;
                const int r = f < rows_w ? f : rows_w - 1;
                voff[i] = r * (int)(ldw * 2) + chunk;
            }
        }
        rs_a = v11_rsrc(A + m0 * lda);
        rs_w = v11_rsrc(Wt + (int64_t)n0 * ldw);
    };

    const int sw = (r16 >> 1) & 7;
    const int t3 = G ^ sw;
    const unsigned lds0 = (unsigned)(uintptr_t)(v11_lptr_t)smem;
    // per-lane fragment addresses in stage 0: k-step 0 (chunk t3) and k-step 1 (chunk t3 ^ 4)
    const unsigned pa0 = lds0 + (wm * 128 + r16) * 128 + (t3 << 4), pa1 = lds0 + (wm * 128 + r16) * 128 + ((t3 ^ 4) << 4);
    const unsigned pw0 = lds0 + V11_A_BYTES + (wn * 128 + r16) * 128 + (t3 << 4), pw1 = lds0 + V11_A_BYTES + (wn * 128 + r16) * 128 + ((t3 ^ 4) << 4);
    const int nk = K / V11_BK;                   // >= 2 (launcher)

    int pos = raster ? 0 : bid >> 3;
    if (raster ? p256 >= total : pos >= xcd_count) return;
#ifdef MG_AB_BUILD
    // measurement (flags 2048 / 4096): the workgroups of an XCD start their tile streams SPREAD OVER TIME — slot s of the XCD (0..31) s x 4 us late
    // (2048: 32 phases over ~one N = 5120 tile), or (s & 15) x 4 us (4096: 16 phases over half a tile).  All workgroups of a launch run the same
    // tiles in lock-step, so the 32 CUs of an XCD reach their epilogues in the same microseconds and queue at the XCD's fabric port (32 x 512 KiB of
    // fp32 read-modify-write per gated-residual tile); round 5's stagger delayed whole XCDs against each other, which leaves that queue as it is
    // (profiles/r05y5_gemm_stagger.log: no effect).
    if (flags & (2048 | 4096)) {
        const int units = (flags & 4096) ? (slot & 15) : slot;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles ~ 4 us
    }
#endif
    int64_t m0;
    int n0;
    tile_of(pos, m0, n0);
    set_offsets(m0, n0);
    const unsigned lds_pieces = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + prow0 * 128));
#define V12_PIECE_IMM(p) (((p) < 8 ? 0 : V11_A_BYTES) + ((p) & 7) * 1024)
#define V12_SB __builtin_amdgcn_sched_barrier(0)
    // 16 loads of one k-tile (byte offset kb along K) to the pieces at `base` (scalar): outside the k-loop (kernel start, tile prologue)
    auto cold_loads = [&](unsigned base, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 1" ::"s"(base), "s"(V12_PIECE_IMM(i)) : "scc", "memory");
            v11_dma(voff[i], i < 8 ? rs_a : rs_w, kb);
        }
    };
    bf16x8_t f0a[8], f0w[8], f1a[8], f1w[8];      // fragments of k-step 0 / 1
    // the 16 k-step-0 fragment reads of a k-tile, in the order the bodies' counted waits assume (the generator's NEXT_ORDER)
    auto read_first = [&](unsigned abn, unsigned wbn) __attribute__((always_inline)) {
        v11_rd<0>(f0w[0], wbn);
        v11_rd<0>(f0a[0], abn);
        v11_rd<2048>(f0a[1], abn);
        v11_rd<4096>(f0a[2], abn);
        v11_rd<6144>(f0a[3], abn);
        v11_rd<2048>(f0w[1], wbn);
        v11_rd<4096>(f0w[2], wbn);
        v11_rd<6144>(f0w[3], wbn);
        v11_rd<8192>(f0w[4], wbn);
        v11_rd<10240>(f0w[5], wbn);
        v11_rd<12288>(f0w[6], wbn);
        v11_rd<14336>(f0w[7], wbn);
        v11_rd<8192>(f0a[4], abn);
        v11_rd<10240>(f0a[5], abn);
        v11_rd<12288>(f0a[6], abn);
        v11_rd<14336>(f0a[7], abn);
        // outside the k-loop the compiler may copy a fragment register on the way into the loop: let the reads return first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // The stage toggles below are XORs with the stage size: the two stages must be one naturally aligned 128 KiB block (the kernel's only
    // LDS object sits at LDS address 0).
    if (lds0 & (2 * V11_STAGE - 1)) __builtin_trap();
    {   // kernel start: k-tiles 0 and 1 of the first tile -> stages 0 and 1.  The resources were just written by v_readfirstlane and the
        // loads are inline assembly: keep the 5 wait states by hand (gemm_bf16_v11.hip).
        asm volatile("s_nop 4" ::: "memory");
        cold_loads(lds_pieces, 0);
        cold_loads(lds_pieces + V11_STAGE, V11_BK * 2);
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");        // k-tile 0 has landed, for every wave
        V12_SB;
        read_first(pa0, pw0);
        V12_SB;
    }
    // Loop-carried state of the k-tile stream, ADVANCED INSIDE THE BODY by pinned one-instruction statements in MFMA gaps (V12_X_*, placed by
    // the generator behind each one's last use) — computed by the compiler at the top of the body they were ~10 instructions per k-tile with
    // the matrix pipe empty, and a body whose loop control grew by another dozen ran 7 % slower (profiles/r05i_gemm_v12_cont.log):
    unsigned ab1 = pa1, wb1 = pw1;                                  // k-step-1 fragment addresses in the stage being consumed
    unsigned abn = pa0 + V11_STAGE, wbn = pw0 + V11_STAGE;          // k-step-0 fragment addresses in the other stage (the next k-tile)
    unsigned lload = lds_pieces;                                    // scalar: this wave's pieces in the stage being consumed (refilled for k-tile + 2)
    int kb = 2 * (V11_BK * 2);                                      // scalar: byte offset along K of what the body loads
#define V12_X_ab1 asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(ab1))
#define V12_X_wb1 asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(wb1))
#define V12_X_abn asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(abn))
#define V12_X_wbn asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(wbn))
#define V12_X_lload asm volatile("s_xor_b32 %0, %0, 0x10000" : "+s"(lload) : : "scc")
#define V12_X_kb asm volatile("s_add_u32 %0, %0, 0x80" : "+s"(kb) : : "scc")
    static_assert(V11_STAGE == 0x10000 && V11_BK * 2 == 0x80, "the literals above");
#define V12_TA if (PROF) ta = __builtin_amdgcn_s_memtime()
#define V12_TB(k) if (PROF) pt[k] += __builtin_amdgcn_s_memtime() - ta
#define V12_BAR_LGKM do { V12_TA; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); V12_TB(0); } while (0)
#define V12_BAR_VM(n) do { V12_TA; asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); V12_TB(2); } while (0)
#define V12_M0(p) v11_set_m0<V12_PIECE_IMM(p)>(lload)
#define V12_G(p) v11_dma(voff[p], (p) < 8 ? rs_a : rs_w, kb)
    for (;;) {
        f32x4_t acc[8][8];       // [feature block of 16][token block of 16]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int next_pos = raster ? pos + 1 : pos + per_iter;
        const bool has_next = raster ? next_pos * 256 + p256 < total : next_pos < xcd_count;
        int64_t m0n = m0;
        int n0n = n0;
        // The tile's nk - 1 full bodies run as TWO passes through ONE copy of the body: nk - 2 bodies that load this tile's own k-tiles
        // (kb advances by one k-tile per body, inside the body), then — the offsets switched to the NEXT tile — one body that loads its
        // k-tile 0 (no next tile: a re-load of this very k-tile into its own stage, the same bytes).  Written as a conditional inside a single
        // loop, the switch cost every k-tile a vector compare, two branches and eight scalar moves (the resources' phi copies); as a second
        // textual copy of the body, the register allocator numbers the accumulators differently in each copy and moves them in between.
        int n_bodies = nk - 2, passes = 2;
        asm volatile("" : "+s"(passes));      // opaque: two passes through one loop, not two loops
#pragma nounroll
        for (int pass = 0; pass < passes; ++pass) {
#pragma nounroll
            for (int i = 0; i < n_bodies; ++i) {
                const unsigned long long tq = PROF ? __builtin_amdgcn_s_memtime() : 0;
                V12_SB;
                if constexpr (SCHED == 0) {
#include "gemm_bf16_v12_body_s0.inc"
                } else {
#include "gemm_bf16_v12_body_s2.inc"
                }
                if (PROF) pt[3] += __builtin_amdgcn_s_memtime() - tq, pt[4] += 1;
            }
            if (pass == 0) {
                kb = has_next ? 0 : (nk - 2) * (V11_BK * 2);
                if (has_next) {
                    tile_of(next_pos, m0n, n0n);
                    set_offsets(m0n, n0n);
                }
                n_bodies = 1;
            }
        }
        // The MFMAs above are inline asm: the compiler's hazard recognizer does not know their results are still in the matrix pipe.  Inside
        // the loop nothing else touches an accumulator; behind it the epilogue reads them, and the register allocator may renumber them for the
        // last k-tile's code copy: pad first.  (tools/audit_hot_loops.py checks that no accumulator instruction stands between the loop's last
        // MFMA and this pad.)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        V12_SB;
        const unsigned long long te = PROF ? __builtin_amdgcn_s_memtime() : 0;
        // the tile's last k-tile: MFMAs and the k-step-1 fragment reads only
        const unsigned stage_last = lload - lds_pieces;
        // Measured and not adopted (A/B library, flags & 8): TOUCH the 512 cache lines of `out` the wave's gated-residual epilogue will read (8 loads
        // of one dword per lane, each lane another 128-byte line) in front of the last k-tile.  The address processing of 64 lines per instruction
        // holds the wave's issue — last k-tile 4580 -> 16146 cycles — and the epilogue only drops from 39.1 to 35.6 thousand cycles: it does not
        // wait for HBM (profiles/r05y_gemm_touch.log: 5.70 against 5.50 ms).
        unsigned touch[8] = {};
#ifdef MG_AB_BUILD
        if constexpr (EPI == MG_EPI_GATE_RESID_F32) {
            if ((flags & 8) && m0 + wm * 128 + 128 <= M && n0 + wn * 128 + 128 <= N) {          // wave-uniform
                const u32x4_t rs_o = v11_rsrc((const float*)out + (m0 + wm * 128) * ldo + n0 + wn * 128);
                const int row_bytes = (int)ldo * 4;
                const int tvoff = (lane >> 2) * row_bytes + (lane & 3) * 128;
                asm volatile("s_nop 4" ::: "memory");          // the resource was just written by v_readfirstlane; the loads are opaque to the hazard recognizer
                // (per-load offsets in the VECTOR operand: a scalar offset may come out of a v_readlane spill slot, 5 wait states the assembler
                // statement would not get — tools/audit_hot_loops.py)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(touch[i]) : "v"(tvoff + i * 16 * row_bytes), "s"(rs_o) : "memory");
            }
        }
#endif
        V12_SB;
#include "gemm_bf16_v12_last.inc"
        if (PROF) { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); tsg = __builtin_amdgcn_s_memtime(); ps[0] += tsg - te; }
        // (builtin MFMAs: the compiler orders the epilogue's accumulator reads behind them.)  Every load has landed — the next tile's
        // k-tile 0 among them — and everyone is done reading this stage, the fp32 epilogues' transposition buffer.
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // (the touch loads' destination registers stay allocated until here)
        asm volatile("" :: "v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]), "v"(touch[4]), "v"(touch[5]), "v"(touch[6]), "v"(touch[7]));
        V12_SB;
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ps[1] += tt - tsg; tsg = tt; }
        // ---- epilogue (gemm_v11_common.h) ---- its row / column offsets from a fresh lane index (set_offsets: nothing lane-derived is carried across the k-loop in scratch)
        const int lane_e = FRESH ? v12_lane() : lane0, r16_e = lane_e & 15, G_e = lane_e >> 4;
        if constexpr (PAIRED) {
#ifdef MG_AB_BUILD
            if (flags & 1024)       // measurement: the tile's stores without the non-temporal hint
                v11_epilogue_pair<EPI, false>(acc, m0 + wm * 128, n0 + wn * 128, r16_e, G_e, M, N, bias, out, ldo);
            else
#endif
            // the tile is written once and read by another kernel: non-temporal stores (qkv +0.9 %, ffn.0 +1.6 %, N = 5120 unchanged: profiles/r05y4_gemm_pair_nt.log)
            v11_epilogue_pair<EPI, true>(acc, m0 + wm * 128, n0 + wn * 128, r16_e, G_e, (flags & 4) ? 0 : M, N, bias, out, ldo);      // flags & 4: measurement without the stores
        }
        else {
            const int64_t m_wave = m0 + wm * 128;
            const int n_wave = n0 + wn * 128;
            if (!(flags & 16) && m_wave + 128 <= M && n_wave + 128 <= N)      // the wave's whole 128 x 128 block exists (wave-uniform)
            {
                if (flags & 1)      // measurement: the round-4 form (each residual batch waited for with nothing else in flight)
                    v11_epilogue_rows<EPI>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
#ifdef MG_AB_BUILD
                else if ((flags & 4) && (flags & 128))      // measurement: neither residual loads nor stores / no stores / no residual loads
                    v11_epilogue_rows2<EPI, 3>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
                else if (flags & 4)
                    v11_epilogue_rows2<EPI, 2>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
                else if (flags & 128)
                    v11_epilogue_rows2<EPI, 1>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
                else if ((flags & 256) && (flags & 512))      // no nt hint on the stores and the residual loads / the stores / the loads
                    v11_epilogue_rows2<EPI, 12>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
                else if (flags & 256)
                    v11_epilogue_rows2<EPI, 4>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
                else if (flags & 512)
                    v11_epilogue_rows2<EPI, 8>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
#endif
                else
                    v11_epilogue_rows2<EPI>(acc, smem + stage_last + wave * 16384, lane_e, r16_e, G_e, m_wave, n_wave, bias, gate, out, ldo);
            }
            else
                mg_gemm_epilogue16<EPI, 8, 8>(acc, m_wave, n_wave, r16_e, G_e, (flags & 4) ? 0 : M, N, bias, gate, out, ldo);
        }
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ps[2] += tt - tsg; tsg = tt; }
        if (!has_next) {
            if (PROF) pe[1] += __builtin_amdgcn_s_memtime() - te, pe[2] += 1;
            break;
        }
        pos = next_pos;
        m0 = m0n;
        n0 = n0n;
        // ---- tile prologue: the next tile's k-tile 1 -> the stage the last k-tile was read from (and the fp32 epilogue has used); the
        // k-step-0 fragments of its k-tile 0 (landed: vmcnt(0) + barrier above); then the state steps over the last k-tile ----
        if constexpr (!PAIRED) asm volatile("s_barrier" ::: "memory");      // the other waves' transposition slices overlap this wave's pieces
        V12_SB;
        cold_loads(lload, V11_BK * 2);
        V12_SB;
        read_first(abn, wbn);
        V12_SB;
        V12_X_ab1; V12_X_wb1; V12_X_abn; V12_X_wbn; V12_X_lload;
        kb = 2 * (V11_BK * 2);
        if (PROF) pe[1] += __builtin_amdgcn_s_memtime() - te, pe[2] += 1, ps[3] += __builtin_amdgcn_s_memtime() - tsg;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA load may still be on its way when the workgroup's LDS is released
#undef V12_G
#undef V12_M0
#undef V12_BAR_VM
#undef V12_BAR_LGKM
#undef V12_TB
#undef V12_TA
#undef V12_X_kb
#undef V12_X_lload
#undef V12_X_wbn
#undef V12_X_abn
#undef V12_X_wb1
#undef V12_X_ab1
#undef V12_PIECE_IMM
#undef V12_SB
    if (PROF && lane == 0 && prof) {
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(prof + wave * 5 + i, pt[i]);
        pe[0] = __builtin_amdgcn_s_memtime() - t_start;
#pragma unroll
        for (int i = 0; i < 3; ++i) atomicAdd(prof + 32 + wave * 3 + i, pe[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(prof + 44 + wave * 4 + i, ps[i]);
        if (wave == 0) prof[64 + 2 * bid] = r_start, prof[64 + 2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int mg_gemm_v2_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M, int N, int K,
                      int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);      // gemm_bf16_v2.hip

#ifdef MG_AB_BUILD
static int g_v12_flags = 0;     // measurement bits (mg_gemm_set_variant(200 + flags)): 1 = fp32 outputs: residual batches not pipelined, 2 = raster 0 always, 4 = no stores (timing only), 8 = touch loads in front of the gated-residual epilogue, 128 = fp32 outputs: no residual loads (timing only), 256 / 512 = fp32 outputs: no nt hint on the stores / the residual loads, 1024 = bf16 outputs: no nt hint on the stores, 2048 / 4096 = the XCD's workgroups start 4 us apart in 32 / 16 phases, 16 = fp32 outputs: direct epilogue, 32 * (1 + s) = generated body s (0 / 2)
void mg_gemm_v12_set_flags(int f) { g_v12_flags = f; }
#else
static constexpr int g_v12_flags = 0;
#endif

int mg_gemm_v12_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                       int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st) {
    // one k-tile only, bf16 pitches that only allow 8-byte stores, strides past the 32-bit tile offsets: the general 256 x 128 kernel
    if (K < 2 * V11_BK || lda * 2 * 256 > 0x7fffffffLL || ldw * 2 * 256 > 0x7fffffffLL ||
        ldo * ((epilogue == MG_EPI_BIAS_BF16 || epilogue == MG_EPI_BIAS_GELU_BF16) ? 2 : 4) * 128 > 0x7fffffffLL ||
        ((epilogue == MG_EPI_BIAS_BF16 || epilogue == MG_EPI_BIAS_GELU_BF16) && (ldo & 7)))
        return mg_gemm_v2_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, st);
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;
    if (n_cu < 8) n_cu = 8;
    const int64_t tiles_m64 = (M + V11_BM - 1) / V11_BM;
    const int tiles_n = (N + V11_BN - 1) / V11_BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const int total = tiles_m * tiles_n;
    int nwg = n_cu;
    if (total < nwg) nwg = (total + 7) & ~7;
    const int raster = (nwg == 256 && tiles_n < 32 && !(g_v12_flags & 2)) ? (K > 8192 ? 1 : 3) : 0;      // variant 8's rule
    const dim3 grid((unsigned)nwg), block(V11_THREADS);
    // which generated body: measurement override in bits 5-7 of the flags (mg_gemm_set_variant(200 + 32 * (1 + s))), else body 0 (the six
    // bodies tried differ by < 1 %, profiles/r05c / r05g / r05h_gemm_v12_sched.log)
    const int sched = ((g_v12_flags >> 5) & 3) == 3 ? 2 : 0;
#ifdef MG_AB_BUILD
#define V12_PROF_BUF(P) ((P) ? g_gemm5_prof : nullptr)
#else
#define V12_PROF_BUF(P) nullptr
#endif
#define LAUNCH_S(E, S, P)                                                                                                                \
    hipLaunchKernelGGL((gemm_bf16_v12_kernel<E, S, P>), grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, gate, tiles_m, tiles_n, \
                       raster, g_v12_flags & ~0x60, V12_PROF_BUF(P))
#ifdef MG_AB_BUILD
#define LAUNCH(E) do { if (sched == 2) LAUNCH_S(E, 2, false); else LAUNCH_S(E, 0, false); } while (0)
    if (g_gemm5_prof && (epilogue == MG_EPI_BIAS_BF16 || epilogue == MG_EPI_GATE_RESID_F32)) {
        if (epilogue == MG_EPI_BIAS_BF16) LAUNCH_S(MG_EPI_BIAS_BF16, 0, true);
        else LAUNCH_S(MG_EPI_GATE_RESID_F32, 0, true);
        return mg_check_launch();
    }
#else
    (void)sched;
#define LAUNCH(E) LAUNCH_S(E, 0, false)
#endif
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
#undef LAUNCH_S
#undef V12_PROF_BUF
    return mg_check_launch();
}
