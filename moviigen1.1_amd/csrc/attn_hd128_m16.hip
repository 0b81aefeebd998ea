// Flash-style attention forward, head_dim 128 — schedule "m16": the one-wave-per-SIMD pipeline of attn_hd128_w64.hip
// rebuilt on v_mfma_f32_16x16x32_bf16.
//
// Why the other MFMA shape (experiments/mfma_shape_probe.hip): on random data this kernel is limited by
// the chip's POWER budget, not by issue cycles — cutting 11 % of the w64 kernel's cycles returned 4 % of throughput, the
// rest went into a lower clock.  16x16x32 does the same FLOPs with 8x less accumulator traffic (4 instead of 16
// registers read and written per instruction and lane): with an attention-like filler mix one wave per SIMD sustains
// 1565 TFLOP/s on it against 1420 on 32x32x16 (2.07 vs 1.83 GHz).  Register budget, LDS traffic (a fragment still feeds
// 64 queries), tile images, refill protocol and the software pipeline are those of w64; what changes is the tiling:
//
//   wave = 64 queries = 4 query blocks of 16; lane = (qi = lane & 15: the query inside a block, G = lane >> 4).
//   S^T (16 keys x 16 queries) = K.Q^T: A = K fragment [16 key rows][32 d], B = Q^T [32 d][16 queries]; a 32-key unit
//       = 2 key blocks (a, b) x 4 d chunks = 8 K fragments x 4 query blocks = 32 MFMAs.  Lane (qi, G) receives rows
//       4G..4G+3 of each block; the K tile stores its rows PERMUTED (mg_pack_kv_bf16) so that these are keys
//       8G..8G+3 (block a) and 8G+4..8G+7 (block b) of the unit and the 16 lanes of a read still touch 16 consecutive
//       16-byte slots (no bank conflicts):  tile row 32u + 16*block + i  <->  key 32u + 8*(i>>2) + 4*block + (i&3).
//   P (bf16) of a query block is then directly the B operand [32 keys][16 queries] of O^T = V^T.P^T: lane (qi, G)
//       holds keys 8G..8G+7 = [a0 a1 a2 a3 b0 b1 b2 b3], and the A operand V^T [16 d rows][32 keys] is one 16-byte read
//       of the V tile image (8 consecutive keys of one d): 8 V fragments (d blocks) x 4 query blocks = 32 MFMAs.
//   O^T: lane (qi, G) holds d = 16*db + 4G + r of its query: four consecutive d per d block, 8-byte stores.
//
// Softmax: FIXED per-row reference, folded into the MFMA.  A common factor 2^-m per row cancels in O / l, so the running
// maximum of the textbook online softmax only has to keep p representable — and P is bf16 (fp32's exponent range), O^T
// and l are fp32.  The reference of a row is the maximum of its scores against the FIRST key tile (64 keys, computed in
// the prologue anyway); -m_ref is the C operand of the first of the four MFMAs of every S^T accumulator (where round 3
// had the literal 0: the quad of a query block lives in 4 VGPRs, 16 per lane for the four blocks), so a score leaves
// the matrix pipe already relative to its row's reference: p = 2^(s*c) with no maximum, no subtraction, no rescale and
// no branch in the hot loop.  The reference is that maximum RAISED by 2^64: the row's true maximum is at least the first
// tile's, so l >= 2^-64 by construction (every term that matters stays a normal fp32 / bf16 number) and only grows,
// inf / NaN are sticky, so ONE test of the final row sum — at most 2^90 — covers every partial sum: a row passes unless
// some later key beats the first tile's best by more than 154 bits = 106 natural units.  A workgroup that fails the test sweeps its keys
// once for the TRUE row maxima (S^T only, K double-buffered) and repeats the same pipelined pass with those as the
// reference (then l is in [1, Lk] whatever the logits are); only inf / NaN scores still fail and take the plain exact
// loop, which also serves rows shorter than three full tiles.
// SCALED = false (mg_attn_fwd_bf16_hd128_prescaled, what WanModel.forward uses): q already carries c = scale*log2(e)
// (mg_rmsnorm_rope_bf16 out_scale: the factor enters before q's one rounding to bf16), a score IS its exponent and the
// softmax costs exp + add + half a cvt_pk per score.  SCALED = true (any q, any scale): one v_mul more per score, q is
// used as given (no second rounding); the reference is kept in raw score units either way.
#include <type_traits>
#include "common.h"
#include "../../include/moviigen_hip.h"

#define M16_THREADS 256
#define M16_QB 256
#define M16_TILE 16384
#define M16_K(slot) ((slot) * M16_TILE)
#define M16_V(slot) (3 * M16_TILE + (slot) * M16_TILE)

typedef const __attribute__((address_space(1))) void* m16_gptr_t;
typedef __attribute__((address_space(3))) void* m16_lptr_t;
MG_DEV bf16x8_t m16_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
struct M16State {
    f32x4_t ot[8][4];      // O^T [d block][query block]             (AGPRs: builtin MFMAs)
    f32x4_t st[2][2][4];   // S^T [unit kb][key block a/b][query block]  (arch VGPRs: inline-asm MFMAs)
    bf16x8_t pf[2][4];     // P   [unit kb][query block]: keys 8G..8G+7 of the unit
    f32x4_t mq[4];         // -m_ref of the lane's query in block n, four copies: C operand of an accumulator's first MFMA
    float m_run[4], l_run[4];
    float psa[4], psb[4];  // the pipelined pass's row-sum partials (even / odd score of a pair), folded into l_run ONCE behind the pass
    int bad;
};

template <int OFF>
MG_DEV void m16_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void m16_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// S^T accumulators live in ARCH VGPRs (the softmax reads them with VALU instructions): inline asm with "v" operands,
// as in w64.  A block is written by groups 0-3 (a) / 4-7 (b) of a step and first read by the NEXT step's softmax.
MG_DEV void m16_mfma_s0(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {       // acc = a.b + c
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c));
}
MG_DEV void m16_mfma_s(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {        // acc += a.b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
MG_DEV void m16_mfma_s_after_valu(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {   // start value written by VALU (mask)
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
constexpr int m16_koff(int i) { return (i >> 2) * 256 + (i & 3) * 4096; }   // K fragment i = (key block i>>2, d chunk i&3)
constexpr int m16_voff(int i) { return i * 256; }                           // V fragment i = d block

// exact softmax of unit KB (true maximum of the 32 keys, rescale of O^T and l), all four query blocks
template <int KB>
MG_DEV void m16_softmax_exact(M16State& s, float c) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float tmax = s.st[KB][0][n][0];
#pragma unroll
        for (int r = 1; r < 4; ++r) tmax = fmaxf(tmax, s.st[KB][0][n][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s.st[KB][1][n][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(s.m_run[n], tmax);
        float psum = 0.f;
        float p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            p[r] = __builtin_amdgcn_exp2f((s.st[KB][r >> 2][n][r & 3] - m_new) * c);
            psum += p[r];
        }
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[2 * e], p[2 * e + 1]);
        s.pf[KB][n] = m16_bf(w);
        const float alpha = __builtin_amdgcn_exp2f((s.m_run[n] - m_new) * c);
        s.l_run[n] = s.l_run[n] * alpha + psum;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) s.ot[d][n][e] *= alpha;
        s.m_run[n] = m_new;
    }
}

// softmax of unit KB whose scores are already relative to the row references (prologue): p = 2^s
template <int KB>
MG_DEV void m16_softmax_zero(M16State& s, float c) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float psum = 0.f;
        float p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            p[r] = __builtin_amdgcn_exp2f(s.st[KB][r >> 2][n][r & 3] * c);
            psum += p[r];
        }
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[2 * e], p[2 * e + 1]);
        s.pf[KB][n] = m16_bf(w);
        s.l_run[n] += psum;
    }
}

// accumulator start of S^T for a ragged tile: -1e30 on the key rows >= lim, else `base` of the query block (0 or the
// row reference); the MFMAs add K.Q^T to it
template <int KB>
MG_DEV void m16_mask_init(M16State& s, int lim, int G, const f32x4_t (&base)[4]) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = KB * 32 + G * 8 + blk * 4 + r;
                s.st[KB][blk][n][r] = key >= lim ? -1e30f : base[n][0];
            }
}

// One pipeline step = one 32-key unit.  MFMAs: S^T of unit KB (K fragments at lds_k) when SMODE != 0 (2: masked start,
// s.st[KB] pre-set by m16_mask_init) and P.V of unit KB (V fragments at lds_v, P = s.pf[KB]) when PV.  VALU: softmax
// of unit 1-KB when SM.  `dma(i)` is called once per group (i = 0..7).  Fragment ring: 4 K + 4 V registers, reads two
// groups (16 MFMAs) ahead; the reads of the first two groups must have been issued by the caller (prefetch()), the last
// two groups of this step issue them for the NEXT step from nk / nv (0 = the clamped address of this step: data unused).
// LKO / LVO / NKO / NVO: constant byte offsets added to lds_k / lds_v / nk / nv in the reads' immediate fields — the steady loop passes the
// SLOT's base address (a register carried across tiles) and names the unit inside the slot here: no address arithmetic per step.
template <int KB, int SMODE, bool PV, bool SM, bool SCALED, int ORD, int LKO = 0, int LVO = 0, int NKO = 0, int NVO = 0, typename Dma>
MG_DEV void m16_step(M16State& s, const bf16x8_t (&qf)[4][4], bf16x8_t (&kf)[4], bf16x8_t (&vf)[4], unsigned lds_k,
                     unsigned lds_v, unsigned nk, unsigned nv, float c, Dma dma) {
    constexpr int SB = 1 - KB;          // unit being exponentiated
    // (row-sum partials: s.psa / s.psb, carried through the whole pass — summed into l_run at the end of every step they were 9 VALU
    // instructions, 7 of them packed, at the bottom of each tile with the matrix pipe empty)
    u32x4_t w[4];
    float pa[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
    // softmax of pair pp = 2i + half (16 pairs per unit): query block n = pp >> 2, packed word pp & 3 = (key block,
    // register pair); the pieces below are placed one by one into the eight MFMA gaps of a group
    auto exp_a = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = 2 * i + half, n = pp >> 2, wd = pp & 3;
            const float x = s.st[SB][wd >> 1][n][(wd & 1) * 2];
            pa[half] = __builtin_amdgcn_exp2f(SCALED ? x * c : x);
            asm volatile("" : "+v"(pa[half]));     // opaque use: pins the work HERE (LLVM sinks it otherwise)
        }
    };
    auto exp_b = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = 2 * i + half, n = pp >> 2, wd = pp & 3;
            const float x = s.st[SB][wd >> 1][n][(wd & 1) * 2 + 1];
            pb[half] = __builtin_amdgcn_exp2f(SCALED ? x * c : x);
            asm volatile("" : "+v"(pb[half]));
        }
    };
    auto add_a = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int n = (2 * i + half) >> 2;
            s.psa[n] += pa[half];
            asm volatile("" : "+v"(s.psa[n]));
        }
    };
    auto add_b = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int n = (2 * i + half) >> 2;
            s.psb[n] += pb[half];
            asm volatile("" : "+v"(s.psb[n]));
        }
    };
    auto cvt = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = 2 * i + half, n = pp >> 2, wd = pp & 3;
            unsigned pk = pack_bf2(pa[half], pb[half]);
            asm volatile("" : "+v"(pk));
            w[n][wd] = pk;
        }
    };
    auto S = [&](int i, int n) __attribute__((always_inline)) {
        const int blk = i >> 2, c = i & 3, r = i & 3;
        if (SMODE == 1 && c == 0) m16_mfma_s0(s.st[KB][blk][n], kf[r], qf[n][c], s.mq[n]);
        else if (SMODE == 2 && c == 0) m16_mfma_s_after_valu(s.st[KB][blk][n], kf[r], qf[n][c]);
        else if (SMODE != 0) m16_mfma_s(s.st[KB][blk][n], kf[r], qf[n][c]);
    };
    auto P = [&](int i, int n) __attribute__((always_inline)) {
        if (PV) s.ot[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[i & 3], s.pf[KB][n], s.ot[i][n], 0, 0, 0);
    };
    // ring slot (i+2)&3 was consumed two groups ago: it takes the read for group i+2.  (A ds_read whose result nobody
    // uses would leave its destination free for reuse while the data is still on its way: steps without S^T MFMAs keep
    // the previous occupant alive up to the read.)
    auto rdK = [&](int i) __attribute__((always_inline)) {
        const int r2 = (i + 2) & 3;
        if (SMODE == 0) asm volatile("" ::"v"(kf[r2]));
        if (i < 6) {
            switch (i) {   // compile-time after unrolling
                case 0: m16_rd<m16_koff(2) + LKO>(kf[r2], lds_k); break;
                case 1: m16_rd<m16_koff(3) + LKO>(kf[r2], lds_k); break;
                case 2: m16_rd<m16_koff(4) + LKO>(kf[r2], lds_k); break;
                case 3: m16_rd<m16_koff(5) + LKO>(kf[r2], lds_k); break;
                case 4: m16_rd<m16_koff(6) + LKO>(kf[r2], lds_k); break;
                default: m16_rd<m16_koff(7) + LKO>(kf[r2], lds_k); break;
            }
        } else if (i == 6) m16_rd<m16_koff(0) + NKO>(kf[r2], nk);
        else m16_rd<m16_koff(1) + NKO>(kf[r2], nk);
    };
    auto rdV = [&](int i) __attribute__((always_inline)) {
        const int r2 = (i + 2) & 3;
        if (i < 6) {
            switch (i) {
                case 0: m16_rd<m16_voff(2) + LVO>(vf[r2], lds_v); break;
                case 1: m16_rd<m16_voff(3) + LVO>(vf[r2], lds_v); break;
                case 2: m16_rd<m16_voff(4) + LVO>(vf[r2], lds_v); break;
                case 3: m16_rd<m16_voff(5) + LVO>(vf[r2], lds_v); break;
                case 4: m16_rd<m16_voff(6) + LVO>(vf[r2], lds_v); break;
                default: m16_rd<m16_voff(7) + LVO>(vf[r2], lds_v); break;
            }
        } else if (i == 6) m16_rd<m16_voff(0) + NVO>(vf[r2], nv);
        else m16_rd<m16_voff(1) + NVO>(vf[r2], nv);
    };
#define M16_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        m16_wait<2>();                         // K(i) and V(i) landed; younger: K(i+1), V(i+1)
        M16_SB();
        // Eight MFMAs S0 P0 S1 P1 S2 P2 S3 P3 (asm / builtin alternate) and the gaps behind them:
        //     [exp exp] [add add cvt] [rdK] [rdV] [exp exp] [add add cvt] [dma] [-]
        // The "balanced" placement [exp rdK][exp add][add cvt rdV][exp][exp add][add cvt][dma][-] (never more than
        // three fillers behind an MFMA) was measured: 2765 instead of 2652 cycles per tile, 1472 instead of 1505 TFLOP/s
        // (profiles/r03g_attn_filler_schedule.log) — the exps want to sit together in front of a builtin MFMA.
        // ORD = where the fillers stand (a measurement parameter; the kernel launches ONE of them).  A v_exp_f32 keeps the wave's VALU for
        // four passes: in-order issue parks whatever follows a second VALU instruction of the same gap — including the next MFMA.
        if constexpr (ORD == 0) {
            S(i, 0); M16_SB();
            exp_a(i, 0); exp_b(i, 0); M16_SB();
            P(i, 0); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            S(i, 1); M16_SB();
            rdK(i); M16_SB();
            P(i, 1); M16_SB();
            rdV(i); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); exp_b(i, 1); M16_SB();
            P(i, 2); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            S(i, 3); M16_SB();
            dma(i); M16_SB();
            P(i, 3); M16_SB();
        } else if constexpr (ORD == 1) {          // one exp per gap, alone:  [e][e][a a c][rK rV][e][e][a a c][dma]
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            P(i, 0); M16_SB();
            exp_b(i, 0); M16_SB();
            S(i, 1); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            P(i, 1); M16_SB();
            rdK(i); M16_SB(); rdV(i); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 2); M16_SB();
            exp_b(i, 1); M16_SB();
            S(i, 3); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            P(i, 3); M16_SB();
            dma(i); M16_SB();
        } else if constexpr (ORD == 2) {          // as 1 with the fragment reads beside the exps:  [e rK][e rV][a a c][-][e][e][a a c][dma]
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB(); rdK(i); M16_SB();
            P(i, 0); M16_SB();
            exp_b(i, 0); M16_SB(); rdV(i); M16_SB();
            S(i, 1); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            P(i, 1); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 2); M16_SB();
            exp_b(i, 1); M16_SB();
            S(i, 3); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            P(i, 3); M16_SB();
            dma(i); M16_SB();
        } else if constexpr (ORD == 4) {          // as 1 with the conversion beside the reads:  [e][e][a a][c rK rV][e][e][a a][c dma]
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            P(i, 0); M16_SB();
            exp_b(i, 0); M16_SB();
            S(i, 1); M16_SB();
            add_a(i, 0); add_b(i, 0); M16_SB();
            P(i, 1); M16_SB();
            cvt(i, 0); M16_SB(); rdK(i); M16_SB(); rdV(i); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 2); M16_SB();
            exp_b(i, 1); M16_SB();
            S(i, 3); M16_SB();
            add_a(i, 1); add_b(i, 1); M16_SB();
            P(i, 3); M16_SB();
            cvt(i, 1); M16_SB(); dma(i); M16_SB();
        } else if constexpr (ORD == 5) {          // as 1 with the two fragment reads in different gaps:  [e][e][a a c][rK][e][e][a a c][rV dma]
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            P(i, 0); M16_SB();
            exp_b(i, 0); M16_SB();
            S(i, 1); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            P(i, 1); M16_SB();
            rdK(i); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 2); M16_SB();
            exp_b(i, 1); M16_SB();
            S(i, 3); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            P(i, 3); M16_SB();
            rdV(i); M16_SB(); dma(i); M16_SB();
        } else if constexpr (ORD == 8) {          // placement 1's gaps, the MFMAs in PAIRS that share their A operand (S0 S1 P0 P1 S2 S3 P2 P3): an energy experiment
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            S(i, 1); M16_SB();
            exp_b(i, 0); M16_SB();
            P(i, 0); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            P(i, 1); M16_SB();
            rdK(i); M16_SB(); rdV(i); M16_SB();
            S(i, 2); M16_SB();
            exp_a(i, 1); M16_SB();
            S(i, 3); M16_SB();
            exp_b(i, 1); M16_SB();
            P(i, 2); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            P(i, 3); M16_SB();
            dma(i); M16_SB();
        } else if constexpr (ORD == 9) {          // placement 1's gaps, the MFMAs in QUADS that share their A operand (S0 S1 S2 S3 P0 P1 P2 P3)
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            S(i, 1); M16_SB();
            exp_b(i, 0); M16_SB();
            S(i, 2); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            S(i, 3); M16_SB();
            rdK(i); M16_SB(); rdV(i); M16_SB();
            P(i, 0); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 1); M16_SB();
            exp_b(i, 1); M16_SB();
            P(i, 2); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            P(i, 3); M16_SB();
            dma(i); M16_SB();
        } else {                                  // the four exps in four consecutive gaps:  [e][e][e][e][a a c][a a c][rK rV][dma]
            S(i, 0); M16_SB();
            exp_a(i, 0); M16_SB();
            P(i, 0); M16_SB();
            exp_b(i, 0); M16_SB();
            S(i, 1); M16_SB();
            exp_a(i, 1); M16_SB();
            P(i, 1); M16_SB();
            exp_b(i, 1); M16_SB();
            S(i, 2); M16_SB();
            add_a(i, 0); add_b(i, 0); cvt(i, 0); M16_SB();
            P(i, 2); M16_SB();
            add_a(i, 1); add_b(i, 1); cvt(i, 1); M16_SB();
            S(i, 3); M16_SB();
            rdK(i); M16_SB(); rdV(i); M16_SB();
            P(i, 3); M16_SB();
            dma(i); M16_SB();
        }
    }
#undef M16_SB
    if (SM) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s.pf[SB][n] = m16_bf(w[n]);
        }
    }
}

// the lane index, re-derived where it is called (two VALU instructions the compiler can neither hoist nor merge).  Everything that depends only on
// the lane — LDS read bases, query rows, output addresses — is loop-invariant over the (head, query block) items AND over the tiles of an item: derived
// from threadIdx once, the compiler computes all of it at kernel entry (43 registers), keeps it in scratch, and reloads it value by value, each behind
// its own s_waitcnt vmcnt(0), at the start of an item, between the steps of the drain and in front of every output row (~45 round trips per item).
MG_DEV int m16_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

struct M16NoDma {
    __device__ __forceinline__ void operator()(int) const {}
};

template <bool PROF, bool SCALED, int ORD>
__global__ __launch_bounds__(M16_THREADS, 1) void attn_hd128_m16_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp,
    uint16_t* __restrict__ o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, int dbg,
    unsigned long long* __restrict__ prof, float* __restrict__ lse, unsigned* __restrict__ flagcnt, unsigned* __restrict__ tickets) {
    __shared__ __attribute__((aligned(16))) char smem[6 * M16_TILE];
    __shared__ int s_next_item;
    const int bid = blockIdx.x;
    const unsigned long long r_start = PROF ? __builtin_amdgcn_s_memrealtime() : 0;     // the 100 MHz counter all workgroups share
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent work loop over (head, query block) items, head-major.  Items are handed out by TICKET (`tickets` != nullptr): the first nwg
    // items by workgroup index, every further one by an atomic counter, fetched one item ahead.  The static partition it replaces for long launches — XCD x owns
    // items [x, x + 1) * total / 8 — lost 2.0 % of the metric's launch to its tail: the 32 workgroups of an XCD end within 2 us of each other,
    // but the XCDs do not run at one speed (223.7 ms for the slowest, 211.0 ms for the fastest: profiles/r05v_attn_balance.log).  In ticket
    // order the XCD's workgroups still take neighbouring query blocks of one head at nearly the same moment and walk its K / V together.
    const int total_items = nqb * heads;
    const int nwg = gridDim.x;
    const bool ticketed = tickets != nullptr && nwg != total_items;
    int item, item_end, item_step;
    if (nwg == total_items || ticketed) {
        item = bid, item_end = ticketed ? total_items : bid + 1, item_step = 1;
    } else {            // the static, XCD-contiguous partition: launches of fewer than 32 rounds (mg_attn_m16_launch)
        const int xcd = bid & 7, slot = bid >> 3;
        item = (int)((int64_t)xcd * total_items / 8) + slot;
        item_end = (int)((int64_t)(xcd + 1) * total_items / 8);
        item_step = nwg >> 3;           // host guarantees nwg % 8 == 0 here
    }
    unsigned next_ticket = 0;
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0;      // PROF: per-item phases (wave 0): {Q loads + first K / V tiles landed, first tile's S / reference / softmax / step 1, wait for the refills, steady loop, drain, epilogue, items}
    unsigned long long pf_fence = 0, pf_a = 0, pf_b = 0, pf_n = 0;      // PROF: the steady loop's per-tile parts, summed over the items (stored once, at the end: an atomic per item would sit in the drain's vmcnt wait)
    while (item < item_end) {
    if (ticketed && tid == 0) next_ticket = atomicAdd(tickets, 1u);      // the item after this one; the answer has 2.8 ms to arrive
    const int head = item / nqb;
    const int qb0 = item - head * nqb;
    const int lane = m16_lane();
    int qi = lane & 15, G = lane >> 4;       // (re-derived behind the steady loop and in front of the output rows: m16_lane)
    __syncthreads();                    // the previous item's last LDS reads are done before this one's DMA
    if (PROF) ph_t = __builtin_amdgcn_s_memtime();

    // Q fragments (B operand): query block n of this wave = rows 64*wave + 16*n + qi, d = 32*c + 8*G .. +7
    bf16x8_t qf[4][4];
    const int64_t qrow_base = (int64_t)qb0 * M16_QB + wave * 64 + qi;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int64_t qr = qrow_base + n * 16;
        const int64_t qrow = qr < Lq ? qr : Lq - 1;
        const uint16_t* qp = q + qrow * ldq + head * 128 + G * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[n][c] = m16_bf(*(const u32x4_t*)(qp + c * 32));
    }
    const int T = (int)((Lk + 63) / 64);
    const int last_lim = (int)(Lk - (int64_t)(T - 1) * 64);     // keys in the last tile, 1..64
    // LDS-DMA: a tile is 16 pieces of 1 KiB; wave w moves pieces 4w..4w+3 of the K tile and of the V tile
    const int nfull = last_lim == 64 ? T : T - 1;
    // tile indices past the end are clamped (a redundant reload of the last tile into a free slot) instead of guarded
    const unsigned lds0 = (unsigned)(uintptr_t)(m16_lptr_t)smem;
    // (round 6) EVERY LDS-DMA piece of the kernel is a buffer load behind the head's K / V resource: per-lane offset `dvo` (ONE register, the same for all
    // pieces), the tile's byte offset a scalar, the LDS destination in M0, the pieces' 1 KiB steps in the instruction offset (global and LDS side alike).  The
    // global_load_lds form this replaces outside the steady loop needed a 64-bit per-lane address per piece: the compiler hoisted those out of the item loop,
    // spilled them, and reloaded them one by one — scratch_load, s_waitcnt vmcnt(0), load, 8 to 16 times in a row at the start of every item.
    const uint64_t kb64 = (uint64_t)(uintptr_t)(kp + ((int64_t)head * T) * 8192), vb64 = (uint64_t)(uintptr_t)(vp + ((int64_t)head * T) * 8192);
    const u32x4_t rs_k = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kb64), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(kb64 >> 32) & 0xffffu)),
                          (unsigned)T * M16_TILE, 0x00020000u};
    const u32x4_t rs_v = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vb64), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(vb64 >> 32) & 0xffffu)),
                          (unsigned)T * M16_TILE, 0x00020000u};
    const int dvo = wave * 4096 + lane * 16;      // this lane's 16 bytes of piece 0 inside a tile image
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 4096));      // this wave's piece 0 in K slot 0
    // this wave's four pieces of tile t (clamped: a redundant reload of the last tile into a slot nobody reads) -> LDS at byte `lbase` (wave-uniform)
    auto dma_tile4 = [&](const u32x4_t& rs, int t, unsigned lbase) __attribute__((always_inline)) {
        const int soff = __builtin_amdgcn_readfirstlane((t < T ? t : T - 1) * M16_TILE);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\t"      // (5 wait states: the resource / offset SGPRs may come straight from v_readfirstlane, and the loads are opaque to the hazard recognizer)
                     "buffer_load_dwordx4 %0, %1, %3 offen lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %3 offen offset:1024 lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %3 offen offset:2048 lds\n\t"
                     "buffer_load_dwordx4 %0, %1, %3 offen offset:3072 lds"
                     :: "v"(dvo), "s"(rs), "s"(lbase), "s"(soff) : "memory");
    };
    auto dma_k4 = [&](int t, int slot) __attribute__((always_inline)) { dma_tile4(rs_k, t, lds_w + M16_K(slot)); };
    auto dma_v4 = [&](int t, int slot) __attribute__((always_inline)) { dma_tile4(rs_v, t, lds_w + M16_V(slot)); };
    unsigned kbase = lds0 + G * 1024 + qi * 16;                  // + M16_K(slot) + kb*512 + blk*256 + c*4096
    unsigned vbase = lds0 + 3 * M16_TILE + G * 2048 + qi * 16;   // + slot*TILE + kb*8192 + db*256
    auto k_addr = [&](int slot, int kb) __attribute__((always_inline)) { return kbase + slot * M16_TILE + kb * 512; };
    auto v_addr = [&](int slot, int kb) __attribute__((always_inline)) { return vbase + slot * M16_TILE + kb * 8192; };

    M16State s;
    bf16x8_t kf[4], vf[4];
    auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int n = 0; n < 4; ++n) s.ot[d][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 4; ++n) s.m_run[n] = -1e30f, s.l_run[n] = 0.f, s.psa[n] = 0.f, s.psb[n] = 0.f;
        s.bad = 0;
    };
    reset();
    // Q fragments live in the ACCUMULATOR half of the register file (MFMA B operands may be read from there; the VALU never touches them): the 64
    // arch VGPRs this frees are what the register allocator needs to get into and out of the steady loop without swapping ~60 values through scratch
    // (profiles/r06p_attn_q_agpr.log: "first tile" 19.6 -> 9.9 thousand cycles per item, the 512-key launch 3.10 -> 2.64 ms, the self-attention
    // launch +0.3 %, same bits).  They are pinned there BEHIND the first K / V tiles' requests (q_pin, after the first fence of either pass): pinned
    // here, the wave waited for its 16 Q loads before it asked for the tiles — two memory round trips in a row at the start of every item.
    auto q_pin = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" : "+a"(qf[n][c]));
    };
    auto fence = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // hot-loop form: this wave's LDS-DMA pieces have landed, then the barrier — and nothing else (no lgkmcnt(0): four
    // fragment reads are in flight across the barrier by design)
    auto fence_hot = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto prefetch = [&](unsigned ak, unsigned av) __attribute__((always_inline)) {
        m16_rd<m16_koff(0)>(kf[0], ak);
        m16_rd<m16_voff(0)>(vf[0], av);
        m16_rd<m16_koff(1)>(kf[1], ak);
        m16_rd<m16_voff(1)>(vf[1], av);
    };
    const f32x4_t zero4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // bare S^T of one unit (prologue / reference sweep / exact loop): 32 MFMAs, plain loads, RAW scores
    auto bare_S = [&](auto kbc, int slot, int lim) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        if (lim < 64) m16_mask_init<KB>(s, lim, G, zero4);
        else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int n = 0; n < 4; ++n) s.st[KB][blk][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        const char* base = smem + M16_K(slot) + G * 1024 + qi * 16 + KB * 512;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t f = *(const bf16x8_t*)(base + m16_koff(i));
#pragma unroll
            for (int n = 0; n < 4; ++n)
                s.st[KB][i >> 2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, qf[n][i & 3], s.st[KB][i >> 2][n], 0, 0, 0);
        }
    };
    auto bare_PV = [&](auto kbc, int slot) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        const char* base = smem + M16_V(slot) + G * 2048 + qi * 16 + KB * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t f = *(const bf16x8_t*)(base + m16_voff(i));
#pragma unroll
            for (int n = 0; n < 4; ++n)
                s.ot[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, s.pf[KB][n], s.ot[i][n], 0, 0, 0);
        }
    };
    using KB0 = std::integral_constant<int, 0>;
    using KB1 = std::integral_constant<int, 1>;

    bool exact_pass = nfull < 3;
    if (!exact_pass) {
    for (int attempt = 0;; ++attempt) {
        // per-row maximum of the RAW scores in s.st[0..1] (the 64 keys of one tile), folded into mx
        auto tile_max = [&](float (&mx)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float a = fmaxf(fmaxf(s.st[0][0][n][0], s.st[0][0][n][1]), fmaxf(s.st[0][0][n][2], s.st[0][0][n][3]));
                float b = fmaxf(fmaxf(s.st[0][1][n][0], s.st[0][1][n][1]), fmaxf(s.st[0][1][n][2], s.st[0][1][n][3]));
                float c2 = fmaxf(fmaxf(s.st[1][0][n][0], s.st[1][0][n][1]), fmaxf(s.st[1][0][n][2], s.st[1][0][n][3]));
                float d2 = fmaxf(fmaxf(s.st[1][1][n][0], s.st[1][1][n][1]), fmaxf(s.st[1][1][n][2], s.st[1][1][n][3]));
                mx[n] = fmaxf(mx[n], fmaxf(fmaxf(a, b), fmaxf(c2, d2)));
            }
        };
        auto set_reference = [&](const float (&mx)[4], float above) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {       // a query's keys are spread over the four G lanes
                float m = fmaxf(mx[n], __shfl_xor(mx[n], 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64)) + above;
                s.mq[n] = (f32x4_t){-m, -m, -m, -m};
            }
        };
        if (attempt) {
            // ------------------------------------------------------------------------------------------
            // reference sweep (only after a flagged pass): TRUE row maxima.  S^T of every tile (64 MFMAs
            // per wave and tile), one barrier per tile, no V, no exponentials.
            // ------------------------------------------------------------------------------------------
            float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
            // K tiles ride a ring of all six 16 KiB slots, four tiles ahead (tile t in slot t % 6; indices past the end
            // are clamped: a redundant reload into a slot nobody reads), so a tile's DMA latency is covered by the S^T of
            // three others; vmcnt(12) = this wave's pieces of the OLDEST of the four tiles in flight have landed
#pragma unroll
            for (int a = 0; a < 4; ++a) dma_k4(a, a);
            for (int t = 0; t < T; ++t) {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                __syncthreads();                // K(t) visible to all; everyone is past S(t-1): slot (t+4) % 6 = (t-2) % 6 is free
                dma_k4(t + 4, (t + 4) % 6);
                const int lim = t == T - 1 ? last_lim : 64;
                bare_S(KB0{}, t % 6, lim);
                bare_S(KB1{}, t % 6, lim);
                tile_max(mx);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the clamped tail loads must not land under the pass below
            set_reference(mx, 0.f);
            __syncthreads();                    // the last S reads are done before the pass below refills slots 0 / 1
            reset();
        }
        // ------------------------------------------------------------------------------------------
        // pipelined pass.  Iteration t = steps u = 2t (S(t,1) | P.V(t-1,1) | softmax S(t,0)) and
        // u = 2t+1 (S(t+1,0) | P.V(t,0) | softmax S(t,1)); tile t in slot t % 3.  Needs >= 3 FULL
        // tiles to have a steady state; shorter or all-ragged rows go straight to the exact loop.
        // ------------------------------------------------------------------------------------------
        dma_k4(0, 0), dma_v4(0, 0), dma_k4(1, 1);
        fence();
        q_pin();
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[0] += tt - ph_t; ph_t = tt; }
        dma_k4(2, 2), dma_v4(1, 1);     // iteration 0's refill
        bare_S(KB0{}, 0, 64);
        bare_S(KB1{}, 0, 64);
        if (attempt == 0) {
            // first attempt: reference = the row's maximum over tile 0, RAISED by 2^64.  The true row maximum is at least
            // tile 0's, so the largest term of a row is at least 2^-64 (every term that matters stays a normal fp32 /
            // bf16 number) and the headroom above grows to 2^(90+64): the pass is good for rows whose best key beats
            // the best of the first 64 by up to 154 bits = 106 natural units
            float mx[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
            tile_max(mx);
            set_reference(mx, c_log2 > 1e-20f ? 64.f / c_log2 : 0.f);       // (raw score units: an exponent is score * c)
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)          // tile 0 was computed raw: make it relative like every later tile
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int n = 0; n < 4; ++n) s.st[kb][blk][n] += s.mq[n];
        m16_softmax_zero<0>(s, c_log2);
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(s.l_run[n]));     // summed HERE: left alone, the compiler sinks these 32 adds behind the steady loop (their
        //                                                                      first use) and carries the 32 addends across it — through scratch
        // step u = 1: S(1,0) | P.V(0,0) | softmax S(0,1)
        prefetch(k_addr(1, 0), v_addr(0, 0));
        m16_step<0, 1, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(1, 0), v_addr(0, 0), k_addr(1, 1), v_addr(0, 1), c_log2, M16NoDma());
        int t = 1;
        // ---- steady loop, with its state CARRIED instead of recomputed (round 5).  Per tile the loop used to spend ~35 instructions outside
        // MFMA gaps — slot indices -> eight LDS read addresses, two 64-bit global addresses with their clamps, M0 values — clumped at the top of
        // the tile and in front of the first K / V piece, each with the matrix pipe empty (one wave per SIMD).  Now:
        //   * the read addresses of slots (tile t-1, t, t+1) live in three registers per operand and ROTATE by two v_swap_b32 each, in the
        //     last gap of step B; the unit inside a slot (+512 / +8192 bytes) is in the reads' immediate fields (m16_step LKO ...);
        //   * the refill loads are BUFFER loads: the head's K / V images behind two resources (num_records = the head's bytes: a tile index past
        //     the end reads zeros into a slot nobody reads — no clamp), one per-lane offset per piece, the tile's byte offset in a scalar that
        //     advances by 16 KiB in a gap; the LDS destination = a scalar per slot, rotating with three s_mov in a gap;
        //   * M0 is saved ONCE in front of the loop and restored ONCE behind it: the compiler tracks M0 for its own (builtin) LDS-DMA loads of the
        //     prologue and the tail — it hoists and merges identical M0 writes — and must find the register as it left it.  (Saved and restored
        //     inside every load's statement, the restore stood directly behind the load and waited for it: 110 cycles per piece.)
        unsigned ka0 = k_addr(0, 0), ka1 = k_addr(1, 0), ka2 = k_addr(2, 0);      // K read bases of the slots of tiles t-1, t, t+1
        unsigned va0 = v_addr(0, 0), va1 = v_addr(1, 0), va2 = v_addr(2, 0);
        unsigned lk0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + M16_K(0) + wave * 4096));      // this wave's piece 0 in those slots (K; V: + 3 tiles)
        unsigned lk1 = lk0 + M16_TILE, lk2 = lk0 + 2 * M16_TILE;
        int offk = (t + 2) * M16_TILE, offv = (t + 1) * M16_TILE;                 // byte offsets of tiles t+2 (K) / t+1 (V) in the head's image
        unsigned keep_m0, rot_tmp;
#define M16_BDMA0(VOFF, RS, SOFF, LBASE, IMM)                                                                                              \
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %4 offen lds"                                             \
                 :: "v"(VOFF), "s"(RS), "s"(LBASE), "n"(IMM), "s"(SOFF) : "memory", "scc")
#define M16_BDMAN(VOFF, RS, SOFF, IMM)                                                                                                     \
    asm volatile("buffer_load_dwordx4 %0, %1, %3 offen offset:%2 lds" :: "v"(VOFF), "s"(RS), "n"(IMM), "s"(SOFF) : "memory")
        asm volatile("s_mov_b32 %0, m0\n\ts_nop 4" : "=s"(keep_m0) :: "memory");      // the resources were just written by v_readfirstlane; the loads below are opaque to the hazard recognizer
        // The carried registers may be scratch reloads of the preheader: with no VMEM instruction of its own in the loop (the loads are opaque
        // asm) the compiler's wait-count pass would carry "reload pending" around the back edge and wait vmcnt(1) in front of a fragment read in
        // the MIDDLE of step A — i.e. for the refill loads just issued (measured: step A 1275 -> 2150 cycles).  A wait it can see, out here:
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[1] += tt - ph_t; ph_t = tt; }
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
        asm volatile("" : "+v"(ka0), "+v"(ka1), "+v"(ka2), "+v"(va0), "+v"(va1), "+v"(va2));
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[2] += tt - ph_t; ph_t = tt; }
        for (; t + 1 < nfull; ++t) {
            const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            fence_hot();                // K(t+1), V(t) visible; everyone is past iteration t-1
            const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t: S(t,1) [K slot of t] | P.V(t-1,1) [V slot of t-1] | softmax S(t,0); refill K(t+2) -> slot of t-1, V(t+1) -> slot of t+1
            m16_step<1, 1, true, true, SCALED, ORD, 512, 8192, 0, 0>(s, qf, kf, vf, ka1, va0, ka2, va1, c_log2,
                                       [&](int n) __attribute__((always_inline)) {   // all 8 refill pieces here (step B gives them time to land)
                                           switch (n) {
                                               case 0: M16_BDMA0(dvo, rs_k, offk, lk0, 0); break;       // M0 once per operand: the pieces'
                                               case 1: M16_BDMAN(dvo, rs_k, offk, 1024); break;         // 1 KiB steps are instruction offsets
                                               case 2: M16_BDMAN(dvo, rs_k, offk, 2048); break;         // (global and LDS side alike)
                                               case 3: M16_BDMAN(dvo, rs_k, offk, 3072); break;
                                               case 4: M16_BDMA0(dvo, rs_v, offv, lk2, 3 * M16_TILE); break;
                                               case 5: M16_BDMAN(dvo, rs_v, offv, 1024); break;
                                               case 6: M16_BDMAN(dvo, rs_v, offv, 2048); break;
                                               default: M16_BDMAN(dvo, rs_v, offv, 3072); break;
                                           }
                                       });
            const unsigned long long c2 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t+1: S(t+1,0) [K slot of t+1] | P.V(t,0) [V slot of t] | softmax S(t,1); in its free gaps the state steps to tile t+1
            m16_step<0, 1, true, true, SCALED, ORD, 0, 0, 512, 8192>(s, qf, kf, vf, ka2, va1, ka2, va1, c_log2,
                                       [&](int n) __attribute__((always_inline)) {
                                           if (n == 0)
                                               asm volatile("s_add_u32 %0, %0, 0x4000\n\ts_add_u32 %1, %1, 0x4000" : "+s"(offk), "+s"(offv) : : "scc");
                                           else if (n == 1)      // slots (t-1, t, t+1) -> (t, t+1, t-1)
                                               asm volatile("s_mov_b32 %3, %0\n\ts_mov_b32 %0, %1\n\ts_mov_b32 %1, %2\n\ts_mov_b32 %2, %3"
                                                            : "+s"(lk0), "+s"(lk1), "+s"(lk2), "=&s"(rot_tmp));
                                           else if (n == 7) {    // behind the step's last fragment reads: rotate the read bases the same way, in place
                                               asm volatile("v_swap_b32 %0, %1\n\tv_swap_b32 %1, %2" : "+v"(ka0), "+v"(ka1), "+v"(ka2));
                                               asm volatile("v_swap_b32 %0, %1\n\tv_swap_b32 %1, %2" : "+v"(va0), "+v"(va1), "+v"(va2));
                                           }
                                       });
            if (PROF) {
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                pf_fence += c1 - c0, pf_a += c2 - c1, pf_b += c3 - c2, pf_n += 1;
            }
        }
#undef M16_BDMA0
#undef M16_BDMAN
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep_m0) : "memory");
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[3] += tt - ph_t; ph_t = tt; }
        static_assert(M16_TILE == 0x4000, "the literal above");
        const int s0 = (t - 1) % 3, s1 = t % 3, s2 = (t + 1) % 3;     // slots of tiles t-1, t, t+1 (tile i lives in slot i % 3)
        {       // the drain's read bases and mask rows from a fresh lane index: nothing lane-derived lives across the steady loop
            const int l2 = m16_lane();
            qi = l2 & 15, G = l2 >> 4;
            kbase = lds0 + G * 1024 + qi * 16;
            vbase = lds0 + 3 * M16_TILE + G * 2048 + qi * 16;
        }
        // here t == nfull - 1 (last full tile), S(t,0) is complete, P(t-1,1) is ready, ring primed for u = 2t
        fence();
        if (t + 1 < T) {                // a ragged tile t+1 follows: its V is staged now (slot s2)
            dma_v4(t + 1, s2);
        }
        m16_step<1, 1, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s0, 1), k_addr(s2, 0), v_addr(s1, 0), c_log2, M16NoDma());
        if (t + 1 < T) {
            // u = 2t+1 with the masked start for S(t+1,0)
            m16_mask_init<0>(s, last_lim, G, s.mq);
            m16_step<0, 2, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(s2, 0), v_addr(s1, 0), k_addr(s2, 1), v_addr(s1, 1), c_log2, M16NoDma());
            fence();                    // V(t+1) landed
            // u = 2t+2: S(t+1,1) masked | P.V(t,1) | softmax S(t+1,0)
            m16_mask_init<1>(s, last_lim, G, s.mq);
            m16_step<1, 2, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s1, 1), k_addr(s2, 1), v_addr(s2, 0), c_log2, M16NoDma());
            // u = 2t+3: P.V(t+1,0) | softmax S(t+1,1)
            m16_step<0, 0, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s2, 0), k_addr(s2, 1), v_addr(s2, 1), c_log2, M16NoDma());
            // u = 2t+4: P.V(t+1,1)
            m16_step<1, 0, true, false, SCALED, ORD>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s2, 1), k_addr(s2, 1), v_addr(s2, 1), c_log2, M16NoDma());
        } else {
            // u = 2t+1: P.V(t,0) | softmax S(t,1)
            m16_step<0, 0, true, true, SCALED, ORD>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s1, 0), k_addr(s1, 1), v_addr(s1, 1), c_log2, M16NoDma());
            // u = 2t+2: P.V(t,1)
            m16_step<1, 0, true, false, SCALED, ORD>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s1, 1), k_addr(s1, 1), v_addr(s1, 1), c_log2, M16NoDma());
        }
        m16_wait<0>();
        // the last prefetches of the chain are never consumed: keep the ring alive until they have landed
        asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
#pragma unroll
        for (int n = 0; n < 4; ++n) {       // ONE range test of the final row sums (2^-64 .. 2^90 expected; inf / NaN fail too)
            s.l_run[n] += s.psa[n] + s.psb[n];      // (l only grows, inf / NaN are sticky)
            s.psa[n] = s.psb[n] = 0.f;
            float lt = s.l_run[n] + __shfl_xor(s.l_run[n], 16, 64);
            lt += __shfl_xor(lt, 32, 64);
            s.bad |= !(lt >= 8.4703295e-22f && lt <= 1.2379400e27f);       // 2^-70, 2^90
            s.m_run[n] = -s.mq[n][0];       // what the lse below is relative to
        }
        const bool flagged = __syncthreads_or(s.bad) != 0;      // workgroup-uniform: the sweep and the exact loop have barriers
        if (PROF) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[4] += tt - ph_t; ph_t = tt; }
        if (!flagged || (dbg & 1)) break;                       // (debug bit 0: keep the pipelined result even when flagged)
        if (flagcnt && tid == 0) atomicAdd(flagcnt + (attempt ? 1 : 0), 1u);   // debug hook: [0] blocks repeated, [1] blocks sent to the exact loop
        if (attempt) {                      // flagged with the true maxima as reference: inf / NaN scores
            exact_pass = true;
            break;
        }
    }   // attempts
    }
    // ------------------------------------------------------------------------------------------
    // exact pass: plain one-slot loop, running maxima; short rows, and blocks whose scores are not finite
    // ------------------------------------------------------------------------------------------
    if (exact_pass) {
        reset();
        for (int t = 0; t < T; ++t) {
            __syncthreads();
            dma_k4(t, 0), dma_v4(t, 0);
            fence();
            q_pin();
            const int lim = t == T - 1 ? last_lim : 64;
            bare_S(KB0{}, 0, lim);
            bare_S(KB1{}, 0, lim);
            m16_softmax_exact<0>(s, c_log2);            // P of a unit is relative to the maximum at ITS softmax:
            bare_PV(KB0{}, 0);                  // it must reach O^T before the next rescale
            m16_softmax_exact<1>(s, c_log2);
            bare_PV(KB1{}, 0);
        }
    }

    {
        const int l3 = m16_lane();
        qi = l3 & 15, G = l3 >> 4;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        float l_tot = s.l_run[n] + __shfl_xor(s.l_run[n], 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = 1.f / l_tot;
        const int64_t qr = (int64_t)qb0 * M16_QB + wave * 64 + n * 16 + qi;
        if (lse && G == 0 && qr < Lq) lse[(int64_t)head * Lq + qr] = (s.m_run[n] * c_log2 + __log2f(l_tot)) * 0.6931471805599453f;
        if (qr < Lq) {
            uint16_t* op = o + qr * ldo + head * 128 + G * 4;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint2 pk;
                pk.x = pack_bf2(s.ot[d][n][0] * inv, s.ot[d][n][1] * inv);
                pk.y = pack_bf2(s.ot[d][n][2] * inv, s.ot[d][n][3] * inv);
                *(uint2*)(op + d * 16) = pk;
            }
        }
    }
    if (PROF) { ph[7] += __builtin_amdgcn_s_memtime() - ph_t; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long tt = __builtin_amdgcn_s_memtime(); ph[5] += tt - ph_t; ph[6] += 1; }      // ([7]: up to the last store's ISSUE)      // (PROF waits for the stores: what the next item's loads queue behind)
    if (ticketed) {
        if (tid == 0) s_next_item = nwg + (int)next_ticket;
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_next_item);
    } else
        item += item_step;
    }   // work loop
    // the last workgroup to leave re-arms the pair {next ticket, workgroups done} for the launch that uses this slot of the ring next
    if (ticketed && tid == 0 && atomicAdd(tickets + 1, 1u) == (unsigned)nwg - 1) {
        __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (PROF && prof && tid == 0) prof[16 + 2 * bid] = r_start, prof[16 + 2 * bid + 1] = __builtin_amdgcn_s_memrealtime();      // the last launch's {start, end} per workgroup
    if (PROF && prof && (tid & 63) == 0) {
        atomicAdd(prof + wave * 4 + 0, pf_fence);
        atomicAdd(prof + wave * 4 + 1, pf_a);
        atomicAdd(prof + wave * 4 + 2, pf_b);
        atomicAdd(prof + wave * 4 + 3, pf_n);
    }
    if (PROF && prof && tid == 0) {     // [1040 .. 1047): wave 0's per-item phases, summed over the workgroups and launches
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(prof + 1040 + i, ph[i]);
    }
}

// Ticket counters of the persistent launches: ONE pair {next ticket, workgroups done} in the CALLER's workspace (mg_attn_workspace_bytes()
// zero-initialised device bytes, include/moviigen_hip.h): zero when idle — the last workgroup of a launch re-arms it — so launches that are
// ordered with respect to one another (one stream, or event-ordered) share it, and a captured launch that is replayed finds it re-armed by its
// previous run.  The library allocates nothing and synchronises nothing here; with no workspace the launch uses the static per-XCD partition.
static int g_m16_dbg = 0;
static unsigned long long* g_m16_prof = nullptr;
static unsigned* g_m16_flagcnt = nullptr;
void mg_attn_m16_hooks(int dbg, unsigned long long* prof, unsigned* flagcnt) { g_m16_dbg = dbg, g_m16_prof = prof, g_m16_flagcnt = flagcnt; }

// c_log2 = scale*log2(e) of the scores; prescaled != 0: q already carries that factor (mg_rmsnorm_rope_bf16 out_scale).
// kp must be in the m16 row order (mg_pack_kv_bf16 with the m16 kernel selected).
// reserve_cus: CUs this launch leaves free (rounded up to a multiple of the 8 XCDs: workgroups are dealt round-robin over
// the XCDs, so 8 = one CU per XCD).  A persistent workgroup owns its CU's whole register file for the length of the
// launch: a kernel on another stream — RCCL's all-to-all of the next head group (wan/distributed/ulysses.py) — can only
// run on CUs this grid does not occupy.
int mg_attn_m16_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int prescaled, int nqb, float* lse, int reserve_cus,
                       unsigned* workspace, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (96 KiB LDS), a multiple of the 8 XCDs
    n_cu -= (reserve_cus + 7) & ~7;
    if (n_cu < 8) n_cu = 8;
    const int total = nqb * heads;
    const unsigned grid = total <= n_cu ? (unsigned)total : (unsigned)n_cu;   // persistent when there is more work than CUs
    // Items by ticket from 32 rounds on (the metric's launch has 80: +1.35 %, 220.3 against 223.3 ms; 16 rounds: -0.2 %, and at the 10 rounds of
    // a sequence-parallel rank nobody has a spare item to take — profiles/r05w_attn_tickets.log); below that the static per-XCD partition.
    unsigned* tickets = nullptr;
    if (workspace && (int64_t)total >= 32 * (int64_t)grid && !(g_m16_dbg & 16))      // (debug bit 4, A/B library: the static partition always)
        tickets = workspace;
#define M16_ORD 9      // what the library ships (m16_step): one v_exp_f32 per MFMA gap, alone (placement 1, profiles/r05m_attn_order.log: +4.8 % over placement 0), and
                       // the MFMAs of a group in quads that share their A operand (profiles/r06s_attn_mfma_order.log: +0.3 ... +0.5 % over S P S P, same cycles, same bits)
#define M16_LAUNCH(PROF, SCALED, ORD)                                                                                             \
    hipLaunchKernelGGL((attn_hd128_m16_kernel<PROF, SCALED, ORD>), dim3(grid), dim3(M16_THREADS), 0, st, q, ldq, kp, vp, o, ldo, Lq, \
                       Lk, heads, prescaled ? 1.0f : c_log2, nqb, g_m16_dbg, PROF ? g_m16_prof : nullptr, lse, g_m16_flagcnt, tickets)
#ifdef MG_AB_BUILD
    const int ord = (g_m16_dbg >> 1) & 7;     // measurement: mg_attn_w64_debug(2 * k) runs the pre-scaled entry on placement k
    const int mo = (g_m16_dbg >> 5) & 3;      // measurement: debug bits 5-6 = MFMA order of a group (1: pairs, 2: quads sharing their A operand) under placement 1's gaps
    if (mo && prescaled && !g_m16_prof) {
        if (mo == 1) M16_LAUNCH(false, false, 8);
        else if (mo == 2) M16_LAUNCH(false, false, 9);
        else M16_LAUNCH(false, false, 1);     // 3: round 5's order (S P S P) under the same gaps
        return mg_check_launch();
    }
    if (ord && prescaled && !g_m16_prof) {
        if (ord == 1) M16_LAUNCH(false, false, 1);
        else if (ord == 2) M16_LAUNCH(false, false, 2);
        else if (ord == 3) M16_LAUNCH(false, false, 3);
        else if (ord == 4) M16_LAUNCH(false, false, 4);
        else if (ord == 5) M16_LAUNCH(false, false, 5);
        else M16_LAUNCH(false, false, 0);
        return mg_check_launch();
    }
#endif
    if (g_m16_prof) {
        if (prescaled) M16_LAUNCH(true, false, M16_ORD);
        else M16_LAUNCH(true, true, M16_ORD);
    } else {
        if (prescaled) M16_LAUNCH(false, false, M16_ORD);
        else M16_LAUNCH(false, true, M16_ORD);
    }
#undef M16_LAUNCH
    return mg_check_launch();
}
