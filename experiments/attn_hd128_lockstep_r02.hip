// ARCHIVED (round 3): the two-level lock-step attention kernel (8 waves x 32 queries, 32x32x16 MFMA) that served key
// sequences shorter than 512 and was the A/B partner of the w64 kernel in rounds 1-2, together with the round-2
// dispatcher and the K/V pack kernel in the row order of those kernels.  Not built.  The product library now has
// csrc/attn_hd128_m16.hip (16x16x32, all key counts) with csrc/attn_hd128_w64.hip as its A/B partner.
// Flash-style attention forward, head_dim 128, non-causal, bf16 in/out, fp32 accumulate — gfx950.
//
// Replaces flash_attn_varlen_func as the reference calls it (wan/modules/attention.py:96-127)
// from WanSelfAttention (wan/modules/model.py:146-151; L = 75 600 .. 166 320 keys) and
// WanT2VCrossAttention (model.py:176; 512 keys).  72-85 % of all FLOPs of the path.
//
// Operand layout (ABI 2).  Q and O are row-major [L][heads*128].  K and V arrive PRE-PACKED per
// 64-key tile (mg_pack_kv_bf16, one pass per layer, 0.2 % of the attention time):
//     kp[head][tile][c = d/8 (16)][r = key%64][8]      vp[head][tile][kc = (key%64)/8 (8)][d (128)][8]
// i.e. every tile is one contiguous 16 KiB block whose byte image IS the LDS image:
//   * staging is a pure contiguous LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction,
//     fully coalesced) — no staging VGPRs, no ds_write pass, no swizzle arithmetic;
//   * an MFMA fragment is the 16-byte chunk (c, row): lanes of a ds_read_b128 group read
//     different rows of the SAME chunk column = 16 consecutive 16-byte slots: conflict-free by
//     construction, and the address is ONE per-lane base (row*16 [+ g*chunk-stride]) plus an
//     immediate — the XOR-swizzled row-major image of ABI 1 needed a separate address VGPR per
//     (fragment, slot) and ~17 VALU ops per tile to form them.
//
// Math (all schedules):  S^T = K.Q^T with K as the MFMA A-operand and Q^T (registers, whole
// kernel) as B: a lane owns ONE query, softmax statistics are per-lane scalars.  K rows enter the
// MFMA permuted (row bits 2<->3) so a lane's 8 consecutive accumulator registers are 8 consecutive
// keys: P (bf16) is directly the B-operand of O^T = V^T.P^T, the A-operand the chunk
// vp[kc][d] = V[8 keys][d].  Maximum-free softmax against the running reference (att_softmax_fast).
//
// Two kernels share this ABI.  This file: "two-level lock-step", 8 waves x 32 queries, one barrier per
// 64-key tile; the hot loop only knows the maximum-free softmax branch and leaves for an exact tile when
// its check fails.  attn_hd128_w64.hip: 4 waves x 64 queries, one wave per SIMD, software-pipelined.
// mg_attn_set_variant: 0 = auto (w64 for Lk >= 512, else two-level), 1 = two-level with fragment reads
// scheduled by hipcc (4-deep ring), 2 = two-level with a hand-issued ds_read_b128 ring (8 deep,
// counted lgkmcnt), 3 = w64.  What was measured and dropped in
// round 1 (ping-pong role split, intra-wave pipelined softmax, accumulator rotation on/off: all
// within 960-1095 TFLOP/s) is archived in experiments/attn_hd128_schedules_r01.hip; DESIGN.md 3.1
// has the s_memtime breakdown that explains why.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define ATT_THREADS 512
#define ATT_QB 256
#define ATT_KV 64
#define TILE_BYTES 16384
#define K_OFF(slot) ((slot) * TILE_BYTES)
#define V_OFF(slot) (2 * TILE_BYTES + (slot) * TILE_BYTES)

typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;

MG_DEV bf16x8_t att_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
MG_DEV void att_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((att_gptr_t)g, (att_lptr_t)l, 16, 0, 0);
}

struct AttState {
    f32x16_t ot[4];
    f32x16_t st[2];
    bf16x8_t pf[1][2][2];   // [buffer][kb][h] (one buffer: the two-level kernel is not double-buffered)
    float m_run, l_run;
};

// online softmax of one 64-key tile: st -> pf[PB], (m, l) update.
//
// LAZY (default): the tile is exponentiated against the running reference m_run of the earlier tiles
// WITHOUT first reducing its own maximum: p = 2^(s*c - m_run*c).  That is exact as long as no p
// overflows the deferred-rescale bound 2^8, and since p >= 0 a lane's row sum bounds its maximum, so
// `sum(p) <= 2^8` on every lane is a sufficient check that costs nothing extra: no v_max3 chain, no
// cross-lane exchange — 3.5 VALU per score (fma, exp, add, 1/2 cvt_pk) instead of 5.6.  m_run is any
// reference point, not necessarily the true maximum; when the check fails (first tile of a row:
// m_run = -1e30 gives inf; later only when a row's maximum grows by more than ~2^3..2^8) the tile
// is redone by the exact branch below from the same S registers, which also rescales O^T and l.
MG_DEV void att_mask_tail(AttState& s, int lim, int g) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r >> 3) * 16 + g * 8 + (r & 7);
            if (key >= lim) s.st[kb][r] = -1e30f;
        }
}

// fast branch: consumes s.st (the caller recomputes S^T for the exact branch when this returns false)
template <int PB>
MG_DEV bool att_softmax_fast(AttState& s, int lim, int g, float c_log2) {
    if (lim < ATT_KV) att_mask_tail(s, lim, g);
    const float mc = s.m_run * c_log2;
    float ps0 = 0.f, ps1 = 0.f;
    u32x4_t w[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pa = __builtin_amdgcn_exp2f(s.st[kb][2 * j] * c_log2 - mc);
            const float pb = __builtin_amdgcn_exp2f(s.st[kb][2 * j + 1] * c_log2 - mc);
            ps0 += pa;
            ps1 += pb;
            w[kb][j >> 2][j & 3] = pack_bf2(pa, pb);
        }
    const float psum = ps0 + ps1;
    if (!__all(psum <= 256.f)) return false;      // inf / NaN fail too
    s.l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int h = 0; h < 2; ++h) s.pf[PB][kb][h] = att_bf(w[kb][h]);
    return true;
}

// exact branch: true tile maximum, rescale of O^T and l
template <int PB>
MG_DEV void att_softmax_exact(AttState& s, int lim, int g, float c_log2) {
    if (lim < ATT_KV) att_mask_tail(s, lim, g);
    float tmax = s.st[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s.st[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s.st[1][r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(s.m_run, tmax);
    const float mc = m_new * c_log2;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(s.st[kb][r] * c_log2 - mc);
            psum += p[r];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[h * 8 + 2 * e], p[h * 8 + 2 * e + 1]);
            s.pf[PB][kb][h] = att_bf(w);
        }
    }
    const float alpha = __builtin_amdgcn_exp2f((s.m_run - m_new) * c_log2);
    s.l_run = s.l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) s.ot[d][e] *= alpha;
    s.m_run = m_new;
}

// MFMA issue order.  Item i < 16 is a P.V step (d, kb, h), item 16+j an S^T step (kb, kk).  With
// ATT_ROTATE consecutive items never target the same accumulator (P.V cycles through the four
// O^T blocks, S^T alternates the two key halves): an instruction issued between two MFMAs on the
// SAME accumulator breaks the back-to-back accumulate forwarding (+~43 cycles, MI355X_MICROARCH.md).
#ifndef ATT_ROTATE
#define ATT_ROTATE 1
#endif
constexpr int att_pv_d(int i) { return ATT_ROTATE ? (i & 3) : (i >> 2); }
constexpr int att_pv_kb(int i) { return ATT_ROTATE ? (i >> 3) : ((i >> 1) & 1); }
constexpr int att_pv_h(int i) { return ATT_ROTATE ? ((i >> 2) & 1) : (i & 1); }
constexpr int att_st_kb(int j) { return ATT_ROTATE ? (j & 1) : (j >> 3); }
constexpr int att_st_kk(int j) { return ATT_ROTATE ? (j >> 1) : (j & 7); }

// fragment addresses: K (kb, kk) -> kbase + slot + kk*2048 + kb*512 ; V (d, kb, h) -> vbase + slot + kb*8192 + h*4096 + d*512
// item i < 16: V^T fragment of P.V (d = i>>2, kb = (i>>1)&1, h = i&1); item 16+j: K fragment of S^T (kb = j>>3, kk = j&7)
MG_DEV const char* att_frag(const char* smem, int kbase, int vbase, int i, int vslot, int kslot) {
    if (i < 16) return smem + V_OFF(vslot) + vbase + att_pv_kb(i) * 8192 + att_pv_h(i) * 4096 + att_pv_d(i) * 512;
    const int j = i - 16;
    return smem + K_OFF(kslot) + kbase + att_st_kk(j) * 2048 + att_st_kb(j) * 512;
}

// byte offset (compile-time) of fragment item i relative to the per-lane base (vbase for i < 16, kbase otherwise)
constexpr int att_frag_off(int i, int vslot, int kslot) {
    return i < 16 ? V_OFF(vslot) + att_pv_kb(i) * 8192 + att_pv_h(i) * 4096 + att_pv_d(i) * 512
                  : K_OFF(kslot) + att_st_kk(i - 16) * 2048 + att_st_kb(i - 16) * 512;
}

// Matrix work, items [I0, I1), with the fragment reads HAND-ISSUED: hipcc guards a ring of plain
// loads with `s_waitcnt lgkmcnt(0)` every few MFMAs, i.e. it always waits for the read it issued
// last, with zero cover.  Here ds_read_b128 is inline asm (invisible to hipcc's waitcnt pass), DEPTH
// reads are kept in flight and MFMA i is preceded by a COUNTED lgkmcnt(min(DEPTH-1, I1-1-i)):
// LDS returns in order, so that is exactly "fragment i has landed".  sched_barrier(0) pins the
// builtin MFMAs between the asm statements (cdna guide §5.4 rule 18).  lds_v / lds_k are the
// per-lane LDS byte addresses (workgroup LDS base + vbase / kbase); VS/KS the tile slots.
// Compile-time recursion: the asm "i" operands must be constants before unrolling.
template <int I, int VS, int KS>
MG_DEV void att_rd(bf16x8_t& dst, unsigned lds_v, unsigned lds_k) {
    if constexpr (I < 16)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_v), "i"(att_frag_off(I, VS, KS)));
    else
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_k), "i"(att_frag_off(I, VS, KS)));
}
template <int I, int I0, int I1, int DEPTH, int VS, int KS>
MG_DEV void att_matrix_step(AttState& s, const bf16x8_t (&qf)[8], bf16x8_t (&f)[DEPTH], unsigned lds_v, unsigned lds_k) {
    if constexpr (I < I1) {
        constexpr int r = (I - I0) % DEPTH;
        constexpr int pending = (I1 - 1 - I) < (DEPTH - 1) ? (I1 - 1 - I) : (DEPTH - 1);  // reads younger than item I
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(pending) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I < 16) {
            s.ot[att_pv_d(I)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r], s.pf[0][att_pv_kb(I)][att_pv_h(I)], s.ot[att_pv_d(I)], 0, 0, 0);
        } else {
            constexpr int j = I - 16;
            if constexpr (att_st_kk(j) == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s.st[att_st_kb(j)][e] = 0.f;
            }
            s.st[att_st_kb(j)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r], qf[att_st_kk(j)], s.st[att_st_kb(j)], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I + DEPTH < I1) att_rd<I + DEPTH, VS, KS>(f[r], lds_v, lds_k);
        att_matrix_step<I + 1, I0, I1, DEPTH, VS, KS>(s, qf, f, lds_v, lds_k);
    }
}
template <int I, int I0, int I1, int DEPTH, int VS, int KS>
MG_DEV void att_matrix_fill(bf16x8_t (&f)[DEPTH], unsigned lds_v, unsigned lds_k) {
    if constexpr (I < I0 + DEPTH && I < I1) {
        att_rd<I, VS, KS>(f[I - I0], lds_v, lds_k);
        att_matrix_fill<I + 1, I0, I1, DEPTH, VS, KS>(f, lds_v, lds_k);
    }
}
template <int I0, int I1, int DEPTH, int VS, int KS>
MG_DEV void att_matrix_asm(AttState& s, const bf16x8_t (&qf)[8], unsigned lds_v, unsigned lds_k) {
    bf16x8_t f[DEPTH];
    att_matrix_fill<I0, I0, I1, DEPTH, VS, KS>(f, lds_v, lds_k);
    att_matrix_step<I0, I0, I1, DEPTH, VS, KS>(s, qf, f, lds_v, lds_k);
}

// matrix work with an explicit DEPTH-deep fragment ring; items [I0, I1) of the list above.
// `between(n)` is called after the n-th MFMA of the segment (n = 0, 1, ...): the hook for work that
// must be spread between the MFMAs instead of issued in a block (LDS-DMA pieces: one
// global_load_lds costs the issuing wave ~60-150 cycles, during which its partner keeps the pipe busy).
struct AttNoHook {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int I0, int I1, int DEPTH, int PB = 0, typename Hook = AttNoHook>
MG_DEV void att_matrix(AttState& s, const bf16x8_t (&qf)[8], const char* smem, int kbase, int vbase, int vslot,
                       int kslot, Hook between = Hook()) {
    bf16x8_t f[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (I0 + i < I1) f[i] = *(const bf16x8_t*)att_frag(smem, kbase, vbase, I0 + i, vslot, kslot);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const int r = (i - I0) % DEPTH;
        if (i < 16) {
            s.ot[att_pv_d(i)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r], s.pf[PB][att_pv_kb(i)][att_pv_h(i)], s.ot[att_pv_d(i)], 0, 0, 0);
        } else {
            const int j = i - 16;
            if (att_st_kk(j) == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s.st[att_st_kb(j)][e] = 0.f;
            }
            s.st[att_st_kb(j)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[r], qf[att_st_kk(j)], s.st[att_st_kb(j)], 0, 0, 0);
        }
        if (i + DEPTH < I1) f[r] = *(const bf16x8_t*)att_frag(smem, kbase, vbase, i + DEPTH, vslot, kslot);
        between(i - I0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// -------------------------------------------------------------------------------------------------
// The kernel, "two-level": lock-step tiles, but the exact softmax (true maximum, rescale of O^T) is
// moved OUT of the hot loop.  A rescale inside the tile loop makes O^T (64 VGPRs) a phi of two
// definitions and hipcc pays 32 v_mov_b64 per tile on the hot edge; here the inner loop only knows
// the fast branch (att_softmax_fast: no maximum, no rescale) and LEAVES when its check fails; the
// outer loop redoes that tile exactly and re-enters.  LDS slots are addressed at run time (t & 1
// folded into the per-lane base), so there is no parity unrolling.
// -------------------------------------------------------------------------------------------------
template <bool LAZY, int DEPTH, bool PROF = false, bool ASM = false>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_hd128_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp,
    uint16_t* __restrict__ o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb,
    unsigned long long* __restrict__ prof, float* __restrict__ lse) {
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pc = 0;   // PROF: s_memtime sums of S^T / softmax / P.V / fence / fast tiles
    auto tick = [&](int i) __attribute__((always_inline)) {
        if (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (i >= 0) pt[i] += now - pc;
            pc = now;
        }
    };
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int bid = blockIdx.x;
    const int head = bid / nqb;
    const int qb = bid - head * nqb;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;

    const int64_t qrow_raw = (int64_t)qb * ATT_QB + wave * 32 + l31;
    const int64_t qrow = qrow_raw < Lq ? qrow_raw : Lq - 1;
    bf16x8_t qf[8];
    {
        const uint16_t* qp = q + qrow * ldq + head * 128 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = att_bf(*(const u32x4_t*)(qp + kk * 16));
    }
    const int nkv = (int)((Lk + ATT_KV - 1) / ATT_KV);
    const uint16_t* k_src = kp + ((int64_t)head * nkv) * 8192 + wave * 1024 + lane * 8;
    const uint16_t* v_src = vp + ((int64_t)head * nkv) * 8192 + wave * 1024 + lane * 8;
    auto dma_piece = [&](int t, int n) __attribute__((always_inline)) {   // piece n of 4 of tile t
        if (t < nkv) {
            const uint16_t* src = (n < 2 ? k_src : v_src) + (int64_t)t * 8192 + (n & 1) * 512;
            char* dst = smem + (n < 2 ? K_OFF(t & 1) : V_OFF(t & 1)) + wave * 2048 + (n & 1) * 1024;
            att_glds16(src, dst);
        }
    };
    auto dma = [&](int t) __attribute__((always_inline)) {     // K(t), V(t) -> slot t & 1
        if (t < nkv) {
            char* dk = smem + K_OFF(t & 1) + wave * 2048;
            char* dv = smem + V_OFF(t & 1) + wave * 2048;
            att_glds16(k_src + (int64_t)t * 8192, dk);
            att_glds16(k_src + (int64_t)t * 8192 + 512, dk + 1024);
            att_glds16(v_src + (int64_t)t * 8192, dv);
            att_glds16(v_src + (int64_t)t * 8192 + 512, dv + 1024);
        }
    };
    const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int kbase = g * 1024 + kperm * 16;
    const int vbase = g * 2048 + l31 * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(att_lptr_t)smem;
    const unsigned lds_k = lds0 + kbase, lds_v = lds0 + vbase;

    AttState s;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) s.ot[d][e] = 0.f;
    s.m_run = -1e30f;
    s.l_run = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));

    auto fence = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // S^T(t) / P.V(t) segments; the tile slot (t & 1) is folded into the per-lane base address
    auto seg_st = [&](int t) __attribute__((always_inline)) {
        if constexpr (ASM) att_matrix_asm<16, 32, DEPTH, 0, 0>(s, qf, lds_v + (t & 1) * TILE_BYTES, lds_k + (t & 1) * TILE_BYTES);
        else att_matrix<16, 32, DEPTH>(s, qf, smem, kbase, vbase, t & 1, t & 1);
    };
    auto seg_pv = [&](int t) __attribute__((always_inline)) {
        if constexpr (ASM) att_matrix_asm<0, 16, DEPTH, 0, 0>(s, qf, lds_v + (t & 1) * TILE_BYTES, lds_k + (t & 1) * TILE_BYTES);
        else att_matrix<0, 16, DEPTH>(s, qf, smem, kbase, vbase, t & 1, t & 1);
    };
    dma(0);
    fence();
    const int last_lim = (int)(Lk - (int64_t)(nkv - 1) * ATT_KV);      // keys in the last tile, 1..64
    const int nfast = (LAZY && last_lim == ATT_KV) ? nkv : (LAZY ? nkv - 1 : 0);   // the fast loop only sees full tiles
    int t = 0;
    bool redo = false;
    while (t < nkv) {
        // exact tile: first tile of the row, a ragged last tile, or the tile the fast loop gave up on
        if (!redo) dma(t + 1);
        seg_st(t);
        att_softmax_exact<0>(s, t == nkv - 1 ? last_lim : ATT_KV, g, c_log2);
        seg_pv(t);
        fence();
        ++t;
        redo = false;
        while (t < nfast) {
            tick(-1);
            if constexpr (ASM) {
                seg_st(t);
                dma(t + 1);
            } else {   // S^T(t) with the 4 refill pieces of tile t+1 behind MFMAs 1, 3, 5, 7
                att_matrix<16, 32, DEPTH>(s, qf, smem, kbase, vbase, t & 1, t & 1, [&](int n) __attribute__((always_inline)) {
                    if (n < 8 && (n & 1)) dma_piece(t + 1, n >> 1);
                });
            }
            tick(0);
            if (__builtin_expect(!att_softmax_fast<0>(s, ATT_KV, g, c_log2), 0)) {
                redo = true;
                break;
            }
            tick(1);
            seg_pv(t);                                               // O^T += V^T(t).P^T
            tick(2);
            fence();
            tick(3);
            if (PROF) pt[4] += 1;
            ++t;
        }
    }

    if (PROF && lane == 0 && prof) {
#pragma unroll
        for (int i = 0; i < 5; ++i) atomicAdd(prof + wave * 5 + i, pt[i]);
    }
    const float l_tot = s.l_run + __shfl_xor(s.l_run, 32, 64);
    const float inv = 1.f / l_tot;
    // optional log-sum-exp of the scaled scores (ring attention merges partial results with it)
    if (lse && g == 0 && qrow_raw < Lq) lse[(int64_t)head * Lq + qrow_raw] = (s.m_run * c_log2 + __log2f(l_tot)) * 0.6931471805599453f;
    if (qrow_raw < Lq) {
        uint16_t* op = o + qrow_raw * ldo + head * 128 + g * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2 pk;
                pk.x = pack_bf2(s.ot[d][rq * 4 + 0] * inv, s.ot[d][rq * 4 + 1] * inv);
                pk.y = pack_bf2(s.ot[d][rq * 4 + 2] * inv, s.ot[d][rq * 4 + 3] * inv);
                *(uint2*)(op + d * 32 + rq * 8) = pk;
            }
    }
}

// -------------------------------------------------------------------------------------------------
// K / V -> packed tiles (see the header of this file)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kv_kernel(const uint16_t* __restrict__ k, int64_t ldk,
                                                      const uint16_t* __restrict__ v, int64_t ldv, int64_t L,
                                                      uint16_t* __restrict__ kp, uint16_t* __restrict__ vp, int nt) {
    __shared__ uint16_t tile[64][128 + 8];
    const int head = blockIdx.y, t = blockIdx.x;
    const int64_t k0 = (int64_t)t * 64;
    const int64_t tbase = ((int64_t)head * nt + t) * 8192;
    if (k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + i * 256;      // (r, c): coalesced 256-B row reads
            const int r = id >> 4, c = id & 15;
            u16x8_t u = {0, 0, 0, 0, 0, 0, 0, 0};
            if (k0 + r < L) u = *(const u16x8_t*)(k + (k0 + r) * ldk + head * 128 + c * 8);
            *(u16x8_t*)(kp + tbase + c * 512 + r * 8) = u;
        }
    }
    if (v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + i * 256;
            const int r = id >> 4, c = id & 15;
            u16x8_t u = {0, 0, 0, 0, 0, 0, 0, 0};
            if (k0 + r < L) u = *(const u16x8_t*)(v + (k0 + r) * ldv + head * 128 + c * 8);
            *(u16x8_t*)&tile[r][c * 8] = u;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = threadIdx.x + i * 256;      // (kc, d): 2-KiB contiguous runs per kc
            const int kc = id >> 7, d = id & 127;
            u16x8_t u;
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = tile[kc * 8 + j][d];
            *(u16x8_t*)(vp + tbase + kc * 1024 + d * 8) = u;
        }
    }
}

extern "C" int mg_pack_kv_bf16(const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, int64_t L, int heads,
                               int head_dim, uint16_t* kp, uint16_t* vp, void* stream) {
    if ((!k && !v) || (k && !kp) || (v && !vp)) return MG_ERR_ARG;
    if (head_dim != 128 || heads <= 0 || L <= 0 || (k && (ldk & 7)) || (v && (ldv & 7))) return MG_ERR_SHAPE;
    const int nt = (int)((L + 63) / 64);
    hipLaunchKernelGGL(pack_kv_kernel, dim3(nt, heads), dim3(256), 0, (hipStream_t)stream, k, ldk, v, ldv, L, kp, vp,
                       nt);
    return mg_check_launch();
}

static unsigned long long* g_attn_prof = nullptr;
// debug hook (not in the public header): device buffer of 8 waves x 5 counters for schedule 5's PROF build
extern "C" void mg_attn_debug_profile(unsigned long long* dev_buf) { g_attn_prof = dev_buf; }
int mg_attn_w64_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, float* lse, hipStream_t st);
static int g_attn_lazy = 1;
static int g_attn_variant = 0;
extern "C" void mg_attn_set_lazy_rescale(int on) { g_attn_lazy = on; }
extern "C" void mg_attn_set_variant(int v) { g_attn_variant = v; }

extern "C" int mg_attn_fwd_bf16_hd128_lse(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                          uint16_t* o, int64_t ldo, float* lse, int64_t Lq, int64_t Lk, int heads,
                                          float scale, void* stream) {
    if (!q || !kp || !vp || !o) return MG_ERR_ARG;
    if (Lq < 0 || Lk <= 0 || heads <= 0) return MG_ERR_SHAPE;
    if ((ldq & 7) || (ldo & 3)) return MG_ERR_SHAPE;
    if (((uintptr_t)q & 15) || ((uintptr_t)kp & 15) || ((uintptr_t)vp & 15) || ((uintptr_t)o & 7)) return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    const int64_t nqb64 = (Lq + ATT_QB - 1) / ATT_QB;
    if (nqb64 * heads > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int nqb = (int)nqb64;
    const float c_log2 = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)(nqb * heads)), block(ATT_THREADS);
    hipStream_t st = (hipStream_t)stream;
    // 0 = auto: the one-wave-per-SIMD "w64" kernel from 8 key tiles up (self-attention, and the 512-key
    // cross-attention: 660 vs 600 TFLOP/s), the two-level lock-step kernel for shorter key sequences
    const int variant = g_attn_variant == 0 ? (Lk >= 512 ? 3 : 1) : g_attn_variant;
    if (variant == 3) return mg_attn_w64_launch(q, ldq, kp, vp, o, ldo, Lq, Lk, heads, c_log2, nqb, lse, st);
#define ATT_LAUNCH(LZ, DP, PROF, ASM) \
    hipLaunchKernelGGL((attn_hd128_kernel<LZ, DP, PROF, ASM>), grid, block, 0, st, q, ldq, kp, vp, o, ldo, Lq, Lk, heads, c_log2, nqb, g_attn_prof, lse)
    if (variant == 2) {
        if (g_attn_prof) ATT_LAUNCH(true, 8, true, true);
        else if (g_attn_lazy) ATT_LAUNCH(true, 8, false, true);
        else ATT_LAUNCH(false, 8, false, true);
    } else {
        if (g_attn_prof) ATT_LAUNCH(true, 4, true, false);
        else if (g_attn_lazy) ATT_LAUNCH(true, 4, false, false);
        else ATT_LAUNCH(false, 4, false, false);
    }
#undef ATT_LAUNCH
    return mg_check_launch();
}

extern "C" int mg_attn_fwd_bf16_hd128(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                      uint16_t* o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float scale,
                                      void* stream) {
    return mg_attn_fwd_bf16_hd128_lse(q, ldq, kp, vp, o, ldo, nullptr, Lq, Lk, heads, scale, stream);
}
