// A/B PARTNER of attn_hd128_m16.hip since round 3 (mg_attn_set_variant(3)): the round-2 kernel, unchanged — it needs
// the K tiles in natural row order (mg_pack_kv_bf16 follows the selection).
// Flash-style attention forward, head_dim 128 — schedule "w64": 4 waves x 64 queries, ONE wave per
// SIMD, software-pipelined in half-tile units.  Same operands and math as attn_hd128.hip (packed
// K/V tiles, S^T = K.Q^T, maximum-free softmax against the running reference, O^T accumulators).
//
// Why (experiments/mfma_probe.hip): on gfx950 two waves that share a SIMD serialize
// their MFMA and VALU work (time ~ sum), while a wave ALONE on its SIMD hides up to ~5 single-issue
// instructions behind each of its own 32-cycle MFMAs.  So here every wave owns a SIMD and 64
// queries (a K/V fragment read from LDS feeds two MFMAs: 0.5 ds_read_b128 per MFMA instead of 1),
// and its instruction stream is built so that every MFMA is followed by <= 5 fillers:
//
//   unit u = (tile t, key half kb), 32 keys.  Step u issues, interleaved,
//       S_{u+1}  = K_{u+1}.Q^T            16 MFMAs  (8 K fragments x 2 query blocks)
//       O^T     += V^T_{u-1}.P^T_{u-1}    16 MFMAs  (8 V fragments x 2 query blocks)
//       P_u      = 2^(S_u*c - m*c), l    ~112 VALU  (the three streams are independent)
//     + 16 ds_read_b128 (ring, 2 groups ahead, counted lgkmcnt) + 4 LDS-DMA pieces of a later tile.
//   S and P are double-buffered by kb (compile-time), tiles live in 3 K + 3 V LDS slots (96 KiB),
//   one barrier per tile.  There is no branch in the hot loop and the softmax reference of a row is
//   FIXED at the maximum of its first 32 keys: P is bf16 and O^T / l are fp32, so nothing is lost
//   while 2^((s - m) c) stays inside the fp32 exponent range (floating point is scale-free; rows
//   whose later scores dwarf the reference simply forget the early keys, as the exact softmax
//   does).  A partial row sum above 2^100 (or inf / NaN) only sets a flag, and a flagged workgroup
//   recomputes its block with the plain exact loop (true running maximum) at the end of the kernel.
#include <type_traits>
#include "common.h"
#include "../../include/moviigen_hip.h"

#define W64_THREADS 256
#define W64_QB 256
#define W64_TILE 16384
#define W64_K(slot) ((slot) * W64_TILE)
#define W64_V(slot) (3 * W64_TILE + (slot) * W64_TILE)

typedef const __attribute__((address_space(1))) void* w64_gptr_t;
typedef __attribute__((address_space(3))) void* w64_lptr_t;
MG_DEV bf16x8_t w64_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
MG_DEV void w64_glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((w64_gptr_t)g, (w64_lptr_t)l, 16, 0, 0); }

struct W64State {
    f32x16_t ot[4][2];     // O^T [d block][query block]
    f32x16_t st[2][2];     // S^T [kb][query block]
    bf16x8_t pf[2][2][2];  // P   [kb][query block][h]
    float m_run[2], l_run[2];
    int bad;
};

// hand-issued fragment reads: base VGPR + immediate offset; returns are in order, waits are counted
template <int OFF>
MG_DEV void w64_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void w64_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// S^T accumulators must live in ARCH VGPRs (the softmax reads them with VALU instructions); hipcc
// selects the AGPR form for every builtin MFMA of a 512-register kernel and then pays one
// v_accvgpr_read per score.  These two go through inline asm with "v" operands instead.  hipcc does
// not see their latency: the schedule itself keeps >= 3 MFMAs (96+ cycles) between the last write of
// an S^T block and its first VALU read (w64_step), which covers the 8-pass write-back hazard.
MG_DEV void w64_mfma_s0(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {       // acc = a.b
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
MG_DEV void w64_mfma_s(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {        // acc += a.b
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// same, first MFMA on a VALU-written start value (ragged tile mask): the VALU -> MFMA srcC wait states
// hipcc would insert for a builtin are part of the asm (cold path)
MG_DEV void w64_mfma_s_after_valu(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
constexpr int w64_koff(int i) { return i * 2048; }                          // K fragment kk = i of a unit
constexpr int w64_voff(int i) { return (i >> 2) * 4096 + (i & 3) * 512; }   // V fragment (h = i>>2, d = i&3)

// exact softmax of unit KB (true maximum of the 32 keys, rescale of O^T and l), both query blocks
template <int KB>
MG_DEV void w64_softmax_exact(W64State& s, float c_log2) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float tmax = s.st[KB][qb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s.st[KB][qb][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(s.m_run[qb], tmax);
        const float mc = m_new * c_log2;
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(s.st[KB][qb][r] * c_log2 - mc);
            psum += p[r];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[h * 8 + 2 * e], p[h * 8 + 2 * e + 1]);
            s.pf[KB][qb][h] = w64_bf(w);
        }
        const float alpha = __builtin_amdgcn_exp2f((s.m_run[qb] - m_new) * c_log2);
        s.l_run[qb] = s.l_run[qb] * alpha + psum;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) s.ot[d][qb][e] *= alpha;
        s.m_run[qb] = m_new;
    }
}

// accumulator start of S^T for a ragged tile: -1e30 on the key rows >= lim (the MFMAs add K.Q^T to it)
template <int KB>
MG_DEV void w64_mask_init(W64State& s, int lim, int g) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = KB * 32 + (r >> 3) * 16 + g * 8 + (r & 7);
            s.st[KB][qb][r] = key >= lim ? -1e30f : 0.f;
        }
}

// One pipeline step.  MFMAs: S^T of unit KB (K fragments at lds_k) when SMODE != 0 (2: masked start,
// s.st[KB] pre-set by w64_mask_init) and P.V of unit KB (V fragments at lds_v, P = s.pf[KB]) when PV.
// VALU: softmax of unit 1-KB (fast branch) when SM.  `dma(i)` is called once per group (i = 0..7).
// Fragment ring: 4 K + 4 V registers (8 groups per step: the phase is the same in every step), reads two groups (8 MFMAs) ahead; the reads for the first two
// groups must have been issued by the caller (prefetch()) — the last two groups of this step issue
// them for the NEXT step from nk / nv (pass 0 to skip).
template <int KB, int SMODE, bool PV, bool SM, typename Dma>
MG_DEV void w64_step(W64State& s, const bf16x8_t (&qf)[2][8], bf16x8_t (&kf)[4], bf16x8_t (&vf)[4], unsigned lds_k,
                     unsigned lds_v, unsigned nk, unsigned nv, float c_log2, Dma dma) {
    constexpr int SB = 1 - KB;          // unit being exponentiated
#ifndef W64_PK_ADD
#define W64_PK_ADD 0
#endif
    float mc[2], psa[2] = {0.f, 0.f}, psb[2] = {0.f, 0.f};
    f32x2_t ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    u32x4_t w[2][2];
    if (SM) {
        mc[0] = s.m_run[0] * c_log2;
        mc[1] = s.m_run[1] * c_log2;
    }
    // softmax of 2 scores (2j, 2j+1 of query block i>>2), in two parts so that every MFMA gap gets 3-4 VALU
    float pa = 0.f, pb = 0.f;
    auto pair_a = [&](int i, int half) __attribute__((always_inline)) {   // fma, fma, exp, exp
        if (SM) {
            const int qb = i >> 2, j = (i & 3) * 2 + half;
            pa = __builtin_amdgcn_exp2f(s.st[SB][qb][2 * j] * c_log2 - mc[qb]);
            pb = __builtin_amdgcn_exp2f(s.st[SB][qb][2 * j + 1] * c_log2 - mc[qb]);
            asm volatile("" : "+v"(pa), "+v"(pb));     // opaque use: pins the work HERE (LLVM sinks it otherwise)
        }
    };
    auto pair_b = [&](int i, int half) __attribute__((always_inline)) {   // add, add, cvt_pk
        if (SM) {
            const int qb = i >> 2, j = (i & 3) * 2 + half;
#if W64_PK_ADD
            // both partial row sums in ONE packed add — measured 1117 instead of 1233 TFLOP/s: packed f32 VALU is an
            // anti-lever beside MFMAs on gfx950 (MI355X_MICROARCH.md); kept as an experiment switch, default off
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ps2[qb]) : "v"(__builtin_shufflevector((f32x2_t){pa, pa}, (f32x2_t){pb, pb}, 0, 2)));
            unsigned pk = pack_bf2(pa, pb);
            asm volatile("" : "+v"(pk));
#else
            psa[qb] += pa;
            psb[qb] += pb;
            unsigned pk = pack_bf2(pa, pb);
            asm volatile("" : "+v"(pk), "+v"(psa[qb]), "+v"(psb[qb]));
#endif
            w[qb][j >> 2][j & 3] = pk;
        }
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i & 3, r2 = (i + 2) & 3;
        // Group i: K fragment i and V fragment i, four MFMAs S0 | PV0 | S1 | PV1 with the fillers between.
        // The asm (S^T) and builtin (P.V) MFMAs alternate on purpose: hipcc cannot see into the asm, so
        // a v_exp result consumed right behind an asm MFMA costs an s_nop; pair_a (ends with the exps)
        // always sits in front of a builtin MFMA, pair_b (adds, cvt) in front of an asm one.
        w64_wait<2>();                         // K(i) and V(i) landed; younger: K(i+1), V(i+1)
        __builtin_amdgcn_sched_barrier(0);
        if (SMODE == 1 && i == 0) w64_mfma_s0(s.st[KB][0], kf[r], qf[0][i]);
        else if (SMODE == 2 && i == 0) w64_mfma_s_after_valu(s.st[KB][0], kf[r], qf[0][i]);
        else if (SMODE != 0) w64_mfma_s(s.st[KB][0], kf[r], qf[0][i]);
        __builtin_amdgcn_sched_barrier(0);
        pair_a(i, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (PV) s.ot[i & 3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[r], s.pf[KB][0][i >> 2], s.ot[i & 3][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        pair_b(i, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (SMODE == 1 && i == 0) w64_mfma_s0(s.st[KB][1], kf[r], qf[1][i]);
        else if (SMODE == 2 && i == 0) w64_mfma_s_after_valu(s.st[KB][1], kf[r], qf[1][i]);
        else if (SMODE != 0) w64_mfma_s(s.st[KB][1], kf[r], qf[1][i]);
        __builtin_amdgcn_sched_barrier(0);
        // A ds_read whose result nobody uses would leave its destination VGPR free for the compiler
        // to reuse while the data is still on its way (it lands ~100 cycles later and clobbers the new
        // owner).  Steps without S^T MFMAs still issue the K reads (the counted waits stay the same),
        // so the previous occupant of the ring slot is kept alive up to here, 4 groups after its read.
        if (SMODE == 0) asm volatile("" ::"v"(kf[r2]));
        if (i < 6) {
            switch (i) {   // compile-time after unrolling
                case 0: w64_rd<w64_koff(2)>(kf[r2], lds_k); break;
                case 1: w64_rd<w64_koff(3)>(kf[r2], lds_k); break;
                case 2: w64_rd<w64_koff(4)>(kf[r2], lds_k); break;
                case 3: w64_rd<w64_koff(5)>(kf[r2], lds_k); break;
                case 4: w64_rd<w64_koff(6)>(kf[r2], lds_k); break;
                default: w64_rd<w64_koff(7)>(kf[r2], lds_k); break;
            }
        } else if (i == 6) w64_rd<w64_koff(0)>(kf[r2], nk);
        else w64_rd<w64_koff(1)>(kf[r2], nk);
        pair_a(i, 1);
        dma(i);
        __builtin_amdgcn_sched_barrier(0);
        if (PV) s.ot[i & 3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[r], s.pf[KB][1][i >> 2], s.ot[i & 3][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i < 6) {
            switch (i) {
                case 0: w64_rd<w64_voff(2)>(vf[r2], lds_v); break;
                case 1: w64_rd<w64_voff(3)>(vf[r2], lds_v); break;
                case 2: w64_rd<w64_voff(4)>(vf[r2], lds_v); break;
                case 3: w64_rd<w64_voff(5)>(vf[r2], lds_v); break;
                case 4: w64_rd<w64_voff(6)>(vf[r2], lds_v); break;
                default: w64_rd<w64_voff(7)>(vf[r2], lds_v); break;
            }
        } else if (i == 6) w64_rd<w64_voff(0)>(vf[r2], nv);
        else w64_rd<w64_voff(1)>(vf[r2], nv);
        pair_b(i, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SM) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float psum = W64_PK_ADD ? ps2[qb][0] + ps2[qb][1] : psa[qb] + psb[qb];
            s.bad |= !(psum <= 1.2676506e30f);  // 2^100; inf / NaN too
            s.l_run[qb] += psum;
            s.pf[SB][qb][0] = w64_bf(w[qb][0]);
            s.pf[SB][qb][1] = w64_bf(w[qb][1]);
        }
    }
}

struct W64NoDma {
    __device__ __forceinline__ void operator()(int) const {}
};

template <bool PROF>
__global__ __launch_bounds__(W64_THREADS, 1) void attn_hd128_w64_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp,
    uint16_t* __restrict__ o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, int dbg, unsigned long long* __restrict__ prof, float* __restrict__ lse,
    unsigned* __restrict__ flagcnt) {
    __shared__ __attribute__((aligned(16))) char smem[6 * W64_TILE];
    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    // Persistent, XCD-aware work loop.  Work items = (head, query block), head-major.  Workgroups are
    // dealt to the 8 XCDs round-robin (bid & 7); XCD x owns the contiguous item range
    // [x*total/8, (x+1)*total/8) and its workgroups walk it in lock-step rounds — at any moment the
    // ~32 workgroups behind one L2 stream the SAME head's K/V at (nearly) the same tile, so a tile
    // is fetched from the fabric once per XCD round instead of once per workgroup (the plain
    // one-item-per-workgroup launch lets finished workgroups restart at tile 0 while their
    // neighbours are mid-sequence: measured 216 GB of fabric reads per 720p launch, 53 % L2 hits).
    const int total_items = nqb * heads;
    const int nwg = gridDim.x;
    int item, item_end, item_step;
    if (nwg == total_items) {
        item = bid, item_end = bid + 1, item_step = 1;
    } else {
        const int xcd = bid & 7, slot = bid >> 3;
        item = (int)((int64_t)xcd * total_items / 8) + slot;
        item_end = (int)((int64_t)(xcd + 1) * total_items / 8);
        item_step = nwg >> 3;           // host guarantees nwg % 8 == 0 here
    }
    for (; item < item_end; item += item_step) {
    const int head = item / nqb;
    const int qb0 = item - head * nqb;
    __syncthreads();                    // the previous item's last LDS reads are done before this one's DMA

    // Q fragments: query block b of this wave = rows 64*wave + 32*b + l31
    bf16x8_t qf[2][8];
    const int64_t qrow_base = (int64_t)qb0 * W64_QB + wave * 64 + l31;     // row of query block 0; block 1 is 32 further
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t qr = qrow_base + b * 32;
        const int64_t qrow = qr < Lq ? qr : Lq - 1;
        const uint16_t* qp = q + qrow * ldq + head * 128 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[b][kk] = w64_bf(*(const u32x4_t*)(qp + kk * 16));
    }
    const int T = (int)((Lk + 63) / 64);
    const int last_lim = (int)(Lk - (int64_t)(T - 1) * 64);     // keys in the last tile, 1..64
    // LDS-DMA: a tile is 16 pieces of 1 KiB; wave w moves pieces 4w..4w+3 of the K tile and of the V tile
    const char* k_src = (const char*)(kp + ((int64_t)head * T) * 8192 + wave * 2048);   // wave-uniform (SGPRs)
    const char* v_src = (const char*)(vp + ((int64_t)head * T) * 8192 + wave * 2048);
    const unsigned lane_off = lane * 16;                                                  // the only per-lane part
    // Key tiles are walked in a ROTATED order (softmax over keys is order-free): logical tile i of the
    // pipelined pass is physical tile (i + rot) mod nfull, a ragged last tile stays last.  In the
    // persistent launch the 32 workgroups behind one L2 start 1/32 of the sequence apart: nobody
    // requests the same lines at the same moment (in-phase walking measured 870 instead of 1200
    // TFLOP/s: the L2 channels of the one hot tile serialize 32 CUs), yet every tile a workgroup
    // wants was fetched ~T/32 tiles earlier by the workgroup ahead of it and is still in the 4 MB L2.
    const int nfull = last_lim == 64 ? T : T - 1;
#ifndef W64_ROTATE
#define W64_ROTATE 0
#endif
    int rot = (!W64_ROTATE || nwg == total_items || nfull < 3) ? 0 : (int)((int64_t)(bid >> 3) * nfull / (nwg >> 3));
    auto phys = [&](int t) __attribute__((always_inline)) {
        if (t >= nfull) return t;
        const int u = t + rot;
        return u < nfull ? u : u - nfull;
    };
    // tile indices past the end are clamped (a redundant reload of the last tile into a free slot)
    // instead of guarded: no branch per piece in the hot loop.  (The SADDR form of the LDS-DMA — SGPR
    // base + 32-bit lane offset + immediate, hand-written — was measured: 139 cycles per piece instead
    // of 55 with hipcc's 64-bit per-lane address form.)
    const unsigned lds0 = (unsigned)(uintptr_t)(w64_lptr_t)smem;
    auto dma_k = [&](int t, int slot, int n) __attribute__((always_inline)) {
        const int tt = t < T ? t : T - 1;
        w64_glds16(k_src + ((int64_t)phys(tt) * 16384 + n * 1024) + lane_off, smem + W64_K(slot) + wave * 4096 + n * 1024);
    };
    auto dma_v = [&](int t, int slot, int n) __attribute__((always_inline)) {
        const int tt = t < T ? t : T - 1;
        w64_glds16(v_src + ((int64_t)phys(tt) * 16384 + n * 1024) + lane_off, smem + W64_V(slot) + wave * 4096 + n * 1024);
    };
    const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const unsigned kbase = lds0 + g * 1024 + kperm * 16;               // + W64_K(slot) + kb*512 + kk*2048
    const unsigned vbase = lds0 + 3 * W64_TILE + g * 2048 + l31 * 16;  // + slot*TILE + kb*8192 + h*4096 + d*512
    auto k_addr = [&](int slot, int kb) __attribute__((always_inline)) { return kbase + slot * W64_TILE + kb * 512; };
    auto v_addr = [&](int slot, int kb) __attribute__((always_inline)) { return vbase + slot * W64_TILE + kb * 8192; };

    W64State s;
    bf16x8_t kf[4], vf[4];
    auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) s.ot[d][b][e] = 0.f;
        s.m_run[0] = s.m_run[1] = -1e30f;
        s.l_run[0] = s.l_run[1] = 0.f;
        s.bad = 0;
    };
    reset();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[b][kk]));
    auto fence = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // prefetch of the first two K and V fragments of a step (what w64_step expects to be in flight)
    auto prefetch = [&](unsigned ak, unsigned av) __attribute__((always_inline)) {
        w64_rd<w64_koff(0)>(kf[0], ak);
        w64_rd<w64_voff(0)>(vf[0], av);
        w64_rd<w64_koff(1)>(kf[1], ak);
        w64_rd<w64_voff(1)>(vf[1], av);
    };
    // bare S^T of one unit (prologue / exact loop): 16 MFMAs, plain loads
    auto bare_S = [&](auto kbc, int slot, int lim) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        if (lim < 64) w64_mask_init<KB>(s, lim, g);
        else {
#pragma unroll
            for (int e = 0; e < 16; ++e) s.st[KB][0][e] = 0.f, s.st[KB][1][e] = 0.f;
        }
        const char* base = smem + W64_K(slot) + g * 1024 + kperm * 16 + KB * 512;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8_t f = *(const bf16x8_t*)(base + kk * 2048);
            s.st[KB][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[0][kk], s.st[KB][0], 0, 0, 0);
            s.st[KB][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, qf[1][kk], s.st[KB][1], 0, 0, 0);
        }
    };
    auto bare_PV = [&](auto kbc, int slot) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        const char* base = smem + W64_V(slot) + g * 2048 + l31 * 16 + KB * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t f = *(const bf16x8_t*)(base + w64_voff(i));
            s.ot[i & 3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, s.pf[KB][0][i >> 2], s.ot[i & 3][0], 0, 0, 0);
            s.ot[i & 3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, s.pf[KB][1][i >> 2], s.ot[i & 3][1], 0, 0, 0);
        }
    };
    using KB0 = std::integral_constant<int, 0>;
    using KB1 = std::integral_constant<int, 1>;

    // ------------------------------------------------------------------------------------------
    // pipelined pass.  Iteration t = steps u = 2t (S(t,1) | P.V(t-1,1) | softmax S(t,0)) and
    // u = 2t+1 (S(t+1,0) | P.V(t,0) | softmax S(t,1)); tile t in slot t % 3.  Needs >= 3 FULL
    // tiles to have a steady state; shorter or all-ragged rows go straight to the exact loop.
    // ------------------------------------------------------------------------------------------
    bool exact_pass = nfull < 3;
    if (!exact_pass) {
#pragma unroll
        for (int n = 0; n < 4; ++n) dma_k(0, 0, n), dma_v(0, 0, n), dma_k(1, 1, n);
        fence();
#pragma unroll
        for (int n = 0; n < 4; ++n) dma_k(2, 2, n), dma_v(1, 1, n);     // iteration 0's refill
        bare_S(KB0{}, 0, 64);
        w64_softmax_exact<0>(s, c_log2);
        bare_S(KB1{}, 0, 64);
        // step u = 1: S(1,0) | P.V(0,0) | softmax S(0,1)
        prefetch(k_addr(1, 0), v_addr(0, 0));
        w64_step<0, 1, true, true>(s, qf, kf, vf, k_addr(1, 0), v_addr(0, 0), k_addr(1, 1), v_addr(0, 1), c_log2, W64NoDma());
        int s0 = 0, s1 = 1, s2 = 2;     // slots of tiles t-1, t, t+1
        int t = 1;
        // hot loop: tile t+1 must be full for step 2t+1, and K(t+2) is prefetched for the step after
        unsigned long long pf_fence = 0, pf_a = 0, pf_b = 0, pf_n = 0;
        for (; t + 1 < nfull; ++t) {
            const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            fence();                    // K(t+1), V(t) visible; everyone is past iteration t-1
            const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t: S(t,1) [K slot s1] | P.V(t-1,1) [V slot s0] | softmax S(t,0); refill K(t+2) -> slot s0
            w64_step<1, 1, true, true>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s0, 1), k_addr(s2, 0), v_addr(s1, 0), c_log2,
                                       [&](int n) __attribute__((always_inline)) {   // all 8 refill pieces here:
                                           if (n < 4) dma_k(t + 2, s0, n);            // K(t+2) -> slot of tile t-1,
                                           else dma_v(t + 1, s2, n - 4);              // V(t+1) -> slot of tile t-2;
                                       });                                            // step B gives them time to land
            const unsigned long long c2 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t+1: S(t+1,0) [K slot s2] | P.V(t,0) [V slot s1] | softmax S(t,1); refill V(t+1) -> slot s2
            w64_step<0, 1, true, true>(s, qf, kf, vf, k_addr(s2, 0), v_addr(s1, 0), k_addr(s2, 1), v_addr(s1, 1), c_log2,
                                       W64NoDma());
            if (PROF) {
                const unsigned long long c3 = __builtin_amdgcn_s_memtime();
                pf_fence += c1 - c0, pf_a += c2 - c1, pf_b += c3 - c2, pf_n += 1;
            }
            const int tmp = s0;
            s0 = s1, s1 = s2, s2 = tmp;
        }
        if (PROF && prof && lane == 0) {
            atomicAdd(prof + wave * 4 + 0, pf_fence);
            atomicAdd(prof + wave * 4 + 1, pf_a);
            atomicAdd(prof + wave * 4 + 2, pf_b);
            atomicAdd(prof + wave * 4 + 3, pf_n);
        }
        // here t == nfull - 1 (last full tile), S(t,0) is complete, P(t-1,1) is ready, ring primed for u = 2t
        fence();
        if (t + 1 < T) {                // a ragged tile t+1 follows: its K and V are staged now (slot s2; V via s2 too)
#pragma unroll
            for (int n = 0; n < 4; ++n) dma_v(t + 1, s2, n);
        }
        w64_step<1, 1, true, true>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s0, 1), k_addr(s2, 0), v_addr(s1, 0), c_log2, W64NoDma());
        if (t + 1 < T) {
            // u = 2t+1 with the masked start for S(t+1,0)
            w64_mask_init<0>(s, last_lim, g);
            w64_step<0, 2, true, true>(s, qf, kf, vf, k_addr(s2, 0), v_addr(s1, 0), k_addr(s2, 1), v_addr(s1, 1), c_log2, W64NoDma());
            fence();                    // V(t+1) landed
            // u = 2t+2: S(t+1,1) masked | P.V(t,1) | softmax S(t+1,0)
            w64_mask_init<1>(s, last_lim, g);
            w64_step<1, 2, true, true>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s1, 1), 0, v_addr(s2, 0), c_log2, W64NoDma());
            // u = 2t+3: P.V(t+1,0) | softmax S(t+1,1)
            w64_step<0, 0, true, true>(s, qf, kf, vf, 0, v_addr(s2, 0), 0, v_addr(s2, 1), c_log2, W64NoDma());
            // u = 2t+4: P.V(t+1,1)
            w64_step<1, 0, true, false>(s, qf, kf, vf, 0, v_addr(s2, 1), 0, 0, c_log2, W64NoDma());
        } else {
            // u = 2t+1: P.V(t,0) | softmax S(t,1)
            w64_step<0, 0, true, true>(s, qf, kf, vf, 0, v_addr(s1, 0), 0, v_addr(s1, 1), c_log2, W64NoDma());
            // u = 2t+2: P.V(t,1)
            w64_step<1, 0, true, false>(s, qf, kf, vf, 0, v_addr(s1, 1), 0, 0, c_log2, W64NoDma());
        }
        w64_wait<0>();
        // the last prefetches of the chain are never consumed: keep the ring alive until they have landed
        asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
        exact_pass = __syncthreads_or(s.bad) != 0;      // workgroup-uniform: the exact loop has barriers
        if (exact_pass && flagcnt && tid == 0) atomicAdd(flagcnt, 1u);      // debug hook: how many blocks were redone
        if (dbg & 1) exact_pass = false;                // debug: keep the pipelined result even when flagged
    }
    // ------------------------------------------------------------------------------------------
    // exact pass: plain one-slot loop, true maxima; short rows, and blocks whose pipelined pass flagged
    // ------------------------------------------------------------------------------------------
    if (exact_pass) {
        reset();
        rot = 0;
        for (int t = 0; t < T; ++t) {
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 4; ++n) dma_k(t, 0, n), dma_v(t, 0, n);
            fence();
            const int lim = t == T - 1 ? last_lim : 64;
            bare_S(KB0{}, 0, lim);
            bare_S(KB1{}, 0, lim);
            w64_softmax_exact<0>(s, c_log2);    // P of a unit is relative to the maximum at ITS softmax:
            bare_PV(KB0{}, 0);                  // it must reach O^T before the next rescale
            w64_softmax_exact<1>(s, c_log2);
            bare_PV(KB1{}, 0);
        }
    }

#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float l_tot = s.l_run[b] + __shfl_xor(s.l_run[b], 32, 64);
        const float inv = 1.f / l_tot;
        const int64_t qr = (int64_t)qb0 * W64_QB + wave * 64 + l31 + b * 32;
        if (lse && g == 0 && qr < Lq) lse[(int64_t)head * Lq + qr] = (s.m_run[b] * c_log2 + __log2f(l_tot)) * 0.6931471805599453f;
        if (qr < Lq) {
            uint16_t* op = o + qr * ldo + head * 128 + g * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    uint2 pk;
                    pk.x = pack_bf2(s.ot[d][b][rq * 4 + 0] * inv, s.ot[d][b][rq * 4 + 1] * inv);
                    pk.y = pack_bf2(s.ot[d][b][rq * 4 + 2] * inv, s.ot[d][b][rq * 4 + 3] * inv);
                    *(uint2*)(op + d * 32 + rq * 8) = pk;
                }
        }
    }
    }   // work loop
}

static int g_w64_dbg = 0;
static unsigned long long* g_w64_prof = nullptr;
static unsigned* g_w64_flagcnt = nullptr;
void mg_attn_m16_hooks(int dbg, unsigned long long* prof, unsigned* flagcnt);
static void w64_sync_hooks() { mg_attn_m16_hooks(g_w64_dbg, g_w64_prof, g_w64_flagcnt); }   // the m16 kernel shares the three hooks
extern "C" void mg_attn_w64_profile(unsigned long long* dev_buf) { g_w64_prof = dev_buf; w64_sync_hooks(); }   // debug hook: 4 waves x {fence, step A, step B, iterations}
extern "C" void mg_attn_w64_debug(int flags) { g_w64_dbg = flags; w64_sync_hooks(); }
extern "C" void mg_attn_w64_flag_counter(unsigned* dev_counter) { g_w64_flagcnt = dev_counter; w64_sync_hooks(); }   // debug hook: += query blocks redone by the exact pass

int mg_attn_w64_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, float* lse, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (96 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int total = nqb * heads;
    const unsigned grid = total <= n_cu ? (unsigned)total : (unsigned)n_cu;   // persistent when there is more work than CUs
    if (g_w64_prof)
        hipLaunchKernelGGL((attn_hd128_w64_kernel<true>), dim3(grid), dim3(W64_THREADS), 0, st, q, ldq, kp, vp, o, ldo, Lq, Lk,
                           heads, c_log2, nqb, g_w64_dbg, g_w64_prof, lse, g_w64_flagcnt);
    else
        hipLaunchKernelGGL((attn_hd128_w64_kernel<false>), dim3(grid), dim3(W64_THREADS), 0, st, q, ldq, kp, vp, o, ldo, Lq, Lk,
                           heads, c_log2, nqb, g_w64_dbg, nullptr, lse, g_w64_flagcnt);
    return mg_check_launch();
}
