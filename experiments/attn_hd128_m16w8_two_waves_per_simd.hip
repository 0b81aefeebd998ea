// Flash-style attention forward, head_dim 128 — schedule "m16w8": the m16 pipeline (attn_hd128_m16.hip: zero-reference
// softmax, v_mfma_f32_16x16x32_bf16, packed K/V tile images, LDS-DMA refill, fragment ring) with TWO waves per SIMD.
//
// Why (experiments/mfma_shape_probe.hip, profiles/r03k_mfma_shape_probe_8waves.log): a single wave cannot keep the MFMA
// pipe of its SIMD busy — every VALU / LDS / wait instruction it issues is a slot in which it does not issue an MFMA
// (m16: 77 % MFMA-busy cycles).  With a second wave on the SIMD the hardware interleaves the two instruction streams:
// the attention mix sustains 1676 TFLOP/s with 8 waves per CU against 1591 with 4, even though each wave then owns half
// the queries and every K / V fragment read feeds half as many MFMAs (0.5 instead of 0.25 ds_read_b128 per MFMA; LDS
// read bandwidth 50 % used).  Registers: 2 waves per SIMD leave 256 per lane — O^T 64, Q 32, S^T 32, P 16, rings 32.
//
//   workgroup = 512 threads = 8 waves x 32 queries (2 query blocks of 16) = the same 256-query block as m16: grid,
//   work loop, tile images and K row permutation are unchanged (mg_pack_kv_bf16 serves both).
//   wave w moves pieces 2w, 2w+1 of a 16-piece tile.  A 32-key unit is 8 groups of 4 MFMAs (S0 P0 S1 P1) per wave.
//
// MEASURED AND NOT ADOPTED (round 3, profiles/r03l_attn_m16w8.log).  Correct on the whole selftest shape set, and it does
// what it was built for — 2531 instead of 2647 s_memtime ticks per 64-key tile (-4.4 %) — but at L = 131040, 8 heads it
// runs 47.8 ms / 1472 TFLOP/s against m16's 46.2 ms / 1524: the doubled fragment-read traffic costs more clock under
// the power cap (-8 %) than the second wave wins in issue slots.  The probe's gain needs reads that cost nothing.
// Archived as built (it was wired as mg_attn_set_variant(4): mg_attn_m16w8_launch / mg_attn_m16w8_hooks); not compiled.
#include <type_traits>
#include "common.h"
#include "../../include/moviigen_hip.h"

#define P8_THREADS 512
#define P8_NQ 2            // query blocks of 16 per wave
#define P8_QB 256
#define P8_TILE 16384
#define P8_K(slot) ((slot) * P8_TILE)
#define P8_V(slot) (3 * P8_TILE + (slot) * P8_TILE)

typedef const __attribute__((address_space(1))) void* p8_gptr_t;
typedef __attribute__((address_space(3))) void* p8_lptr_t;
MG_DEV bf16x8_t p8_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
// piece n of a wave's two consecutive 1 KiB pieces: the immediate offset advances the global AND the LDS address
MG_DEV void p8_glds16_n(const void* g, void* l, int n) {
    if (n == 0) __builtin_amdgcn_global_load_lds((p8_gptr_t)g, (p8_lptr_t)l, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((p8_gptr_t)g, (p8_lptr_t)l, 16, 1024, 0);
}

struct P8State {
    f32x4_t ot[8][P8_NQ];      // O^T [d block][query block]             (AGPRs: builtin MFMAs)
    f32x4_t st[2][2][P8_NQ];   // S^T [unit kb][key block a/b][query block]  (arch VGPRs: inline-asm MFMAs)
    bf16x8_t pf[2][P8_NQ];     // P   [unit kb][query block]: keys 8G..8G+7 of the unit
    float m_run[P8_NQ], l_run[P8_NQ];
    int bad;
};

template <int OFF>
MG_DEV void p8_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void p8_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// S^T accumulators live in ARCH VGPRs (the softmax reads them with VALU instructions): inline asm with "v" operands,
// as in w64.  A block is written by groups 0-3 (a) / 4-7 (b) of a step and first read by the NEXT step's softmax.
MG_DEV void p8_mfma_s0(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {       // acc = a.b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
MG_DEV void p8_mfma_s(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {        // acc += a.b
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
MG_DEV void p8_mfma_s_after_valu(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {   // start value written by VALU (mask)
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
constexpr int p8_koff(int i) { return (i >> 2) * 256 + (i & 3) * 4096; }   // K fragment i = (key block i>>2, d chunk i&3)
constexpr int p8_voff(int i) { return i * 256; }                           // V fragment i = d block

// exact softmax of unit KB (true maximum of the 32 keys, rescale of O^T and l), all four query blocks
template <int KB>
MG_DEV void p8_softmax_exact(P8State& s, float c) {
#pragma unroll
    for (int n = 0; n < P8_NQ; ++n) {
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
        s.pf[KB][n] = p8_bf(w);
        const float alpha = __builtin_amdgcn_exp2f((s.m_run[n] - m_new) * c);
        s.l_run[n] = s.l_run[n] * alpha + psum;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) s.ot[d][n][e] *= alpha;
        s.m_run[n] = m_new;
    }
}

// softmax of unit KB against the fixed zero reference (prologue): p = 2^s
template <int KB>
MG_DEV void p8_softmax_zero(P8State& s, float c) {
#pragma unroll
    for (int n = 0; n < P8_NQ; ++n) {
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
        s.pf[KB][n] = p8_bf(w);
        s.l_run[n] += psum;
        s.m_run[n] = 0.f;
    }
}

// accumulator start of S^T for a ragged tile: -1e30 on the key rows >= lim (the MFMAs add K.Q^T to it)
template <int KB>
MG_DEV void p8_mask_init(P8State& s, int lim, int G) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int n = 0; n < P8_NQ; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = KB * 32 + G * 8 + blk * 4 + r;
                s.st[KB][blk][n][r] = key >= lim ? -1e30f : 0.f;
            }
}

// One pipeline step = one 32-key unit.  MFMAs: S^T of unit KB (K fragments at lds_k) when SMODE != 0 (2: masked start,
// s.st[KB] pre-set by p8_mask_init) and P.V of unit KB (V fragments at lds_v, P = s.pf[KB]) when PV.  VALU: softmax
// of unit 1-KB when SM.  `dma(i)` is called once per group (i = 0..7).  Fragment ring: 4 K + 4 V registers, reads two
// groups (16 MFMAs) ahead; the reads of the first two groups must have been issued by the caller (prefetch()), the last
// two groups of this step issue them for the NEXT step from nk / nv (0 = the clamped address of this step: data unused).
template <int KB, int SMODE, bool PV, bool SM, bool SCALED, typename Dma>
MG_DEV void p8_step(P8State& s, const bf16x8_t (&qf)[P8_NQ][4], bf16x8_t (&kf)[4], bf16x8_t (&vf)[4], unsigned lds_k,
                     unsigned lds_v, unsigned nk, unsigned nv, float c, Dma dma) {
    constexpr int SB = 1 - KB;          // unit being exponentiated
    float psa[P8_NQ] = {0.f, 0.f}, psb[P8_NQ] = {0.f, 0.f};
    u32x4_t w[P8_NQ];
    float pa[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
    // softmax of pair pp = i (8 pairs per unit, one per group; `half` is always 0): query block n = pp >> 2, packed word pp & 3 = (key block,
    // register pair); the pieces below are placed one by one into the eight MFMA gaps of a group
    auto exp_a = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = i, n = pp >> 2, wd = pp & 3;
            const float x = s.st[SB][wd >> 1][n][(wd & 1) * 2];
            pa[half] = __builtin_amdgcn_exp2f(SCALED ? x * c : x);
            asm volatile("" : "+v"(pa[half]));     // opaque use: pins the work HERE (LLVM sinks it otherwise)
        }
    };
    auto exp_b = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = i, n = pp >> 2, wd = pp & 3;
            const float x = s.st[SB][wd >> 1][n][(wd & 1) * 2 + 1];
            pb[half] = __builtin_amdgcn_exp2f(SCALED ? x * c : x);
            asm volatile("" : "+v"(pb[half]));
        }
    };
    auto add_a = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int n = i >> 2;
            psa[n] += pa[half];
            asm volatile("" : "+v"(psa[n]));
        }
    };
    auto add_b = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int n = i >> 2;
            psb[n] += pb[half];
            asm volatile("" : "+v"(psb[n]));
        }
    };
    auto cvt = [&](int i, int half) __attribute__((always_inline)) {
        if (SM) {
            const int pp = i, n = pp >> 2, wd = pp & 3;
            unsigned pk = pack_bf2(pa[half], pb[half]);
            asm volatile("" : "+v"(pk));
            w[n][wd] = pk;
        }
    };
    auto S = [&](int i, int n) __attribute__((always_inline)) {
        const int blk = i >> 2, c = i & 3, r = i & 3;
        if (SMODE == 1 && c == 0) p8_mfma_s0(s.st[KB][blk][n], kf[r], qf[n][c]);
        else if (SMODE == 2 && c == 0) p8_mfma_s_after_valu(s.st[KB][blk][n], kf[r], qf[n][c]);
        else if (SMODE != 0) p8_mfma_s(s.st[KB][blk][n], kf[r], qf[n][c]);
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
                case 0: p8_rd<p8_koff(2)>(kf[r2], lds_k); break;
                case 1: p8_rd<p8_koff(3)>(kf[r2], lds_k); break;
                case 2: p8_rd<p8_koff(4)>(kf[r2], lds_k); break;
                case 3: p8_rd<p8_koff(5)>(kf[r2], lds_k); break;
                case 4: p8_rd<p8_koff(6)>(kf[r2], lds_k); break;
                default: p8_rd<p8_koff(7)>(kf[r2], lds_k); break;
            }
        } else if (i == 6) p8_rd<p8_koff(0)>(kf[r2], nk);
        else p8_rd<p8_koff(1)>(kf[r2], nk);
    };
    auto rdV = [&](int i) __attribute__((always_inline)) {
        const int r2 = (i + 2) & 3;
        if (i < 6) {
            switch (i) {
                case 0: p8_rd<p8_voff(2)>(vf[r2], lds_v); break;
                case 1: p8_rd<p8_voff(3)>(vf[r2], lds_v); break;
                case 2: p8_rd<p8_voff(4)>(vf[r2], lds_v); break;
                case 3: p8_rd<p8_voff(5)>(vf[r2], lds_v); break;
                case 4: p8_rd<p8_voff(6)>(vf[r2], lds_v); break;
                default: p8_rd<p8_voff(7)>(vf[r2], lds_v); break;
            }
        } else if (i == 6) p8_rd<p8_voff(0)>(vf[r2], nv);
        else p8_rd<p8_voff(1)>(vf[r2], nv);
    };
#define P8_SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        p8_wait<2>();                          // K(i) and V(i) landed; younger: K(i+1), V(i+1)
        P8_SB();
        // Four MFMAs S0 P0 S1 P1 (asm / builtin alternate) and the gaps behind them: [exp exp] [add add cvt] [rdK rdV] [dma]
        // — one score pair per group; the second wave of the SIMD fills what this one leaves open.
        S(i, 0);
        P8_SB();
        exp_a(i, 0); exp_b(i, 0);
        P8_SB();
        P(i, 0);
        P8_SB();
        add_a(i, 0); add_b(i, 0); cvt(i, 0);
        P8_SB();
        S(i, 1);
        P8_SB();
        rdK(i);
        rdV(i);
        P8_SB();
        P(i, 1);
        P8_SB();
        dma(i);
        P8_SB();
    }
#undef P8_SB
    if (SM) {
#pragma unroll
        for (int n = 0; n < P8_NQ; ++n) {
            s.l_run[n] += psa[n] + psb[n];     // (l only grows, inf / NaN are sticky: ONE range test of the final sum)
            s.pf[SB][n] = p8_bf(w[n]);
        }
    }
}

struct P8NoDma {
    __device__ __forceinline__ void operator()(int) const {}
};

template <bool PROF, bool SCALED>
__global__ __launch_bounds__(P8_THREADS, 1) void attn_hd128_m16w8_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kp, const uint16_t* __restrict__ vp,
    uint16_t* __restrict__ o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, int dbg,
    unsigned long long* __restrict__ prof, float* __restrict__ lse, unsigned* __restrict__ flagcnt) {
    __shared__ __attribute__((aligned(16))) char smem[6 * P8_TILE];
    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, qi = lane & 15, G = lane >> 4;
    // persistent, XCD-aware work loop over (head, query block) items, head-major (see attn_hd128_w64.hip)
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

    // Q fragments (B operand): query block n of this wave = rows 64*wave + 16*n + qi, d = 32*c + 8*G .. +7
    bf16x8_t qf[P8_NQ][4];
    const int64_t qrow_base = (int64_t)qb0 * P8_QB + wave * 32 + qi;
#pragma unroll
    for (int n = 0; n < P8_NQ; ++n) {
        const int64_t qr = qrow_base + n * 16;
        const int64_t qrow = qr < Lq ? qr : Lq - 1;
        const uint16_t* qp = q + qrow * ldq + head * 128 + G * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) qf[n][c] = p8_bf(*(const u32x4_t*)(qp + c * 32));
    }
    const int T = (int)((Lk + 63) / 64);
    const int last_lim = (int)(Lk - (int64_t)(T - 1) * 64);     // keys in the last tile, 1..64
    // LDS-DMA: a tile is 16 pieces of 1 KiB; wave w moves pieces 4w..4w+3 of the K tile and of the V tile
    const char* k_src = (const char*)(kp + ((int64_t)head * T) * 8192 + wave * 1024);   // wave-uniform (SGPRs)
    const char* v_src = (const char*)(vp + ((int64_t)head * T) * 8192 + wave * 1024);
    const unsigned lane_off = lane * 16;                                                  // the only per-lane part
    const int nfull = last_lim == 64 ? T : T - 1;
    // tile indices past the end are clamped (a redundant reload of the last tile into a free slot) instead of guarded
    const unsigned lds0 = (unsigned)(uintptr_t)(p8_lptr_t)smem;
    auto dma_k = [&](int t, int slot, int n) __attribute__((always_inline)) {
        const int tt = t < T ? t : T - 1;
        p8_glds16_n(k_src + (int64_t)tt * 16384 + lane_off, smem + P8_K(slot) + wave * 2048, n);
    };
    auto dma_v = [&](int t, int slot, int n) __attribute__((always_inline)) {
        const int tt = t < T ? t : T - 1;
        p8_glds16_n(v_src + (int64_t)tt * 16384 + lane_off, smem + P8_V(slot) + wave * 2048, n);
    };
    const unsigned kbase = lds0 + G * 1024 + qi * 16;                  // + P8_K(slot) + kb*512 + blk*256 + c*4096
    const unsigned vbase = lds0 + 3 * P8_TILE + G * 2048 + qi * 16;   // + slot*TILE + kb*8192 + db*256
    auto k_addr = [&](int slot, int kb) __attribute__((always_inline)) { return kbase + slot * P8_TILE + kb * 512; };
    auto v_addr = [&](int slot, int kb) __attribute__((always_inline)) { return vbase + slot * P8_TILE + kb * 8192; };

    P8State s;
    bf16x8_t kf[4], vf[4];
    auto reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int n = 0; n < P8_NQ; ++n) s.ot[d][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < P8_NQ; ++n) s.m_run[n] = -1e30f, s.l_run[n] = 0.f;
        s.bad = 0;
    };
    reset();
#pragma unroll
    for (int n = 0; n < P8_NQ; ++n)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(qf[n][c]));
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
        p8_rd<p8_koff(0)>(kf[0], ak);
        p8_rd<p8_voff(0)>(vf[0], av);
        p8_rd<p8_koff(1)>(kf[1], ak);
        p8_rd<p8_voff(1)>(vf[1], av);
    };
    // bare S^T of one unit (prologue / exact loop): 32 MFMAs, plain loads
    auto bare_S = [&](auto kbc, int slot, int lim) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        if (lim < 64) p8_mask_init<KB>(s, lim, G);
        else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int n = 0; n < P8_NQ; ++n) s.st[KB][blk][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        const char* base = smem + P8_K(slot) + G * 1024 + qi * 16 + KB * 512;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t f = *(const bf16x8_t*)(base + p8_koff(i));
#pragma unroll
            for (int n = 0; n < P8_NQ; ++n)
                s.st[KB][i >> 2][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, qf[n][i & 3], s.st[KB][i >> 2][n], 0, 0, 0);
        }
    };
    auto bare_PV = [&](auto kbc, int slot) __attribute__((always_inline)) {
        constexpr int KB = decltype(kbc)::value;
        const char* base = smem + P8_V(slot) + G * 2048 + qi * 16 + KB * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8_t f = *(const bf16x8_t*)(base + p8_voff(i));
#pragma unroll
            for (int n = 0; n < P8_NQ; ++n)
                s.ot[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, s.pf[KB][n], s.ot[i][n], 0, 0, 0);
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
        for (int n = 0; n < P8_NQ; ++n) dma_k(0, 0, n), dma_v(0, 0, n), dma_k(1, 1, n);
        fence();
#pragma unroll
        for (int n = 0; n < P8_NQ; ++n) dma_k(2, 2, n), dma_v(1, 1, n);     // iteration 0's refill
        bare_S(KB0{}, 0, 64);
        p8_softmax_zero<0>(s, c_log2);
        bare_S(KB1{}, 0, 64);
        // step u = 1: S(1,0) | P.V(0,0) | softmax S(0,1)
        prefetch(k_addr(1, 0), v_addr(0, 0));
        p8_step<0, 1, true, true, SCALED>(s, qf, kf, vf, k_addr(1, 0), v_addr(0, 0), k_addr(1, 1), v_addr(0, 1), c_log2, P8NoDma());
        int s0 = 0, s1 = 1, s2 = 2;     // slots of tiles t-1, t, t+1
        int t = 1;
        unsigned long long pf_fence = 0, pf_a = 0, pf_b = 0, pf_n = 0;
        for (; t + 1 < nfull; ++t) {
            const unsigned long long c0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            fence_hot();                // K(t+1), V(t) visible; everyone is past iteration t-1
            const unsigned long long c1 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t: S(t,1) [K slot s1] | P.V(t-1,1) [V slot s0] | softmax S(t,0); refill K(t+2) -> slot s0, V(t+1) -> slot s2
            p8_step<1, 1, true, true, SCALED>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s0, 1), k_addr(s2, 0), v_addr(s1, 0), c_log2,
                                       [&](int n) __attribute__((always_inline)) {   // all 4 refill pieces here:
                                           if (n < 2) dma_k(t + 2, s0, n);            // K(t+2) -> slot of tile t-1,
                                           else if (n < 4) dma_v(t + 1, s2, n - 2);              // V(t+1) -> slot of tile t-2;
                                       });                                            // step B gives them time to land
            const unsigned long long c2 = PROF ? __builtin_amdgcn_s_memtime() : 0;
            // u = 2t+1: S(t+1,0) [K slot s2] | P.V(t,0) [V slot s1] | softmax S(t,1)
            p8_step<0, 1, true, true, SCALED>(s, qf, kf, vf, k_addr(s2, 0), v_addr(s1, 0), k_addr(s2, 1), v_addr(s1, 1), c_log2, P8NoDma());
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
        if (t + 1 < T) {                // a ragged tile t+1 follows: its V is staged now (slot s2)
#pragma unroll
            for (int n = 0; n < P8_NQ; ++n) dma_v(t + 1, s2, n);
        }
        p8_step<1, 1, true, true, SCALED>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s0, 1), k_addr(s2, 0), v_addr(s1, 0), c_log2, P8NoDma());
        if (t + 1 < T) {
            // u = 2t+1 with the masked start for S(t+1,0)
            p8_mask_init<0>(s, last_lim, G);
            p8_step<0, 2, true, true, SCALED>(s, qf, kf, vf, k_addr(s2, 0), v_addr(s1, 0), k_addr(s2, 1), v_addr(s1, 1), c_log2, P8NoDma());
            fence();                    // V(t+1) landed
            // u = 2t+2: S(t+1,1) masked | P.V(t,1) | softmax S(t+1,0)
            p8_mask_init<1>(s, last_lim, G);
            p8_step<1, 2, true, true, SCALED>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s1, 1), k_addr(s2, 1), v_addr(s2, 0), c_log2, P8NoDma());
            // u = 2t+3: P.V(t+1,0) | softmax S(t+1,1)
            p8_step<0, 0, true, true, SCALED>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s2, 0), k_addr(s2, 1), v_addr(s2, 1), c_log2, P8NoDma());
            // u = 2t+4: P.V(t+1,1)
            p8_step<1, 0, true, false, SCALED>(s, qf, kf, vf, k_addr(s2, 1), v_addr(s2, 1), k_addr(s2, 1), v_addr(s2, 1), c_log2, P8NoDma());
        } else {
            // u = 2t+1: P.V(t,0) | softmax S(t,1)
            p8_step<0, 0, true, true, SCALED>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s1, 0), k_addr(s1, 1), v_addr(s1, 1), c_log2, P8NoDma());
            // u = 2t+2: P.V(t,1)
            p8_step<1, 0, true, false, SCALED>(s, qf, kf, vf, k_addr(s1, 1), v_addr(s1, 1), k_addr(s1, 1), v_addr(s1, 1), c_log2, P8NoDma());
        }
        p8_wait<0>();
        // the last prefetches of the chain are never consumed: keep the ring alive until they have landed
        asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]));
#pragma unroll
        for (int n = 0; n < P8_NQ; ++n) {       // absolute scale: ONE range test of the final row sums (2^-60 .. 2^90; inf / NaN too)
            float lt = s.l_run[n] + __shfl_xor(s.l_run[n], 16, 64);
            lt += __shfl_xor(lt, 32, 64);
            s.bad |= !(lt >= 8.6736174e-19f && lt <= 1.2379400e27f);
        }
        exact_pass = __syncthreads_or(s.bad) != 0;      // workgroup-uniform: the exact loop has barriers
        if (exact_pass && flagcnt && tid == 0) atomicAdd(flagcnt, 1u);      // debug hook: how many blocks were redone
        if (dbg & 1) exact_pass = false;                // debug: keep the pipelined result even when flagged
    }
    // ------------------------------------------------------------------------------------------
    // exact pass: plain one-slot loop, true maxima; short rows, and blocks whose pipelined pass flagged
    // ------------------------------------------------------------------------------------------
    if (exact_pass) {
        reset();
        for (int t = 0; t < T; ++t) {
            __syncthreads();
#pragma unroll
            for (int n = 0; n < P8_NQ; ++n) dma_k(t, 0, n), dma_v(t, 0, n);
            fence();
            const int lim = t == T - 1 ? last_lim : 64;
            bare_S(KB0{}, 0, lim);
            bare_S(KB1{}, 0, lim);
            p8_softmax_exact<0>(s, c_log2);            // P of a unit is relative to the maximum at ITS softmax:
            bare_PV(KB0{}, 0);                  // it must reach O^T before the next rescale
            p8_softmax_exact<1>(s, c_log2);
            bare_PV(KB1{}, 0);
        }
    }

#pragma unroll
    for (int n = 0; n < P8_NQ; ++n) {
        float l_tot = s.l_run[n] + __shfl_xor(s.l_run[n], 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = 1.f / l_tot;
        const int64_t qr = (int64_t)qb0 * P8_QB + wave * 32 + n * 16 + qi;
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
    }   // work loop
}

static int g_p8_dbg = 0;
static unsigned long long* g_p8_prof = nullptr;
static unsigned* g_p8_flagcnt = nullptr;
void mg_attn_m16w8_hooks(int dbg, unsigned long long* prof, unsigned* flagcnt) { g_p8_dbg = dbg, g_p8_prof = prof, g_p8_flagcnt = flagcnt; }

// c_log2 = scale*log2(e) of the scores; prescaled != 0: q already carries that factor (mg_rmsnorm_rope_bf16 out_scale).
// kp must be in the m16 row order (mg_pack_kv_bf16 with the m16 kernel selected).
int mg_attn_m16w8_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int prescaled, int nqb, float* lse, hipStream_t st) {
    int n_cu = mg_cu_count();
    if (n_cu < 0) return MG_ERR_LAUNCH;
    n_cu &= ~7;                                         // one workgroup per CU (96 KiB LDS), a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    const int total = nqb * heads;
    const unsigned grid = total <= n_cu ? (unsigned)total : (unsigned)n_cu;   // persistent when there is more work than CUs
#define P8_LAUNCH(PROF, SCALED)                                                                                             \
    hipLaunchKernelGGL((attn_hd128_m16w8_kernel<PROF, SCALED>), dim3(grid), dim3(P8_THREADS), 0, st, q, ldq, kp, vp, o, ldo, Lq, \
                       Lk, heads, prescaled ? 1.0f : c_log2, nqb, g_p8_dbg, PROF ? g_p8_prof : nullptr, lse, g_p8_flagcnt)
    if (g_p8_prof) {
        if (prescaled) P8_LAUNCH(true, false);
        else P8_LAUNCH(true, true);
    } else {
        if (prescaled) P8_LAUNCH(false, false);
        else P8_LAUNCH(false, true);
    }
#undef P8_LAUNCH
    return mg_check_launch();
}
