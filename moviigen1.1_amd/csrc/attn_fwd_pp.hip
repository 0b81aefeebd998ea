// Flash-style attention forward, head_dim 128 — PING-PONG schedule (variant 1 of
// mg_attn_fwd_bf16_hd128; same contract and data layout as attn_fwd.hip, see there).
//
// Why: a CU runs this kernel with 8 waves = 2 per SIMD.  In the plain schedule both waves of a
// SIMD execute the same phase at the same time (S^T MFMAs, then softmax VALU, then P.V MFMAs):
// the matrix pipe idles while both do softmax, the VALU idles while both queue MFMAs.
// Here the workgroup is split into group A (waves 0-3, one per SIMD) and group B (waves 4-7) that
// run HALF A TILE OUT OF PHASE, separated by workgroup barriers:
//
//      phase 1 of iteration t            phase 2 of iteration t
//   A: softmax(t)                        P.V(t) ; S^T(t+1)            <- 32 MFMAs
//   B: P.V(t-1) ; S^T(t)   <- 32 MFMAs   softmax(t)
//
// so on every SIMD exactly one wave is in a matrix segment while its partner is in its VALU
// segment.  S^T(t+1) is software-pipelined behind P.V(t) (its registers are dead after the
// softmax), which makes matrix and vector segments alternate 1:1.
//
// LDS: K and V^T tiles stay double-buffered (slot = tile & 1).  The slots of K(t) / V^T(t-1) are
// dead once phase 1(t) has ended and are next read in phase 2(t+1); the refill K(t+2) / V^T(t+1)
// is issued by every wave at the START of phase 2(t) as LDS-DMA (global_load_lds_dwordx4: no
// staging VGPRs — this schedule has none to spare — and no ds_write pass) and is retired by the
// vmcnt(0) that precedes the barrier closing phase 2(t).  The DMA image is lane-linear per wave
// instruction (K: 4 rows x 256 B, V^T: 8 rows x 128 B); the bank-conflict swizzle is applied to
// the per-lane SOURCE address and undone in the ds_read_b128 addresses.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define ATT_THREADS 512
#define ATT_QB 256
#define ATT_KV 64
#define K_TILE_BYTES (ATT_KV * 256)
#define V_TILE_BYTES (128 * 128)
#define K_OFF(slot) ((slot) * K_TILE_BYTES)
#define V_OFF(slot) (2 * K_TILE_BYTES + (slot) * V_TILE_BYTES)

typedef const __attribute__((address_space(1))) void* pp_gptr_t;
typedef __attribute__((address_space(3))) void* pp_lptr_t;

MG_DEV bf16x8_t as_bf16x8_pp(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
MG_DEV void pp_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((pp_gptr_t)g, (pp_lptr_t)l, 16, 0, 0);
}

template <bool LAZY>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd_hd128_pp_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k, int64_t ldk,
    const uint16_t* __restrict__ vt, int64_t ldvt, uint16_t* __restrict__ o, int64_t ldo, int64_t Lq,
    int64_t Lk, int heads, float c_log2, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * V_TILE_BYTES];

    const int bid = blockIdx.x;
    const int head = bid / nqb;
    const int qb = bid - head * nqb;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0 = A, 1 = B (the second wave of each SIMD)
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;

    const int64_t qrow_raw = (int64_t)qb * ATT_QB + wave * 32 + l31;
    const int64_t qrow = qrow_raw < Lq ? qrow_raw : Lq - 1;
    bf16x8_t qf[8];
    {
        const uint16_t* qp = q + qrow * ldq + head * 128 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = as_bf16x8_pp(*(const u32x4_t*)(qp + kk * 16));
    }

    // ---- LDS-DMA staging: wave w issues K instructions 2w, 2w+1 (4 rows each) and V^T
    //      instructions 2w, 2w+1 (8 rows each); swizzle on the source chunk index ----------------
    const int nkv = (int)((Lk + ATT_KV - 1) / ATT_KV);
    const int kr0 = wave * 8 + (lane >> 4);                 // K tile row of instruction 0 (+4 for 1)
    const int kp = lane & 15;                               // 16-B position within the 256-B row
    const uint16_t* k_src0 = k + head * 128 + ((kp ^ (kr0 & 15)) << 3);
    const uint16_t* k_src1 = k + head * 128 + ((kp ^ ((kr0 + 4) & 15)) << 3);
    const int vd0 = wave * 16 + (lane >> 3);                // V^T tile row d of instruction 0 (+8 for 1)
    const int vp = lane & 7;
    const uint16_t* v_src0 = vt + ((int64_t)head * 128 + vd0) * ldvt + ((vp ^ ((vd0 >> 1) & 7)) << 3);
    const uint16_t* v_src1 = vt + ((int64_t)head * 128 + vd0 + 8) * ldvt + ((vp ^ (((vd0 + 8) >> 1) & 7)) << 3);
    auto dma_k = [&](int t) {
        if (t < nkv) {
            int64_t r0 = (int64_t)t * ATT_KV + kr0, r1 = r0 + 4;
            if (r0 > Lk - 1) r0 = Lk - 1;
            if (r1 > Lk - 1) r1 = Lk - 1;
            char* dst = smem + K_OFF(t & 1) + wave * 8 * 256;
            pp_glds16(k_src0 + r0 * ldk, dst);
            pp_glds16(k_src1 + r1 * ldk, dst + 4 * 256);
        }
    };
    auto dma_v = [&](int t) {
        if (t < nkv) {
            char* dst = smem + V_OFF(t & 1) + wave * 16 * 128;
            pp_glds16(v_src0 + (int64_t)t * ATT_KV, dst);
            pp_glds16(v_src1 + (int64_t)t * ATT_KV, dst + 8 * 128);
        }
    };

    const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int k_roff = kperm * 256;
    const int k_t = g ^ (kperm & 15);
    const int v_roff = l31 * 128;
    const int v_t = g ^ ((l31 >> 1) & 7);

    f32x16_t ot[4], st[2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[d][e] = 0.f;
    bf16x8_t pf[2][2];
    float m_run = -1e30f, l_run = 0.f;

    auto qk = [&](int t) {  // S^T(t) = K(t).Q^T
        const char* kb_ = smem + K_OFF(t & 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8_t kf = *(const bf16x8_t*)(kb_ + kb * 32 * 256 + k_roff + ((k_t ^ (kk << 1)) << 4));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
            }
        }
    };
    auto pv = [&](int t) {  // O^T += V^T(t).P^T
        const char* vb_ = smem + V_OFF(t & 1);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8_t vf =
                        *(const bf16x8_t*)(vb_ + v_roff + d * 32 * 128 + ((v_t ^ (kb * 4 + s * 2)) << 4));
                    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], ot[d], 0, 0, 0);
                }
    };
    auto softmax = [&](int t) {  // st -> pf, running (m, l), deferred rescale of O^T
        const int64_t kv0 = (int64_t)t * ATT_KV;
        const int lim = (int)((Lk - kv0) < ATT_KV ? (Lk - kv0) : ATT_KV);
        if (lim < ATT_KV) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * 32 + (r >> 3) * 16 + g * 8 + (r & 7);
                    if (key >= lim) st[kb][r] = -1e30f;
                }
        }
        float tmax = st[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, st[1][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        float m_new = fmaxf(m_run, tmax);
        bool rescale = true;
        if (LAZY) {
            rescale = !__all((tmax - m_run) * c_log2 <= 8.f);
            if (!rescale) m_new = m_run;
        }
        const float mc = m_new * c_log2;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(st[kb][r] * c_log2 - mc);
                psum += p[r];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4_t w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[s * 8 + 2 * e], p[s * 8 + 2 * e + 1]);
                pf[kb][s] = as_bf16x8_pp(w);
            }
        }
        if (rescale) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_log2);
            l_run *= alpha;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[d][e] *= alpha;
            m_run = m_new;
        }
        l_run += psum;
    };

    // ---- prologue: K(0), V(0), K(1) resident -----------------------------------------------------
    dma_k(0);
    dma_v(0);
    dma_k(1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));  // retire the Q loads before the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Both groups execute the SAME instruction stream  [vector segment | barrier | matrix segment |
    // barrier] per tile; group B passes one extra barrier first, so it runs exactly one segment
    // behind A: in every barrier interval one wave of each SIMD is in its matrix segment and its
    // partner in its vector segment.  Interval 2t+1 is where K(t)'s and V^T(t-1)'s slots die
    // (last read by B's matrix segment in interval 2t) and A's matrix / B's vector segment of tile
    // t begin: every wave issues its share of the refill DMA K(t+2), V^T(t+1) right there; the
    // vmcnt(0) ahead of the next barrier retires it, two intervals before its first reader.
    if (grp == 1) __syncthreads();
    qk(0);
    __syncthreads();
    for (int t = 0; t < nkv; ++t) {
        // ---- vector segment: online softmax of tile t
        if (grp == 1) {
            dma_k(t + 2);
            dma_v(t + 1);
        }
        softmax(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- matrix segment: O^T += V^T(t).P^T ; S^T(t+1) = K(t+1).Q^T   (32 MFMAs)
        if (grp == 0) {
            dma_k(t + 2);
            dma_v(t + 1);
        }
        pv(t);
        if (t + 1 < nkv) qk(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (grp == 0) __syncthreads();

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow_raw < Lq) {
        uint16_t* op = o + qrow_raw * ldo + head * 128 + g * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2 pk;
                pk.x = pack_bf2(ot[d][rq * 4 + 0] * inv, ot[d][rq * 4 + 1] * inv);
                pk.y = pack_bf2(ot[d][rq * 4 + 2] * inv, ot[d][rq * 4 + 3] * inv);
                *(uint2*)(op + d * 32 + rq * 8) = pk;
            }
    }
}

int mg_attn_pp_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* vt,
                      int64_t ldvt, uint16_t* o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2,
                      int nqb, int lazy, hipStream_t st) {
    const dim3 grid((unsigned)(nqb * heads)), block(ATT_THREADS);
    if (lazy)
        hipLaunchKernelGGL(attn_fwd_hd128_pp_kernel<true>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo, Lq,
                           Lk, heads, c_log2, nqb);
    else
        hipLaunchKernelGGL(attn_fwd_hd128_pp_kernel<false>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo, Lq,
                           Lk, heads, c_log2, nqb);
    return mg_check_launch();
}
