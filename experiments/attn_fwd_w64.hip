// Flash-style attention forward, head_dim 128 — "one wave per SIMD" schedule (variant 2 of
// mg_attn_fwd_bf16_hd128; same contract, math and LDS image as attn_fwd.hip).
//
// PMC on the 8-wave kernel (profiles/r01_pmc_attn_v0.txt): matrix pipe 46 % busy, no LDS bank
// conflicts, LDS 29 % busy — the two waves of a SIMD run softmax at the same time and nothing
// feeds the pipe meanwhile.  This variant gets the overlap INSIDE one instruction stream:
//
//  * workgroup = 4 waves (one per SIMD, whole 512-entry register file each), wave = 64 queries =
//    two 32-query blocks.  Every K / V^T fragment read from LDS feeds TWO MFMAs (one per block):
//    half the LDS reads per MFMA.
//  * S^T is double-buffered in registers: iteration t issues the 32 MFMAs of S^T(t+1) = K(t+1).Q^T
//    while the VALU runs the online softmax of S^T(t) — independent instruction streams in one
//    basic block, so MFMA (asynchronous in the matrix pipe) and VALU overlap; then the 32 MFMAs of
//    O^T += V^T(t).P^T(t).
//  * K(t+2) / V^T(t+1) arrive by LDS-DMA issued at the top of iteration t into the slots that
//    K(t) / V^T(t-1) vacated in iteration t-1; one barrier per tile.
#include <type_traits>
#include "common.h"
#include "../../include/moviigen_hip.h"

#define W64_THREADS 256
#define W64_QB 256
#define W64_KV 64
#define K_TILE_BYTES (W64_KV * 256)
#define V_TILE_BYTES (128 * 128)
#define K_OFF(slot) ((slot) * K_TILE_BYTES)
#define V_OFF(slot) (2 * K_TILE_BYTES + (slot) * V_TILE_BYTES)

typedef const __attribute__((address_space(1))) void* w64_gptr_t;
typedef __attribute__((address_space(3))) void* w64_lptr_t;

MG_DEV bf16x8_t w64_bf(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
MG_DEV void w64_glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((w64_gptr_t)g, (w64_lptr_t)l, 16, 0, 0);
}

template <bool LAZY>
__global__ __launch_bounds__(W64_THREADS, 1) void attn_fwd_hd128_w64_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k, int64_t ldk,
    const uint16_t* __restrict__ vt, int64_t ldvt, uint16_t* __restrict__ o, int64_t ldo, int64_t Lq,
    int64_t Lk, int heads, float c_log2, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * V_TILE_BYTES];

    const int bid = blockIdx.x;
    const int head = bid / nqb;
    const int qb = bid - head * nqb;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;

    // ---- Q fragments for the wave's two 32-query blocks --------------------------------------------
    int64_t qrow_raw[2];
    bf16x8_t qf[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        qrow_raw[b] = (int64_t)qb * W64_QB + wave * 64 + b * 32 + l31;
        const int64_t qr = qrow_raw[b] < Lq ? qrow_raw[b] : Lq - 1;
        const uint16_t* qp = q + qr * ldq + head * 128 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[b][kk] = w64_bf(*(const u32x4_t*)(qp + kk * 16));
    }

    // ---- LDS-DMA staging: wave w issues K instructions 4w..4w+3 (4 rows each) and V^T
    //      instructions 4w..4w+3 (8 rows each); swizzle on the per-lane source chunk ---------------
    const int nkv = (int)((Lk + W64_KV - 1) / W64_KV);
    const int kp = lane & 15, vp = lane & 7;
    const uint16_t* k_src[4];
    const uint16_t* v_src[4];
    int k_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        k_row[i] = wave * 16 + i * 4 + (lane >> 4);
        k_src[i] = k + head * 128 + ((kp ^ (k_row[i] & 15)) << 3);
        const int d = wave * 32 + i * 8 + (lane >> 3);
        v_src[i] = vt + ((int64_t)head * 128 + d) * ldvt + ((vp ^ ((d >> 1) & 7)) << 3);
    }
    auto dma_k = [&](int t) __attribute__((always_inline)) {
        if (t < nkv) {
            char* dst = smem + K_OFF(t & 1) + wave * 16 * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int64_t r = (int64_t)t * W64_KV + k_row[i];
                if (r > Lk - 1) r = Lk - 1;
                w64_glds16(k_src[i] + r * ldk, dst + i * 4 * 256);
            }
        }
    };
    auto dma_v = [&](int t) __attribute__((always_inline)) {
        if (t < nkv) {
            char* dst = smem + V_OFF(t & 1) + wave * 32 * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) w64_glds16(v_src[i] + (int64_t)t * W64_KV, dst + i * 8 * 128);
        }
    };

    const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int k_roff = kperm * 256;
    const int k_t = g ^ (kperm & 15);
    const int v_roff = l31 * 128;
    const int v_t = g ^ ((l31 >> 1) & 7);

    f32x16_t ot[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[b][d][e] = 0.f;
    f32x16_t sA[2][2], sB[2][2];  // S^T double buffer: [block][key block]
    bf16x8_t pf[2][2][2];         // P (bf16): [block][key block][k-step]
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};

    auto qk = [&](f32x16_t (&s)[2][2], int slot) __attribute__((always_inline)) {
        const char* kb_ = smem + K_OFF(slot);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[b][kb][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8_t kf = *(const bf16x8_t*)(kb_ + kb * 32 * 256 + k_roff + ((k_t ^ (kk << 1)) << 4));
                s[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][kk], s[0][kb], 0, 0, 0);
                s[1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][kk], s[1][kb], 0, 0, 0);
            }
        }
    };
    auto pv = [&](int slot) __attribute__((always_inline)) {
        const char* vb_ = smem + V_OFF(slot);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8_t vf =
                        *(const bf16x8_t*)(vb_ + v_roff + d * 32 * 128 + ((v_t ^ (kb * 4 + s2 * 2)) << 4));
                    ot[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0][kb][s2], ot[0][d], 0, 0, 0);
                    ot[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1][kb][s2], ot[1][d], 0, 0, 0);
                }
    };
    auto softmax = [&](f32x16_t (&s)[2][2], int t) __attribute__((always_inline)) {
        const int64_t kv0 = (int64_t)t * W64_KV;
        const int lim = (int)((Lk - kv0) < W64_KV ? (Lk - kv0) : W64_KV);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (lim < W64_KV) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + (r >> 3) * 16 + g * 8 + (r & 7);
                        if (key >= lim) s[b][kb][r] = -1e30f;
                    }
            }
            float tmax = s[b][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[b][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[b][1][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            float m_new = fmaxf(m_run[b], tmax);
            bool rescale = true;
            if (LAZY) {
                rescale = !__all((tmax - m_run[b]) * c_log2 <= 8.f);
                if (!rescale) m_new = m_run[b];
            }
            const float mc = m_new * c_log2;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(s[b][kb][r] * c_log2 - mc);
                    psum += p[r];
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4_t w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = pack_bf2(p[s2 * 8 + 2 * e], p[s2 * 8 + 2 * e + 1]);
                    pf[b][kb][s2] = w64_bf(w);
                }
            }
            if (rescale) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new) * c_log2);
                l_run[b] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ot[b][d][e] *= alpha;
                m_run[b] = m_new;
            }
            l_run[b] += psum;
        }
    };

    // one tile: [DMA K(t+2), V^T(t+1)] ; { S^T(t+1) MFMAs  ||  softmax(t) VALU } ; P.V(t) MFMAs ; barrier
    auto tile = [&](int t, f32x16_t (&s_cur)[2][2], f32x16_t (&s_nxt)[2][2], auto parity) __attribute__((always_inline)) {
        constexpr int P = decltype(parity)::value;  // t & 1
        dma_k(t + 2);
        dma_v(t + 1);
        if (t + 1 < nkv) qk(s_nxt, 1 - P);
        softmax(s_cur, t);
        pv(P);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- prologue ---------------------------------------------------------------------------------------
    dma_k(0);
    dma_v(0);
    dma_k(1);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[b][kk]));  // retire the Q loads before the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qk(sA, 0);
    __syncthreads();  // nobody may still be reading K(0) when iteration 0 refills its slot

    int t = 0;
    for (; t + 1 < nkv; t += 2) {
        tile(t, sA, sB, std::integral_constant<int, 0>{});
        tile(t + 1, sB, sA, std::integral_constant<int, 1>{});
    }
    if (t < nkv) tile(t, sA, sB, std::integral_constant<int, 0>{});

    // ---- epilogue ---------------------------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.f / l_tot;
        if (qrow_raw[b] < Lq) {
            uint16_t* op = o + qrow_raw[b] * ldo + head * 128 + g * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    uint2 pk;
                    pk.x = pack_bf2(ot[b][d][rq * 4 + 0] * inv, ot[b][d][rq * 4 + 1] * inv);
                    pk.y = pack_bf2(ot[b][d][rq * 4 + 2] * inv, ot[b][d][rq * 4 + 3] * inv);
                    *(uint2*)(op + d * 32 + rq * 8) = pk;
                }
        }
    }
}

int mg_attn_w64_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* vt,
                       int64_t ldvt, uint16_t* o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float c_log2,
                       int nqb, int lazy, hipStream_t st) {
    const dim3 grid((unsigned)(nqb * heads)), block(W64_THREADS);
    if (lazy)
        hipLaunchKernelGGL(attn_fwd_hd128_w64_kernel<true>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo, Lq,
                           Lk, heads, c_log2, nqb);
    else
        hipLaunchKernelGGL(attn_fwd_hd128_w64_kernel<false>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo,
                           Lq, Lk, heads, c_log2, nqb);
    return mg_check_launch();
}
