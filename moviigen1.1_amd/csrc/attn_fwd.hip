// Flash-style attention forward, head_dim 128, non-causal, bf16 in/out, fp32 accumulate — gfx950.
//
// Replaces flash_attn_varlen_func as the reference calls it (wan/modules/attention.py:96-127)
// from WanSelfAttention (wan/modules/model.py:146-151; L = 75 600 .. 166 320 keys) and
// WanT2VCrossAttention (model.py:176; 512 keys).  72-85 % of all FLOPs of the path.
//
// Design (written for CDNA4, wave = 64, MFMA 32x32x16):
//  * workgroup = 8 waves = 256 queries of one head; each wave owns 32 queries; the K/V stream of
//    that head is walked in 64-key tiles, double-buffered in LDS (2 x 32 KiB).
//  * S^T = K.Q^T is computed with K as the MFMA A-operand and Q^T (kept in registers for the whole
//    kernel) as the B-operand, so a lane owns ONE query (lane&31) and 16 keys per 32-key block:
//    the online-softmax row statistics are per-lane scalars, the row reduction is 31 register
//    ops + one cross-half exchange.
//  * K rows are fed to the MFMA in a permuted order (bits 2<->3 of the row index swapped), which
//    makes every lane's 8 consecutive accumulator registers cover 8 CONSECUTIVE keys.  P (bf16)
//    is then directly the B-operand of O^T = V^T.P^T with no cross-lane movement, and the
//    A-operand is a plain 16-byte read of V^T[d][key..key+7] — V is pre-transposed once per layer
//    (mg_transpose_v_bf16, 0.3 % of the attention time) instead of transposed per tile.
//  * O^T accumulators keep lane = query, so the alpha / 1/l rescales are per-lane scalars too.
//  * LDS tiles are XOR-swizzled in 16-byte chunks (K: chunk ^= row&15, 256-B rows; V^T:
//    chunk ^= (row>>1)&7, 128-B rows) so that every ds_read_b128 lane group touches 16 distinct
//    16-byte slots; global->LDS goes through registers (issue early / write late, one barrier
//    per tile) so HBM/L2 latency hides under the MFMA phase.
//  * workgroups are issued head-major: all CUs stream the same head's K/V (38.7 MB at 720p),
//    which then lives in L2 / Infinity Cache instead of being re-read from HBM per query tile.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define ATT_THREADS 512
#define ATT_QB 256
#define ATT_KV 64
#define K_TILE_BYTES (ATT_KV * 256)  // 64 keys x 128 d x 2 B
#define V_TILE_BYTES (128 * 128)     // 128 d x 64 keys x 2 B
#define BUF_BYTES (K_TILE_BYTES + V_TILE_BYTES)

MG_DEV bf16x8_t as_bf16x8(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }

template <bool LAZY>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_fwd_hd128_kernel(
    const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k, int64_t ldk,
    const uint16_t* __restrict__ vt, int64_t ldvt, uint16_t* __restrict__ o, int64_t ldo, int64_t Lq,
    int64_t Lk, int heads, float c_log2, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

    const int bid = blockIdx.x;
    const int head = bid / nqb;
    const int qb = bid - head * nqb;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;

    // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[query l31][16kk + 8g .. +7] ----
    const int64_t qrow_raw = (int64_t)qb * ATT_QB + wave * 32 + l31;
    const int64_t qrow = qrow_raw < Lq ? qrow_raw : Lq - 1;
    bf16x8_t qf[8];
    {
        const uint16_t* qp = q + qrow * ldq + head * 128 + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) qf[kk] = as_bf16x8(*(const u32x4_t*)(qp + kk * 16));
    }

    // ---- staging addresses (global -> registers -> swizzled LDS) -------------------------------
    // K tile: 1024 16-B chunks, thread handles ids tid and tid+512: row = id>>4, chunk = id&15
    // V^T tile: 1024 chunks: d = id>>3, chunk = id&7
    int k_row[2], k_dst[2], v_dst[2];
    const uint16_t* v_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * ATT_THREADS;
        const int row = id >> 4, c = id & 15;
        k_row[i] = row;
        k_dst[i] = row * 256 + ((c ^ (row & 15)) << 4);
        const int d = id >> 3, cv = id & 7;
        v_dst[i] = K_TILE_BYTES + d * 128 + ((cv ^ ((d >> 1) & 7)) << 4);
        v_src[i] = vt + ((int64_t)head * 128 + d) * ldvt + cv * 8;
    }
    const uint16_t* k_base = k + head * 128 + (tid & 15) * 8;

    u32x4_t kreg[2], vreg[2];
    auto load_tile = [&](int64_t kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t kr = kv0 + k_row[i];
            if (kr > Lk - 1) kr = Lk - 1;
            kreg[i] = *(const u32x4_t*)(k_base + kr * ldk);
            vreg[i] = *(const u32x4_t*)(v_src[i] + kv0);
        }
    };
    auto store_tile = [&](int buf) {
        char* b = smem + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4_t*)(b + k_dst[i]) = kreg[i];
            *(u32x4_t*)(b + v_dst[i]) = vreg[i];
        }
    };

    // ---- fragment read offsets ------------------------------------------------------------------
    const int kperm = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);  // swap bits 2,3
    const int k_roff = kperm * 256;
    const int k_t = g ^ (kperm & 15);          // chunk = k_t ^ (kk<<1)
    const int v_roff = K_TILE_BYTES + l31 * 128;
    const int v_t = g ^ ((l31 >> 1) & 7);      // chunk = v_t ^ (4kb+2s)

    f32x16_t ot[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[d][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nkv = (int)((Lk + ATT_KV - 1) / ATT_KV);
    load_tile(0);
    store_tile(0);
    // Retire the Q-fragment loads HERE: an (empty) consumer of every fragment makes hipcc place
    // its vmcnt wait before the loop.  Otherwise the loop body inherits "Q may still be in
    // flight" and guards every S^T MFMA with s_waitcnt vmcnt(7..0) — which in steady state
    // drains the NEXT tile's prefetch at the top of each iteration instead of under the MFMAs.
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(qf[kk]));
    __syncthreads();

    for (int t = 0; t < nkv; ++t) {
        const int64_t kv0 = (int64_t)t * ATT_KV;
        if (t + 1 < nkv) load_tile(kv0 + ATT_KV);
        const char* kb_ = smem + (t & 1) * BUF_BYTES;

        // ---- S^T = K.Q^T : 2 key blocks x 8 d-steps ------------------------------------------------
        f32x16_t st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) st[kb][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8_t kf =
                    *(const bf16x8_t*)(kb_ + kb * 32 * 256 + k_roff + ((k_t ^ (kk << 1)) << 4));
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], st[kb], 0, 0, 0);
            }
        }

        // ---- online softmax (lane = query; regs = keys 32kb + 16(r>>3) + 8g + (r&7)) --------------
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
            // defer-max: keep the old running max while no query in the wave grew by more than 2^8
            rescale = !__all((tmax - m_run) * c_log2 <= 8.f);
            if (!rescale) m_new = m_run;
        }
        const float mc = m_new * c_log2;
        float psum = 0.f;
        bf16x8_t pf[2][2];
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
                pf[kb][s] = as_bf16x8(w);
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

        // ---- O^T += V^T.P^T : 4 d-blocks x (2 key blocks x 2 k-steps of 16 keys) ------------------
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8_t vf = *(const bf16x8_t*)(kb_ + v_roff + d * 32 * 128 +
                                                            ((v_t ^ (kb * 4 + s * 2)) << 4));
                    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][s], ot[d], 0, 0, 0);
                }
        }

        if (t + 1 < nkv) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane = query, regs = d (32 dblk + 8 rq + 4 g + e) ------------------------------
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
                      int nqb, int lazy, hipStream_t st);  // attn_fwd_pp.hip

static int g_attn_lazy = 1;
static int g_attn_variant = 0;  // 0 = lock-step schedule (this file, fastest: 969 TF/s), 1 = ping-pong (attn_fwd_pp.hip, 829 TF/s)
extern "C" void mg_attn_set_lazy_rescale(int on) { g_attn_lazy = on; }
extern "C" void mg_attn_set_variant(int v) { g_attn_variant = v; }

extern "C" int mg_attn_fwd_bf16_hd128(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                                      const uint16_t* vt, int64_t ldvt, uint16_t* o, int64_t ldo,
                                      int64_t Lq, int64_t Lk, int heads, float scale, void* stream) {
    if (!q || !k || !vt || !o) return MG_ERR_ARG;
    if (Lq < 0 || Lk <= 0 || heads <= 0) return MG_ERR_SHAPE;
    if ((ldq & 7) || (ldk & 7) || (ldvt & 63) || ldvt < Lk || (ldo & 3)) return MG_ERR_SHAPE;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)o & 7))
        return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    const int64_t nqb64 = (Lq + ATT_QB - 1) / ATT_QB;
    if (nqb64 * heads > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int nqb = (int)nqb64;
    const float c_log2 = scale * 1.4426950408889634f;
    const dim3 grid((unsigned)(nqb * heads)), block(ATT_THREADS);
    hipStream_t st = (hipStream_t)stream;
    if (g_attn_variant == 1)
        return mg_attn_pp_launch(q, ldq, k, ldk, vt, ldvt, o, ldo, Lq, Lk, heads, c_log2, nqb, g_attn_lazy, st);
    if (g_attn_lazy)
        hipLaunchKernelGGL(attn_fwd_hd128_kernel<true>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo,
                           Lq, Lk, heads, c_log2, nqb);
    else
        hipLaunchKernelGGL(attn_fwd_hd128_kernel<false>, grid, block, 0, st, q, ldq, k, ldk, vt, ldvt, o, ldo,
                           Lq, Lk, heads, c_log2, nqb);
    return mg_check_launch();
}
