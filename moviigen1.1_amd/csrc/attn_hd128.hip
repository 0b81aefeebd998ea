// Attention, head_dim 128: operand packing and the entry points (the kernels are attn_hd128_m16.hip and its A/B
// partner attn_hd128_w64.hip).
//
// Replaces flash_attn_varlen_func as the reference calls it (wan/modules/attention.py:96-127) from WanSelfAttention
// (wan/modules/model.py:146-151; L = 75 600 .. 166 320 keys) and WanT2VCrossAttention (model.py:176; 512 keys):
// 72-85 % of all FLOPs of the path.
//
// Operand layout.  Q and O are row-major [L][heads*128].  K and V arrive PRE-PACKED per 64-key tile (mg_pack_kv_bf16,
// one pass per layer, 0.4 % of the attention time):
//     kp[head][tile][c = d/8 (16)][row (64)][8]      vp[head][tile][kc = (key%64)/8 (8)][d (128)][8 keys]
// i.e. every tile is one contiguous 16 KiB block whose byte image IS the LDS image:
//   * staging is a pure contiguous LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, fully coalesced) —
//     no staging VGPRs, no ds_write pass, no swizzle arithmetic;
//   * an MFMA fragment is the 16-byte chunk (c, row): the lanes of a ds_read_b128 group read different rows of the
//     SAME chunk column = consecutive 16-byte slots: conflict-free by construction, and the address is ONE per-lane
//     base plus an immediate.
// Row order of a K tile (the kernel's S^T = K.Q^T puts key rows into accumulator registers, and P must come out as the
// B operand of O^T = V^T.P^T without any cross-lane shuffle — see attn_hd128_m16.hip):
//     m16 (default):  key 32u + 8a + 4b + j  sits in row  32u + 16b + 4a + j      (u < 2, a < 4, b < 2, j < 4)
//     w64 (mg_attn_set_variant(3)): natural order, row = key % 64 (that kernel permutes at read time)
// mg_pack_kv_bf16 writes the order of the kernel mg_attn_fwd_bf16_hd128* will run (same process-global selection).
#include "common.h"
#include "../../include/moviigen_hip.h"

#define ATT_QB 256      // queries per workgroup of both kernels

// -------------------------------------------------------------------------------------------------
// K / V -> packed tiles
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kv_kernel(const uint16_t* __restrict__ k, int64_t ldk,
                                                      const uint16_t* __restrict__ v, int64_t ldv, int64_t L,
                                                      uint16_t* __restrict__ kp, uint16_t* __restrict__ vp, int nt,
                                                      int korder) {
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
            const int row = korder ? ((r & 32) | ((r & 4) << 2) | ((r & 24) >> 1) | (r & 3)) : r;
            *(u16x8_t*)(kp + tbase + c * 512 + row * 8) = u;
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

#ifdef MG_AB_BUILD
int mg_attn_w64_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int nqb, float* lse, hipStream_t st);
#endif
int mg_attn_m16_launch(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                       int64_t Lq, int64_t Lk, int heads, float c_log2, int prescaled, int nqb, float* lse, int reserve_cus,
                       unsigned* workspace, hipStream_t st);

// The product library has ONE head-dim-128 kernel (m16) and no switch.  The A/B library (-DMG_AB_BUILD) adds the round-2 kernel (w64, NOT the
// same bits) behind mg_attn_set_variant — a process-global measurement switch that also selects the K row order mg_pack_kv_bf16 writes.
#ifdef MG_AB_BUILD
static int g_attn_variant = 0;      // 0 = m16 (default), 3 = w64 (round-2 kernel, A/B partner)
extern "C" int mg_attn_set_variant(int v) {
    if (v != 0 && v != 3) return MG_ERR_ARG;
    g_attn_variant = v;
    return MG_OK;
}
#else
static constexpr int g_attn_variant = 0;
#endif

extern "C" int mg_pack_kv_bf16(const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, int64_t L, int heads,
                               int head_dim, uint16_t* kp, uint16_t* vp, void* stream) {
    if ((!k && !v) || (k && !kp) || (v && !vp)) return MG_ERR_ARG;
    if (head_dim != 128 || heads <= 0 || L <= 0 || (k && (ldk & 7)) || (v && (ldv & 7))) return MG_ERR_SHAPE;
    const int nt = (int)((L + 63) / 64);
    hipLaunchKernelGGL(pack_kv_kernel, dim3(nt, heads), dim3(256), 0, (hipStream_t)stream, k, ldk, v, ldv, L, kp, vp,
                       nt, g_attn_variant == 3 ? 0 : 1);
    return mg_check_launch();
}

static int attn_fwd_impl(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp, uint16_t* o, int64_t ldo,
                         float* lse, int64_t Lq, int64_t Lk, int heads, float scale, int prescaled, int reserve_cus, void* workspace,
                         void* stream) {
    if (!q || !kp || !vp || !o) return MG_ERR_ARG;
    if ((uintptr_t)workspace & 7) return MG_ERR_SHAPE;
    if (Lq < 0 || Lk <= 0 || heads <= 0 || reserve_cus < 0) return MG_ERR_SHAPE;
    if ((ldq & 7) || (ldo & 3)) return MG_ERR_SHAPE;
    if (((uintptr_t)q & 15) || ((uintptr_t)kp & 15) || ((uintptr_t)vp & 15) || ((uintptr_t)o & 7)) return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    const int64_t nqb64 = (Lq + ATT_QB - 1) / ATT_QB;
    if (nqb64 * heads > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int nqb = (int)nqb64;
    const float c_log2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#ifdef MG_AB_BUILD
    if (g_attn_variant == 3)    // the round-2 kernel knows no pre-scaled q other than "its own factor is 1"
        return mg_attn_w64_launch(q, ldq, kp, vp, o, ldo, Lq, Lk, heads, prescaled ? 1.0f : c_log2, nqb, lse, st);
#endif
    return mg_attn_m16_launch(q, ldq, kp, vp, o, ldo, Lq, Lk, heads, c_log2, prescaled, nqb, lse, reserve_cus, (unsigned*)workspace, st);
}

// bytes of the caller-owned, zero-initialised workspace of mg_attn_fwd_bf16_hd128* (one {next ticket, workgroups done} pair, padded)
extern "C" int64_t mg_attn_workspace_bytes(void) { return 256; }

extern "C" int mg_attn_fwd_bf16_hd128_lse(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                          uint16_t* o, int64_t ldo, float* lse, int64_t Lq, int64_t Lk, int heads,
                                          float scale, void* workspace, void* stream) {
    return attn_fwd_impl(q, ldq, kp, vp, o, ldo, lse, Lq, Lk, heads, scale, 0, 0, workspace, stream);
}
extern "C" int mg_attn_fwd_bf16_hd128(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                      uint16_t* o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float scale,
                                      void* workspace, void* stream) {
    return attn_fwd_impl(q, ldq, kp, vp, o, ldo, nullptr, Lq, Lk, heads, scale, 0, 0, workspace, stream);
}
extern "C" int mg_attn_fwd_bf16_hd128_prescaled(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                                uint16_t* o, int64_t ldo, float* lse, int64_t Lq, int64_t Lk, int heads,
                                                int reserve_cus, void* workspace, void* stream) {
    return attn_fwd_impl(q, ldq, kp, vp, o, ldo, lse, Lq, Lk, heads, 1.0f, 1, reserve_cus, workspace, stream);
}
