// Ulysses sequence-parallel exchange layouts (HBM-bound block copies, 16-byte vectors).
//
// The reference moves q, k, v from "my tokens, all heads" to "all tokens, my heads" with three
// all-to-alls inside xfuser's xFuserLongContextAttention (wan/distributed/xdit_context_parallel.py:
// 185-190) and FastVideo's all_to_all_4D (scripts/train/model/model_seq.py:232-234,256), each of
// which reshapes/transposes its operand into a contiguous send buffer first.  Here ONE packed
// exchange per head group carries q, k and v together: the send buffer is written straight from
// the fused qkv activations as
//
//     send[dest p][token t][ q | k | v ][w]         w = columns of the head group (heads_g * head_dim)
//
// so that after all_to_all_single the receive buffer IS a row-major [P*Lloc tokens][3w] matrix in
// rank (= token) order whose column slices are the attention operands (row stride 3w) — no unpack
// pass on the receive side.  The attention output goes back as [dest p][token][w] blocks (already
// contiguous: rows p*Lloc.. of the output) and is scattered into the [Lloc][heads*head_dim] layout
// by the inverse copy.
#include "common.h"
#include "../../include/moviigen_hip.h"

// dst[b][r][0:8*vpr) = src[b][r][0:8*vpr): blocks b (stride *_blk), rows r (stride *_row), in elements.
__global__ void sp_copy_blocks_kernel(const uint16_t* __restrict__ src, int64_t s_blk, int64_t s_row,
                                      uint16_t* __restrict__ dst, int64_t d_blk, int64_t d_row, int64_t rows, int vpr) {
    const int64_t total = rows * vpr;
    const uint16_t* s = src + (int64_t)blockIdx.y * s_blk;
    uint16_t* d = dst + (int64_t)blockIdx.y * d_blk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / vpr;
        const int v = (int)(i - r * vpr);
        *reinterpret_cast<u32x4_t*>(d + r * d_row + v * 8) = *reinterpret_cast<const u32x4_t*>(s + r * s_row + v * 8);
    }
}

static int copy_blocks(const uint16_t* src, int64_t s_blk, int64_t s_row, uint16_t* dst, int64_t d_blk, int64_t d_row,
                       int blocks, int64_t rows, int width, hipStream_t st) {
    if (rows == 0 || blocks == 0) return MG_OK;
    const int vpr = width / 8;
    int64_t g = (rows * vpr + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(sp_copy_blocks_kernel, dim3((unsigned)g, (unsigned)blocks), dim3(256), 0, st, src, s_blk, s_row, dst,
                       d_blk, d_row, rows, vpr);
    return mg_check_launch();
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int mg_sp_pack_qkv_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v,
                                   int64_t ldv, int64_t Lloc, int P, int cols_per_dest, int col0, int w, uint16_t* send,
                                   void* stream) {
    if (!q || !k || !v || !send) return MG_ERR_ARG;
    if (Lloc < 0 || P <= 0 || w <= 0 || (w & 7) || (col0 & 7) || (cols_per_dest & 7) || col0 + w > cols_per_dest ||
        (ldq & 7) || (ldk & 7) || (ldv & 7) || !aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(send))
        return MG_ERR_SHAPE;
    const uint16_t* src[3] = {q, k, v};
    const int64_t ld[3] = {ldq, ldk, ldv};
    for (int i = 0; i < 3; ++i) {
        const int rc = copy_blocks(src[i] + col0, cols_per_dest, ld[i], send + (int64_t)i * w, Lloc * 3 * w, 3 * (int64_t)w, P,
                                   Lloc, w, (hipStream_t)stream);
        if (rc) return rc;
    }
    return MG_OK;
}

extern "C" int mg_sp_unpack_o_bf16(const uint16_t* recv, int64_t Lloc, int P, int cols_per_src, int col0, int w, uint16_t* o,
                                   int64_t ldo, void* stream) {
    if (!recv || !o) return MG_ERR_ARG;
    if (Lloc < 0 || P <= 0 || w <= 0 || (w & 7) || (col0 & 7) || (cols_per_src & 7) || col0 + w > cols_per_src || (ldo & 7) ||
        !aligned16(recv) || !aligned16(o))
        return MG_ERR_SHAPE;
    return copy_blocks(recv, Lloc * w, w, o + col0, cols_per_src, ldo, P, Lloc, w, (hipStream_t)stream);
}

extern "C" int mg_sp_copy_blocks_bf16(const uint16_t* src, int64_t s_blk, int64_t s_row, uint16_t* dst, int64_t d_blk,
                                      int64_t d_row, int blocks, int64_t rows, int width, void* stream) {
    if (!src || !dst) return MG_ERR_ARG;
    if (blocks < 0 || rows < 0 || width <= 0 || (width & 7) || (s_blk & 7) || (s_row & 7) || (d_blk & 7) || (d_row & 7) ||
        !aligned16(src) || !aligned16(dst))
        return MG_ERR_SHAPE;
    return copy_blocks(src, s_blk, s_row, dst, d_blk, d_row, blocks, rows, width, (hipStream_t)stream);
}
