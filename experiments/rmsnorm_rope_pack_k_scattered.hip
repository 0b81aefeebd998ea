// experiments/ — measured and not shipped (round 6).  RMS-norm + RoPE of the self-attention keys written STRAIGHT into the attention kernel's
// packed K tile layout by the wave-per-row kernel (a lane's 16-byte chunk of a row goes to [head][tile][c][row][8]: 64 lanes -> 64 different
// 1-KiB-strided 16-byte pieces per store instruction).  Bit-identical to mg_rmsnorm_rope_bf16 + mg_pack_kv_bf16(k, NULL) (test kept in
// git history: tests/test_gpu_parity.py::test_rmsnorm_rope_pack_k at commit "wave-per-row rmsnorm") — and SLOWER than the two passes it
// replaces: 1450 us against 536 (norm + rope, row-major) + ~407 (the k half of pack_kv) = 943 us at L = 131 040 x 5120
// (profiles/r06g_rowwise.log): the scattered 16-byte stores run at 1.85 TB/s.  Coalescing them needs 64 rows of one (head, chunk) column in
// one wave — i.e. the row-wise norm's transposed mapping, a second pass over the row or 640 KiB of LDS per tile; the whole re-layout pass
// costs 0.4 ms of a 270 ms layer.  Not compiled; excerpt of csrc/dit_elementwise.hip as it was.

// Round 6: ONE WAVE PER ROW.  The kernel above gives a row to a 256-thread workgroup: two __syncthreads per row for the sum of squares, the
// (cos, sin) row staged through LDS behind a third, and `weight[col + j]` re-read from L1 for every row (20 KB of fp32 weights per 10 KB
// row): 3.2-4.3 TB/s at dim 5120.  Here a wave owns a row (dim <= 8192: <= 16 16-byte chunks per lane), the sum of
// squares is a wave reduction, the norm weights of the lane's columns live in registers for all rows the wave walks, and — 512 % head_dim == 0
// makes a lane's position inside its head the same for all its chunks — the lane loads its own 4 (cos, sin) pairs of the row's token straight
// from the tables (L1 / L2 resident, 33 KB at 21 x 52 x 120).  A wave streams: 10 loads, one reduction, 10 stores per row, nothing shared.
// PACK: the result goes straight into the attention kernel's packed K tile layout (mg_pack_kv_bf16's kp: [head][tile][c = d/8][row (64)][8],
// key 32u + 8a + 4b + j of a tile in row 32u + 16b + 4a + j) instead of row-major — the k-side re-layout pass of round 5 is gone; rows past
// `rows` up to the end of the last 64-key tile are written as zeros, as mg_pack_kv_bf16 does.  Needs head_dim == 128.
template <int MAXN, bool PACK>
__global__ __launch_bounds__(NT) void rmsnorm_rope_wave_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out, int64_t ldo, int64_t rows, int64_t rows_out,
    int dim, const float* __restrict__ weight, float eps, int head_dim, const float2* __restrict__ rope_cs,
    int F, int H, int W, int64_t pos0, float out_scale, int nt) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nc = dim >> 3;
    const int c = head_dim >> 1, c1 = c / 3, c0 = c - 2 * c1;
    const float2* tab_f = rope_cs;
    const float2* tab_h = rope_cs ? rope_cs + (int64_t)F * c0 : nullptr;
    const float2* tab_w = rope_cs ? tab_h + (int64_t)H * c1 : nullptr;
    const int64_t grid_tokens = (int64_t)F * H * W;
    const int p0 = ((lane * 8) % head_dim) >> 1;            // first of the lane's 4 rotation pairs: the same for all its chunks (512 % head_dim == 0)
    float wgt[MAXN][8];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        const int ch = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) wgt[i][j] = ch < nc ? weight[ch * 8 + j] : 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * (NT / 64) + wave; row < rows_out; row += (int64_t)gridDim.x * (NT / 64)) {
        u32x4_t o[MAXN];
        if (row < rows) {
            const u16x8_t* xr = (const u16x8_t*)(x + row * ldx);
            u16x8_t u[MAXN];
#pragma unroll
            for (int i = 0; i < MAXN; ++i) {
                const int ch = lane + 64 * i;
                u[i] = ch < nc ? xr[ch] : (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            }
            const int64_t tok = pos0 + row;
            const bool do_rope = rope_cs != nullptr && tok < grid_tokens;
            float2 cs[4] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
            if (do_rope) {
                const int pf = (int)(tok / ((int64_t)H * W));
                const int rem = (int)(tok - (int64_t)pf * H * W);
                const int ph = rem / W, pw = rem - ph * W;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = p0 + j;
                    cs[j] = p < c0 ? tab_f[(int64_t)pf * c0 + p] : (p < c0 + c1 ? tab_h[(int64_t)ph * c1 + (p - c0)] : tab_w[(int64_t)pw * c1 + (p - c0 - c1)]);
                }
            }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < MAXN; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f(u[i][j]);
                    ss += f * f;
                }
            const float r = rsqrtf(wave_sum(ss) / (float)dim + eps);
#pragma unroll
            for (int i = 0; i < MAXN; ++i) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = round_bf(bf2f(u[i][j]) * r) * wgt[i][j];
                if (do_rope) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = y[2 * j], b = y[2 * j + 1];
                        y[2 * j] = a * cs[j].x - b * cs[j].y;
                        y[2 * j + 1] = a * cs[j].y + b * cs[j].x;
                    }
                }
                o[i][0] = pack_bf2(y[0] * out_scale, y[1] * out_scale);
                o[i][1] = pack_bf2(y[2] * out_scale, y[3] * out_scale);
                o[i][2] = pack_bf2(y[4] * out_scale, y[5] * out_scale);
                o[i][3] = pack_bf2(y[6] * out_scale, y[7] * out_scale);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXN; ++i) o[i] = (u32x4_t){0u, 0u, 0u, 0u};      // PACK only: keys past the sequence in the last tile
        }
        if (PACK) {
            const int r64 = (int)(row & 63);
            const int prow = (r64 & 32) | ((r64 & 4) << 2) | ((r64 & 24) >> 1) | (r64 & 3);
            uint16_t* tile = out + (row >> 6) * 8192 + prow * 8;
#pragma unroll
            for (int i = 0; i < MAXN; ++i) {
                const int ch = lane + 64 * i;                   // head = ch / 16, 16-byte chunk c = ch % 16 of the head's 128 dims
                if (ch < nc) *(u32x4_t*)(tile + (int64_t)(ch >> 4) * nt * 8192 + (ch & 15) * 512) = o[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXN; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nc) ((u32x4_t*)(out + row * ldo))[ch] = o[i];
            }
        }
    }
}

template <bool PACK>
static void launch_rmsnorm_rope_wave(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int64_t rows, int64_t rows_out, int dim,
                                     const float* weight, float eps, int head_dim, const float2* cs, int F, int H, int W, int64_t pos0,
                                     float out_scale, int nt, hipStream_t st) {
    const int nc = dim >> 3;
    int64_t grid = (rows_out + 3) / 4;
    if (grid > 2048) grid = 2048;                 // 8 workgroups per CU: every wave keeps its weights for ~rows / 8192 rows
#define RRW(N) hipLaunchKernelGGL((rmsnorm_rope_wave_kernel<N, PACK>), dim3((unsigned)grid), dim3(NT), 0, st, x, ldx, out, ldo, rows, rows_out, dim, \
                                  weight, eps, head_dim, cs, F, H, W, pos0, out_scale, nt)
    if (nc <= 64) RRW(1);
    else if (nc <= 256) RRW(4);
    else if (nc <= 640) RRW(10);
    else RRW(16);
#undef RRW
}

// WanRMSNorm + RoPE of the self-attention KEYS written straight into the packed tile layout of mg_attn_fwd_bf16_hd128* (what
// mg_rmsnorm_rope_bf16 + mg_pack_kv_bf16(k, NULL) produce, bit for bit, in one pass): x [rows][>= heads*128] -> kp [heads][ceil(rows/64)][8192];
// key index = row (pos0 only positions the rotation).  head_dim 128.
extern "C" int mg_rmsnorm_rope_pack_k_bf16(const uint16_t* x, int64_t ldx, uint16_t* kp, int64_t rows, int dim, const float* weight, float eps,
                                           int head_dim, const float* rope_cs, int F, int H, int W, int64_t pos0, void* stream) {
    if (rows == 0) return MG_OK;
    if (!x || !kp || !weight) return MG_ERR_ARG;
    if (rows < 0 || dim <= 0 || dim > 8192 || head_dim != 128 || dim % 128 || (ldx & 7)) return MG_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)kp) & 15) return MG_ERR_SHAPE;
    if (rope_cs && (F <= 0 || H <= 0 || W <= 0)) return MG_ERR_SHAPE;
    const int64_t nt = (rows + 63) / 64;
    if (nt > 0x7fffffffLL) return MG_ERR_SHAPE;
    launch_rmsnorm_rope_wave<true>(x, ldx, kp, 0, rows, nt * 64, dim, weight, eps, 128, (const float2*)rope_cs, F, H, W, pos0, 1.0f, (int)nt,
                                   (hipStream_t)stream);
    return mg_check_launch();
}

