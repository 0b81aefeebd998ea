// Token-wise (HBM-bound) kernels of the WanModel DiT forward for gfx950.
// One workgroup per token row, 16-byte vector accesses, whole row kept in registers
// between the reduction and the normalise/modulate pass (one HBM read, one HBM write).
//
// Reference arithmetic (file:line under the reference tree) is named at each kernel.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define NT 256

// ---------------------------------------------------------------------------------------------
// LayerNorm (no affine, eps) + modulate  — wan/modules/model.py:89-99, :299, :306, :307, :340-342
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(NT) void ln_modulate_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t rows, int dim, const float* __restrict__ scale,
    const float* __restrict__ shift, int add_one, float eps, int do_round, void* __restrict__ out,
    int out_f32, int64_t ldo) {
    __shared__ float red[NT / 64];
    const int nv = dim >> 2;
    const float inv_dim = 1.f / (float)dim;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float4* xr = (const float4*)(x + row * ldx);
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                v[i] = xr[c];
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            } else {
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float mean = block_sum<NT>(s, red) * inv_dim;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float var = block_sum<NT>(q, red) * inv_dim;
        const float rstd = rsqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < nv) {
                float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd,
                              (v[i].w - mean) * rstd};
                float4 sc = scale ? ((const float4*)scale)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
                float4 sh = shift ? ((const float4*)shift)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (scale && add_one) { sc.x += 1.f; sc.y += 1.f; sc.z += 1.f; sc.w += 1.f; }
                if (do_round) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = round_bf(y[j]);
                }
                const float o0 = y[0] * sc.x + sh.x, o1 = y[1] * sc.y + sh.y, o2 = y[2] * sc.z + sh.z,
                            o3 = y[3] * sc.w + sh.w;
                if (out_f32) {
                    ((float4*)((float*)out + row * ldo))[c] = make_float4(o0, o1, o2, o3);
                } else {
                    uint2 p;
                    p.x = pack_bf2(o0, o1);
                    p.y = pack_bf2(o2, o3);
                    ((uint2*)((uint16_t*)out + row * ldo))[c] = p;
                }
            }
        }
    }
}

extern "C" int mg_ln_modulate(const float* x, int64_t ldx, int64_t rows, int dim, const float* scale,
                              const float* shift, int add_one, float eps, int round_norm_bf16,
                              void* out, int out_f32, int64_t ldo, void* stream) {
    if (rows == 0) return MG_OK;  // empty input (data pointers of empty tensors are NULL)
    if (!x || !out) return MG_ERR_ARG;
    if (dim <= 0 || (dim & 3) || dim > 8192 || (ldx & 3) || (ldo & 3)) return MG_ERR_SHAPE;
    if (rows <= 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (int)(rows < 65536 * 4 ? rows : 65536 * 4);
    const int nv = dim >> 2;
    if (nv <= 2 * NT)
        hipLaunchKernelGGL(ln_modulate_kernel<2>, dim3(grid), dim3(NT), 0, st, x, ldx, rows, dim, scale,
                           shift, add_one, eps, round_norm_bf16, out, out_f32, ldo);
    else if (nv <= 5 * NT)
        hipLaunchKernelGGL(ln_modulate_kernel<5>, dim3(grid), dim3(NT), 0, st, x, ldx, rows, dim, scale,
                           shift, add_one, eps, round_norm_bf16, out, out_f32, ldo);
    else
        hipLaunchKernelGGL(ln_modulate_kernel<8>, dim3(grid), dim3(NT), 0, st, x, ldx, rows, dim, scale,
                           shift, add_one, eps, round_norm_bf16, out, out_f32, ldo);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// RMSNorm(dim) * weight  (+ 3-axis RoPE)  — model.py:70-86, :39-67; SP slice: xdit_context_parallel.py:23-62
// ---------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(NT) void rmsnorm_rope_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out, int64_t ldo, int64_t rows,
    int dim, const float* __restrict__ weight, float eps, int head_dim, const float2* __restrict__ rope_cs,
    int F, int H, int W, int64_t pos0, float out_scale) {
    __shared__ float red[NT / 64];
    __shared__ float2 cs_row[128];          // the token's (cos, sin) pairs, identical for every head: staged once per row
    const int nc = dim >> 3;
    const int c = head_dim >> 1, c1 = c / 3, c0 = c - 2 * c1;
    const float2* tab_f = rope_cs;
    const float2* tab_h = rope_cs ? rope_cs + (int64_t)F * c0 : nullptr;
    const float2* tab_w = rope_cs ? tab_h + (int64_t)H * c1 : nullptr;
    const int64_t grid_tokens = (int64_t)F * H * W;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const u16x8_t* xr = (const u16x8_t*)(x + row * ldx);
        float v[MAXC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = threadIdx.x + i * NT;
            if (ch < nc) {
                const u16x8_t u = xr[ch];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[i][j] = bf2f(u[j]);
                    ss += v[i][j] * v[i][j];
                }
            }
        }
        const int64_t tok = pos0 + row;
        const bool do_rope = rope_cs != nullptr && tok < grid_tokens;
        if (do_rope && (int)threadIdx.x < c) {       // c <= 128 pairs: [c0 temporal | c1 height | c1 width] of this token
            const int pf = (int)(tok / ((int64_t)H * W));
            const int rem = (int)(tok - (int64_t)pf * H * W);
            const int ph = rem / W, pw = rem - ph * W;
            const int p = threadIdx.x;
            cs_row[p] = p < c0 ? tab_f[(int64_t)pf * c0 + p]
                               : (p < c0 + c1 ? tab_h[(int64_t)ph * c1 + (p - c0)] : tab_w[(int64_t)pw * c1 + (p - c0 - c1)]);
        }
        const float r = rsqrtf(block_sum<NT>(ss, red) / (float)dim + eps);    // (its barriers also publish cs_row)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = threadIdx.x + i * NT;
            if (ch < nc) {
                const int col = ch << 3;
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = round_bf(v[i][j] * r) * weight[col + j];
                if (do_rope) {
                    const int p0 = (col % head_dim) >> 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float2 cs = cs_row[p0 + j];
                        const float a = y[2 * j], b = y[2 * j + 1];
                        y[2 * j] = a * cs.x - b * cs.y;
                        y[2 * j + 1] = a * cs.y + b * cs.x;
                    }
                }
                // out_scale (1 = none: exact): the attention's scale*log2(e) folded into q BEFORE its one rounding to bf16
                u32x4_t o;
                o[0] = pack_bf2(y[0] * out_scale, y[1] * out_scale);
                o[1] = pack_bf2(y[2] * out_scale, y[3] * out_scale);
                o[2] = pack_bf2(y[4] * out_scale, y[5] * out_scale);
                o[3] = pack_bf2(y[6] * out_scale, y[7] * out_scale);
                ((u32x4_t*)(out + row * ldo))[ch] = o;
            }
        }
        if (row + gridDim.x < rows) __syncthreads();     // cs_row is rewritten by the next row of this workgroup
    }
}

// Round 6: ONE WAVE PER ROW.  The kernel above gives a row to a 256-thread workgroup: two __syncthreads per row for the sum of squares, the
// (cos, sin) row staged through LDS behind a third, and `weight[col + j]` re-read from L1 for every row (20 KB of fp32 weights per 10 KB
// row): 3.2-4.3 TB/s at dim 5120.  Here a wave owns a row (dim <= 8192: <= 16 16-byte chunks per lane), the sum of
// squares is a wave reduction, the norm weights of the lane's columns live in registers for all rows the wave walks, and — 512 % head_dim == 0
// makes a lane's position inside its head the same for all its chunks — the lane loads its own 4 (cos, sin) pairs of the row's token straight
// from the tables (L1 / L2 resident, 33 KB at 21 x 52 x 120).  A wave streams: 10 loads, one reduction, 10 stores per row, nothing shared.
// (Writing the keys from here straight into the attention kernel's packed tile layout was built and measured: the scattered 16-byte stores make it
// slower than this kernel + the k half of mg_pack_kv_bf16 — experiments/rmsnorm_rope_pack_k_scattered.hip, profiles/r06g_rowwise.log.)
template <int MAXN>
__global__ __launch_bounds__(NT) void rmsnorm_rope_wave_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out, int64_t ldo, int64_t rows,
    int dim, const float* __restrict__ weight, float eps, int head_dim, const float2* __restrict__ rope_cs,
    int F, int H, int W, int64_t pos0, float out_scale) {
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
    for (int64_t row = (int64_t)blockIdx.x * (NT / 64) + wave; row < rows; row += (int64_t)gridDim.x * (NT / 64)) {
        u32x4_t o[MAXN];
        {
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
        }
#pragma unroll
        for (int i = 0; i < MAXN; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nc) ((u32x4_t*)(out + row * ldo))[ch] = o[i];
        }
    }
}

static void launch_rmsnorm_rope_wave(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int64_t rows, int dim,
                                     const float* weight, float eps, int head_dim, const float2* cs, int F, int H, int W, int64_t pos0,
                                     float out_scale, hipStream_t st) {
    const int nc = dim >> 3;
    int64_t grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;                 // 8 workgroups per CU: every wave keeps its weights for ~rows / 8192 rows
#define RRW(N) hipLaunchKernelGGL((rmsnorm_rope_wave_kernel<N>), dim3((unsigned)grid), dim3(NT), 0, st, x, ldx, out, ldo, rows, dim, \
                                  weight, eps, head_dim, cs, F, H, W, pos0, out_scale)
    if (nc <= 64) RRW(1);
    else if (nc <= 256) RRW(4);
    else if (nc <= 640) RRW(10);
    else RRW(16);
#undef RRW
}

extern "C" int mg_rmsnorm_rope_bf16(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo,
                                    int64_t rows, int dim, const float* weight, float eps, int head_dim,
                                    const float* rope_cs, int F, int H, int W, int64_t pos0, float out_scale,
                                    void* stream) {
    if (rows == 0) return MG_OK;
    if (!x || !out || !weight) return MG_ERR_ARG;
    if (dim <= 0 || (dim & 7) || dim > 8192 || (ldx & 7) || (ldo & 7) || head_dim <= 0 ||
        (head_dim & 7) || dim % head_dim)
        return MG_ERR_SHAPE;
    if (rope_cs && (F <= 0 || H <= 0 || W <= 0 || head_dim > 256)) return MG_ERR_SHAPE;
    if (rows <= 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (512 % head_dim == 0 && !(((uintptr_t)x | (uintptr_t)out) & 15)) {      // the wave-per-row kernel (every head_dim the model family uses)
        launch_rmsnorm_rope_wave(x, ldx, out, ldo, rows, dim, weight, eps, head_dim, (const float2*)rope_cs, F, H, W, pos0, out_scale, st);
        return mg_check_launch();
    }
    const int grid = (int)(rows < 65536 * 4 ? rows : 65536 * 4);
    const int nc = dim >> 3;
    const float2* cs = (const float2*)rope_cs;
    if (nc <= NT)
        hipLaunchKernelGGL(rmsnorm_rope_kernel<1>, dim3(grid), dim3(NT), 0, st, x, ldx, out, ldo, rows, dim,
                           weight, eps, head_dim, cs, F, H, W, pos0, out_scale);
    else if (nc <= 3 * NT)
        hipLaunchKernelGGL(rmsnorm_rope_kernel<3>, dim3(grid), dim3(NT), 0, st, x, ldx, out, ldo, rows, dim,
                           weight, eps, head_dim, cs, F, H, W, pos0, out_scale);
    else
        hipLaunchKernelGGL(rmsnorm_rope_kernel<4>, dim3(grid), dim3(NT), 0, st, x, ldx, out, ldo, rows, dim,
                           weight, eps, head_dim, cs, F, H, W, pos0, out_scale);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// sinusoidal_embedding_1d  — model.py:15-25 (fp64 evaluation, fp32 result)
// ---------------------------------------------------------------------------------------------
__global__ void sinusoid_kernel(const void* t, int t_dtype, int n, int dim, float* out) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int i = idx / half, j = idx - i * half;
    double pos;
    if (t_dtype == 0) pos = (double)((const int64_t*)t)[i];
    else if (t_dtype == 1) pos = (double)((const float*)t)[i];
    else pos = ((const double*)t)[i];
    const double w = pow(10000.0, -((double)j / (double)half));
    const double a = pos * w;
    out[(int64_t)i * dim + j] = (float)cos(a);
    out[(int64_t)i * dim + half + j] = (float)sin(a);
}

extern "C" int mg_sinusoid_embed(const void* t, int t_dtype, int n, int dim, float* out, void* stream) {
    if (!t || !out) return MG_ERR_ARG;
    if (n <= 0 || dim <= 0 || (dim & 1) || t_dtype < 0 || t_dtype > 2) return MG_ERR_SHAPE;
    const int total = n * (dim >> 1);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, t,
                       t_dtype, n, dim, out);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMV (time_embedding / time_projection)  — model.py:455-457, 541-545
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemv_f32_kernel(const float* __restrict__ Wm,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ x, float* __restrict__ y,
                                                      int N, int K, int silu_in) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * (NT / 64) + wave;
    if (n >= N) return;
    const float4* wr = (const float4*)(Wm + (int64_t)n * K);
    const float4* xv = (const float4*)x;
    float acc = 0.f;
    for (int c = lane; c < (K >> 2); c += 64) {
        const float4 w4 = wr[c];
        float4 x4 = xv[c];
        if (silu_in) { x4.x = silu(x4.x); x4.y = silu(x4.y); x4.z = silu(x4.z); x4.w = silu(x4.w); }
        acc += (w4.x * x4.x + w4.y * x4.y) + (w4.z * x4.z + w4.w * x4.w);
    }
    acc = wave_sum(acc);
    if (lane == 0) y[n] = acc + (bias ? bias[n] : 0.f);
}

extern "C" int mg_gemv_f32(const float* Wm, const float* bias, const float* x, float* y, int N, int K,
                           int silu_in, void* stream) {
    if (!Wm || !x || !y) return MG_ERR_ARG;
    if (N <= 0 || K <= 0 || (K & 3)) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(NT), 0, (hipStream_t)stream, Wm, bias, x, y,
                       N, K, silu_in);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// out[r] = a[r] + b[r % period]   (modulation + e0)  — model.py:292-295, :340
// ---------------------------------------------------------------------------------------------
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                float* __restrict__ out, int rows, int dim, int period) {
    const int64_t total = (int64_t)rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / dim), c = (int)(i - (int64_t)r * dim);
        out[i] = a[i] + b[(int64_t)(r % period) * dim + c];
    }
}

extern "C" int mg_add_rows_f32(const float* a, const float* b, float* out, int rows, int dim, int period,
                               void* stream) {
    if (!a || !b || !out) return MG_ERR_ARG;
    if (rows <= 0 || dim <= 0 || period <= 0) return MG_ERR_SHAPE;
    const int64_t total = (int64_t)rows * dim;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, rows, dim,
                       period);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Head Linear in fp32: out[M][N<=64] = x[M][K] . W[N][K]^T + bias  — model.py:342
// 64-row x 64-col output tile per workgroup, 4x4 micro-tile per thread, K staged 32 at a time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void head_gemm_f32_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ Wm,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t M, int N, int K) {
    __shared__ float xs[32][64 + 4];  // [k][m]
    __shared__ float ws[32][64 + 4];  // [k][n]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = threadIdx.x + i * NT;  // 2048 = 64 rows x 32 k
            const int r = id >> 5, kk = id & 31;
            const int64_t m = m0 + r;
            xs[kk][r] = (m < M && k0 + kk < K) ? x[m * ldx + k0 + kk] : 0.f;
            ws[kk][r] = (r < N && k0 + kk < K) ? Wm[(int64_t)r * K + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float4 a = *(const float4*)&xs[kk][ty * 4];
            const float4 b = *(const float4*)&ws[kk][tx * 4];
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tx * 4 + j;
            if (n < N) out[m * N + n] = acc[i][j] + (bias ? bias[n] : 0.f);
        }
    }
}

extern "C" int mg_head_gemm_f32(const float* x, int64_t ldx, const float* Wm, const float* bias, float* out,
                                int64_t M, int N, int K, void* stream) {
    if (!x || !Wm || !out) return MG_ERR_ARG;
    if (N <= 0 || N > 64 || K <= 0 || M < 0) return MG_ERR_SHAPE;
    if (M == 0) return MG_OK;
    hipLaunchKernelGGL(head_gemm_f32_kernel, dim3((unsigned)((M + 63) / 64)), dim3(NT), 0,
                       (hipStream_t)stream, x, ldx, Wm, bias, out, M, N, K);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// patchify (im2col of the k=s=(1,ph,pw) Conv3d) and unpatchify — model.py:445-450,529-531; :581-609
// ---------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ lat, int C, int F, int H, int W, int ph, int pw,
                                uint16_t* __restrict__ out, int64_t ldo) {
    const int Hg = H / ph, Wg = W / pw, Kd = C * ph * pw;
    const int64_t total = (int64_t)F * Hg * Wg * Kd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tok = i / Kd;
        const int k = (int)(i - tok * Kd);
        const int c = k / (ph * pw), ij = k - c * ph * pw, ii = ij / pw, jj = ij - ii * pw;
        const int f = (int)(tok / ((int64_t)Hg * Wg));
        const int rem = (int)(tok - (int64_t)f * Hg * Wg);
        const int hg = rem / Wg, wg = rem - hg * Wg;
        const float v = lat[(((int64_t)c * F + f) * H + hg * ph + ii) * W + wg * pw + jj];
        out[tok * ldo + k] = f2bf(v);
    }
}

extern "C" int mg_patchify_bf16(const float* lat, int C, int F, int H, int W, int ph, int pw, uint16_t* out,
                                int64_t ldo, void* stream) {
    if (!lat || !out) return MG_ERR_ARG;
    if (C <= 0 || F <= 0 || ph <= 0 || pw <= 0 || H % ph || W % pw) return MG_ERR_SHAPE;
    const int64_t total = (int64_t)F * (H / ph) * (W / pw) * C * ph * pw;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, lat, C, F, H, W, ph, pw,
                       out, ldo);
    return mg_check_launch();
}

__global__ void unpatchify_kernel(const float* __restrict__ tok, int64_t ldt, int C, int F, int Hg, int Wg,
                                  int ph, int pw, float* __restrict__ lat) {
    const int H = Hg * ph, W = Wg * pw;
    const int64_t total = (int64_t)C * F * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        int64_t r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int f = (int)(r % F);
        const int c = (int)(r / F);
        const int hg = h / ph, ii = h - hg * ph, wg = w / pw, jj = w - wg * pw;
        const int64_t t = ((int64_t)f * Hg + hg) * Wg + wg;
        lat[i] = tok[t * ldt + (ii * pw + jj) * C + c];
    }
}

extern "C" int mg_unpatchify_f32(const float* tok, int64_t ldt, int C, int F, int Hg, int Wg, int ph, int pw,
                                 float* lat, void* stream) {
    if (!tok || !lat) return MG_ERR_ARG;
    if (C <= 0 || F <= 0 || Hg <= 0 || Wg <= 0 || ph <= 0 || pw <= 0) return MG_ERR_SHAPE;
    const int64_t total = (int64_t)C * F * Hg * ph * Wg * pw;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tok, ldt, C, F, Hg,
                       Wg, ph, pw, lat);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// out = c0*x0 + c1*x1 + c2*x2 + c3*x3  — CFG (text2video.py:245-246) and scheduler updates
// ---------------------------------------------------------------------------------------------
__global__ void lincomb4_kernel(float* __restrict__ out, int64_t n, const float* x0, float c0,
                                const float* x1, float c1, const float* x2, float c2, const float* x3,
                                float c3) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        if (x0) a = c0 * x0[i];
        if (x1) a += c1 * x1[i];
        if (x2) a += c2 * x2[i];
        if (x3) a += c3 * x3[i];
        out[i] = a;
    }
}

extern "C" int mg_lincomb4_f32(float* out, int64_t n, const float* x0, float c0, const float* x1, float c1,
                               const float* x2, float c2, const float* x3, float c3, void* stream) {
    if (!out) return MG_ERR_ARG;
    if (n < 0) return MG_ERR_SHAPE;
    if (n == 0) return MG_OK;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(lincomb4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, x0, c0, x1, c1,
                       x2, c2, x3, c3);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// x[r][c] += y[r][c] * gate[c]   (x fp32 residual stream, y bf16, gate fp32 or NULL = 1): the gated residual of
// model.py:301-302,306,308-309 as its own kernel — used when the producer of y is not one of this library's GEMMs
// (a caller-replaced WanSelfAttention.forward, the operator seam of text2video.py:97-100); the GEMM epilogue
// MG_EPI_GATE_RESID_F32 is the fused form of the same arithmetic.
// ---------------------------------------------------------------------------------------------
__global__ void gate_residual_kernel(float* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ y, int64_t ldy,
                                     const float* __restrict__ gate, int64_t rows, int vec_per_row) {
    const int64_t total = rows * vec_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        const int64_t r = i / vec_per_row;
        const int c = (int)(i - r * vec_per_row) * 4;
        const u16x4_t yv = *reinterpret_cast<const u16x4_t*>(y + r * ldy + c);
        f32x4_t xv = *reinterpret_cast<f32x4_t*>(x + r * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)   // product rounded, then added (torch: y * e, then x + .) — no fma contraction
            xv[e] = xv[e] + bf2f(yv[e]) * (gate ? gate[c + e] : 1.f);
        *reinterpret_cast<f32x4_t*>(x + r * ldx + c) = xv;
    }
}

extern "C" int mg_gate_residual_f32(float* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate, int64_t rows,
                                    int dim, void* stream) {
    if (!x || !y) return MG_ERR_ARG;
    if (rows < 0 || dim <= 0 || (dim & 3) || (ldx & 3) || (ldy & 3)) return MG_ERR_SHAPE;
    if (rows == 0) return MG_OK;
    int64_t g = (rows * (dim / 4) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(gate_residual_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, gate, rows,
                       dim / 4);
    return mg_check_launch();
}

extern "C" const char* mg_version(void) { return "moviigen_hip 3 gfx950"; }
extern "C" int mg_abi_version(void) { return 9; }   // 9: mg_attn_fwd_bf16_hd128 / _lse / _prescaled take a caller-owned workspace (mg_attn_workspace_bytes(); the library no longer allocates or synchronises on a launch path), + the W-band VAE entries mg_vae_conv_cols_f32 / mg_vae_upconv_phases_cols_f32 / mg_vae_attn_rows_f32; 8: the product library exports no kernel-selection switch and no profiling hook (they live in libmoviigen_hip_ab.so, -DMG_AB_BUILD; mg_*_set_variant return a status there); 7: the VAE arithmetic mode is an argument of mg_vae_conv_f32 / mg_vae_upconv_phases_f32 (mg_vae_set_mode is gone), mg_attn_w64_flag_counter counts into two words, mg_attn_fwd_bf16_hd128_prescaled gained reserve_cus; 6: + mg_vae_set_mode; 5: + mg_vae_upconv_fold_weights_f32 / mg_vae_upconv_phases_f32; 4: mg_rmsnorm_rope_bf16 gained out_scale, mg_pack_kv_bf16's K row order follows the 16x16x32 attention kernel, + mg_attn_fwd_bf16_hd128_prescaled
