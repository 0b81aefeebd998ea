// Generic-head-size attention (any head_dim <= 256, % 8 == 0): correctness path for model sizes
// whose head_dim is not 128 (e.g. BASELINE.json configs[0], head_dim 32).  Same contract as the
// MFMA kernel — wan/modules/attention.py:24-130 — but V is read un-transposed.
// One wave per (query, head); keys walked 64 at a time (one key per lane), online softmax.
#include "common.h"
#include "../../include/moviigen_hip.h"

__global__ __launch_bounds__(64) void attn_generic_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                          const uint16_t* __restrict__ k, int64_t ldk,
                                                          const uint16_t* __restrict__ v, int64_t ldv,
                                                          uint16_t* __restrict__ o, int64_t ldo, int64_t Lq,
                                                          int64_t Lk, int head_dim, float scale) {
    __shared__ float qs[256];
    __shared__ float ps[64];
    const int64_t qi = blockIdx.x;
    const int head = blockIdx.y;
    const int lane = threadIdx.x;
    for (int d = lane; d < head_dim; d += 64) qs[d] = bf2f(q[qi * ldq + head * head_dim + d]);
    __syncthreads();
    float m = -1e30f, l = 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d = lane + 64*i
    for (int64_t kv0 = 0; kv0 < Lk; kv0 += 64) {
        const int64_t key = kv0 + lane;
        float s = -1e30f;
        if (key < Lk) {
            const uint16_t* kr = k + key * ldk + head * head_dim;
            float a = 0.f;
            for (int d = 0; d < head_dim; ++d) a += qs[d] * bf2f(kr[d]);
            s = a * scale;
        }
        const float mx = fmaxf(m, wave_max(s));
        const float alpha = __expf(m - mx);
        const float p = (key < Lk) ? __expf(s - mx) : 0.f;
        l = l * alpha + wave_sum(p);
        m = mx;
        __syncthreads();
        ps[lane] = round_bf(p);  // P is bf16 on the MFMA path as well
        __syncthreads();
        const int nk = (int)((Lk - kv0) < 64 ? (Lk - kv0) : 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = lane + 64 * i;
            if (d < head_dim) {
                float a = acc[i] * alpha;
                for (int j = 0; j < nk; ++j) a += ps[j] * bf2f(v[(kv0 + j) * ldv + head * head_dim + d]);
                acc[i] = a;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < head_dim) o[qi * ldo + head * head_dim + d] = f2bf(acc[i] / l);
    }
}

extern "C" int mg_attn_fwd_bf16_generic(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                                        const uint16_t* v, int64_t ldv, uint16_t* o, int64_t ldo, int64_t Lq,
                                        int64_t Lk, int heads, int head_dim, float scale, void* stream) {
    if (!q || !k || !v || !o) return MG_ERR_ARG;
    if (Lq < 0 || Lk <= 0 || heads <= 0 || head_dim <= 0 || head_dim > 256 || Lq > 0x7fffffffLL ||
        heads > 65535)
        return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    hipLaunchKernelGGL(attn_generic_kernel, dim3((unsigned)Lq, heads), dim3(64), 0, (hipStream_t)stream, q, ldq,
                       k, ldk, v, ldv, o, ldo, Lq, Lk, head_dim, scale);
    return mg_check_launch();
}

// classifier-free guidance: out = u + g * (c - u)   (wan/text2video.py:245-246, same op order)
__global__ void cfg_combine_kernel(float* __restrict__ out, const float* __restrict__ u,
                                   const float* __restrict__ c, float g, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = u[i] + g * (c[i] - u[i]);
}

extern "C" int mg_cfg_combine_f32(float* out, const float* uncond, const float* cond, float guide_scale,
                                  int64_t n, void* stream) {
    if (!out || !uncond || !cond) return MG_ERR_ARG;
    if (n < 0) return MG_ERR_SHAPE;
    if (n == 0) return MG_OK;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, uncond, cond,
                       guide_scale, n);
    return mg_check_launch();
}
