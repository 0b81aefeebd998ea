// umT5-XXL encoder pieces (reference wan/modules/t5.py) that the DiT kernels do not already cover.
// The encoder runs twice per video on <= 512 tokens (prompt + negative prompt): ~5 TFLOP each, all
// of it in mg_gemm_bf16; these are the glue kernels.  The whole model is bf16 in the reference
// (`model.to(dtype=bf16)`, t5.py:456, config t5_dtype): activations and the residual stream are
// bf16 tensors, fp32 inside each op.
#include "common.h"
#include "../../include/moviigen_hip.h"

// token_embedding lookup (t5.py:273,289): out[i][:] = table[ids[i]][:]
__global__ void t5_embed_kernel(const uint16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                uint16_t* __restrict__ out, int n, int dim, int64_t vocab) {
    const int row = blockIdx.x;
    if (row >= n) return;
    int64_t id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    const u16x8_t* src = (const u16x8_t*)(table + id * dim);
    u16x8_t* dst = (u16x8_t*)(out + (int64_t)row * dim);
    for (int c = threadIdx.x; c < (dim >> 3); c += blockDim.x) dst[c] = src[c];
}

extern "C" int mg_embed_rows_bf16(const uint16_t* table, int64_t vocab, int dim, const int64_t* ids, int n,
                                  uint16_t* out, void* stream) {
    if (!table || !ids || !out) return MG_ERR_ARG;
    if (n < 0 || dim <= 0 || (dim & 7) || vocab <= 0) return MG_ERR_SHAPE;
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(t5_embed_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, table, ids, out, n, dim, vocab);
    return mg_check_launch();
}

// the reference's GELU module is a chain of bf16 elementwise ops (t5.py:46-50), each of which rounds its result:
// pow(x,3), *0.044715, +x, *sqrt(2/pi), tanh, 1+, 0.5*x, * — restated with the same rounding points
MG_DEV float t5_gelu_bf16_chain(float x) {
    const float p = round_bf(x * x * x);
    const float m = round_bf(0.044715f * p);
    const float s = round_bf(x + m);
    const float u = round_bf(0.7978845608028654f * s);
    const float e = __expf(2.f * u);
    const float th = round_bf(1.f - 2.f / (e + 1.f));
    const float o = round_bf(1.f + th);
    return round_bf((0.5f * x) * o);
}

// mode 0: out = bf16(a + b)                                  residual add of bf16 tensors (t5.py:165-166)
// mode 1: out = bf16(a * gelu(b))                            T5FeedForward: fc1(x) * gate(x) (t5.py:135)
__global__ void t5_ew_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                             uint16_t* __restrict__ out, int64_t n8, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u16x8_t ua = ((const u16x8_t*)a)[i], ub = ((const u16x8_t*)b)[i];
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = bf2f(ua[j]), y = bf2f(ub[j]);
            r[j] = mode == 0 ? x + y : x * t5_gelu_bf16_chain(y);
        }
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf2(r[2 * j], r[2 * j + 1]);
        ((u32x4_t*)out)[i] = o;
    }
}

extern "C" int mg_ew_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, int mode, void* stream) {
    if (!a || !b || !out) return MG_ERR_ARG;
    if (n < 0 || (n & 7) || mode < 0 || mode > 1) return MG_ERR_SHAPE;
    if (n == 0) return MG_OK;
    int64_t g = (n / 8 + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(t5_ew_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 8, mode);
    return mg_check_launch();
}

// T5Attention (t5.py:82-113): softmax(q k^T + pos_bias) v — NO 1/sqrt(d) scaling, additive bias
// emb[bucket(j - i)][head]; the bucket of every relative position (the bidirectional log-bucket
// rule of T5RelativeEmbedding, t5.py:242-263) is tabulated on the host with the reference's own
// fp32 torch expression: rel_bucket[(j - i) + Lq - 1].  Keys >= Lk masked.
// One wave per (query, head); head_dim <= 128.
__global__ __launch_bounds__(64) void t5_attn_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k,
                                                     const uint16_t* __restrict__ v, int64_t ld,
                                                     const uint16_t* __restrict__ emb,
                                                     const int* __restrict__ rel_bucket, uint16_t* __restrict__ o, int64_t ldo, int64_t Lq, int64_t Lk,
                                                     int heads, int head_dim) {
    __shared__ float qs[128];
    __shared__ float ps[64];
    const int64_t qi = blockIdx.x;
    const int head = blockIdx.y;
    const int lane = threadIdx.x;
    for (int d = lane; d < head_dim; d += 64) qs[d] = bf2f(q[qi * ld + head * head_dim + d]);
    __syncthreads();
    float m = -3.0e38f, l = 0.f;
    float acc[2] = {0.f, 0.f};
    for (int64_t kv0 = 0; kv0 < Lk; kv0 += 64) {
        const int64_t key = kv0 + lane;
        float s = -3.0e38f;
        if (key < Lk) {
            const uint16_t* kr = k + key * ld + head * head_dim;
            float a = 0.f;
            for (int d = 0; d < head_dim; ++d) a += qs[d] * bf2f(kr[d]);
            // scores are a bf16 tensor in the reference (einsum of bf16 operands) + bf16 bias
            const float bias = bf2f(emb[rel_bucket[key - qi + Lq - 1] * heads + head]);
            s = round_bf(round_bf(a) + bias);
        }
        const float mx = fmaxf(m, wave_max(s));
        const float alpha = __expf(m - mx);
        const float p = (key < Lk) ? __expf(s - mx) : 0.f;
        l = l * alpha + wave_sum(p);
        m = mx;
        __syncthreads();
        ps[lane] = p;
        __syncthreads();
        const int nk = (int)((Lk - kv0) < 64 ? (Lk - kv0) : 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d = lane + 64 * i;
            if (d < head_dim) {
                float a = acc[i] * alpha;
                for (int j = 0; j < nk; ++j) a += ps[j] * bf2f(v[(kv0 + j) * ld + head * head_dim + d]);
                acc[i] = a;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = lane + 64 * i;
        if (d < head_dim) o[qi * ldo + head * head_dim + d] = f2bf(acc[i] / l);
    }
}

extern "C" int mg_t5_attn_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld,
                               const uint16_t* rel_emb, const int* rel_bucket, uint16_t* o, int64_t ldo,
                               int64_t Lq, int64_t Lk, int heads, int head_dim, void* stream) {
    if (!q || !k || !v || !rel_emb || !rel_bucket || !o) return MG_ERR_ARG;
    if (Lq < 0 || Lk <= 0 || heads <= 0 || heads > 65535 || head_dim <= 0 || head_dim > 128 || Lq > 0x7fffffffLL)
        return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    hipLaunchKernelGGL(t5_attn_kernel, dim3((unsigned)Lq, heads), dim3(64), 0, (hipStream_t)stream, q, k, v, ld,
                       rel_emb, rel_bucket, o, ldo, Lq, Lk, heads, head_dim);
    return mg_check_launch();
}
