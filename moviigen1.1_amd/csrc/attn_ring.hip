// Ring attention — merge of partial attention results (the reference delegates
// to yunchang's ring flash-attention inside xFuserLongContextAttention, generate.py:225-229).
// A rank attends its queries to one K/V block per ring step (mg_attn_fwd_bf16_hd128_lse: normalised
// bf16 output + log-sum-exp per row); this kernel folds block j into the running fp32 result:
//     lse' = logaddexp(lse, lse_j),  acc' = acc * exp(lse - lse') + out_j * exp(lse_j - lse')
// and writes the bf16 result on the last step.  Row-local, HBM-bound (6 B read + 4 B written per element).
#include "common.h"
#include "../../include/moviigen_hip.h"

__global__ __launch_bounds__(256) void attn_merge_kernel(float* __restrict__ acc, int64_t lda, float* __restrict__ lse_acc,
                                                         const uint16_t* __restrict__ part, int64_t ldp,
                                                         const float* __restrict__ lse_part, uint16_t* __restrict__ out,
                                                         int64_t ldo, int64_t Lq, int heads, int first) {
    // one wave per (row, head): 64 lanes x 2 elements = 128
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= Lq * heads) return;
    const int64_t row = item / heads;
    const int head = (int)(item - row * heads), lane = threadIdx.x & 63;
    const int64_t li = (int64_t)head * Lq + row;
    const float lp = lse_part[li];
    float wa = 0.f, wp = 1.f, ln = lp;
    if (!first) {
        const float la = lse_acc[li];
        const float mx = fmaxf(la, lp);
        ln = mx + __logf(__expf(la - mx) + __expf(lp - mx));
        wa = __expf(la - ln);
        wp = __expf(lp - ln);
    }
    const int col = head * 128 + lane * 2;
    const unsigned pv = *(const unsigned*)(part + row * ldp + col);
    float2 a = first ? make_float2(0.f, 0.f) : *(const float2*)(acc + row * lda + col);
    a.x = a.x * wa + bf2f((uint16_t)(pv & 0xffff)) * wp;
    a.y = a.y * wa + bf2f((uint16_t)(pv >> 16)) * wp;
    *(float2*)(acc + row * lda + col) = a;
    if (lane == 0) lse_acc[li] = ln;
    if (out) *(unsigned*)(out + row * ldo + col) = pack_bf2(a.x, a.y);
}

extern "C" int mg_attn_merge_f32(float* acc, int64_t lda, float* lse_acc, const uint16_t* part, int64_t ldp,
                                 const float* lse_part, uint16_t* out, int64_t ldo, int64_t Lq, int heads, int first,
                                 void* stream) {
    if (!acc || !lse_acc || !part || !lse_part) return MG_ERR_ARG;
    if (Lq < 0 || heads <= 0 || (lda & 1) || (ldp & 1) || (out && (ldo & 1))) return MG_ERR_SHAPE;
    if (Lq == 0) return MG_OK;
    const int64_t items = Lq * heads;
    if ((items + 3) / 4 > 0x7fffffffLL) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, acc, lda,
                       lse_acc, part, ldp, lse_part, out, ldo, Lq, heads, first);
    return mg_check_launch();
}
