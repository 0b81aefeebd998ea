// Shared device helpers for the MoviiGen1.1 MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes; bf16 travels as uint16_t; all math in fp32 unless stated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MG_OK 0
#define MG_ERR_ARG (-1)
#define MG_ERR_SHAPE (-2)
#define MG_ERR_LAUNCH (-3)
#define MG_ERR_UNAVAILABLE (-4)   // librccl could not be bound at run time
#define MG_ERR_COMM (-5)          // an RCCL call failed

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define MG_DEV static __device__ __forceinline__

// fp32 -> bf16 round-to-nearest-even (matches torch .to(bfloat16)); one v_cvt_pk_bf16_f32 per pair
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
MG_DEV unsigned int pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
MG_DEV unsigned short f2bf(float f) { return (unsigned short)(pack_bf2(f, 0.f) & 0xffffu); }
MG_DEV float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
MG_DEV float round_bf(float f) { return __uint_as_float(pack_bf2(f, 0.f) << 16); }

MG_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
MG_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); `red` is NT/64 floats of LDS.
template <int NT>
MG_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// torch.nn.functional.gelu(x, approximate='tanh') = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).
// With tanh(u) = 1 - 2 / (e^{2u} + 1):  0.5 x (1 + tanh(u)) = x / (1 + e^{-2u}) — ONE v_exp_f32 and ONE v_rcp_f32 per value (7 VALU instructions; the
// round-1 form went through an IEEE division: ~22, and the GELU epilogue of ffn.0 is VALU-bound behind the tile's last MFMA: 0.72 VALU instructions per
// MFMA over the launch, profiles/r06_pmc_gemm_v12.txt).  -2 log2(e) is folded into the polynomial's constants; x -> +inf: e -> 0, x; x -> -inf: e -> inf, -0.
MG_DEV float gelu_tanh(float x) {
    const float a0 = -2.f * 1.4426950408889634f * 0.7978845608028654f, a1 = a0 * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * (a0 + a1 * (x * x)));      // e^{-2u}
    return x * __builtin_amdgcn_rcpf(1.f + e);
}
MG_DEV float silu(float x) { return x / (1.f + __expf(-x)); }

// CUs of the CURRENT device, asked per call (an attribute lookup; a function-static cache would be wrong in a process
// that drives devices of different sizes and racy on first use from two threads); -1 on error
static inline int mg_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return -1;
    return n;
}

static inline int mg_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MG_OK : MG_ERR_LAUNCH;
}
