// bf16 GEMM  out[M][N] = A[M][K] . W[N][K]^T  with fused epilogues, for gfx950 (CDNA4).
//
// Replaces every nn.Linear that the reference runs under autocast(bf16) on the DiT path
// (wan/modules/model.py:139-141,155,168-170,180,267-269,451-453 and the k=s patch Conv3d
// :445-450), with the residual/gate updates of model.py:301-302,306,308-309 fused.
//
// Design (MI355X-first, not a cuBLAS call pattern):
//  * 128(token) x 128(feature) x 64(k) tile per 256-thread workgroup, 4 waves as 2x2, each wave a
//    64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 fp32 regs/lane).
//  * both operands are K-contiguous in HBM, so both tiles are staged with 16-byte LDS-DMA
//    (global_load_lds_dwordx4): no VGPR round trip.  The LDS image is lane-linear per wave
//    instruction (8 rows x 128 B); the bank-conflict swizzle is applied to the per-lane SOURCE
//    address and undone in the ds_read_b128 address (16-B chunk c of row r lives at chunk
//    c ^ ((r>>1)&7)), which makes every ds_read_b128 lane group hit 16 distinct 16-B slots.
//  * double-buffered LDS (2 x 32 KiB): tile t+1 streams in while tile t is multiplied;
//    one barrier per k-tile.  64 KiB/workgroup -> 2 workgroups per CU (8 waves).
//  * MFMA operand roles are swapped (A-operand = W rows, B-operand = token rows) so each lane
//    owns ONE token row and 4 consecutive output features per accumulator quad: the epilogue
//    (bias, GELU, gate, fp32 residual read-modify-write) is row-local with 8/16-byte accesses.
//  * workgroup ids are remapped so each XCD (private 4 MiB L2) walks a contiguous, 8-tile-tall
//    band of the output: the concurrently resident tiles of one XCD share A and W panels.
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define TILE_BYTES (128 * BK * 2)  // 16 KiB per operand tile

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

MG_DEV void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(
    const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, int64_t M, int N, int K, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ gate, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][A|W]

    // ---- workgroup -> tile: XCD-contiguous remap, then 8-tall grouped raster --------------
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = swz / per_group;
    const int first_m = group * GM;
    const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_g = swz - group * per_group;
    const int tm = first_m + in_g % gsz;
    const int tn = in_g / gsz;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int l31 = lane & 31, g = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- LDS-DMA source addresses: wave w stages rows [32w, 32w+32) of both tiles ----------
    const int srow = lane >> 3;                        // row within the 8-row wave instruction
    const uint16_t* ga[4];
    const uint16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        const int cg = (lane & 7) ^ ((row >> 1) & 7);  // logical chunk stored at position lane&7
        int64_t am = m0 + row;
        if (am > M - 1) am = M - 1;
        int wr = n0 + row;
        if (wr > N - 1) wr = N - 1;
        ga[i] = A + am * lda + cg * 8;
        gw[i] = Wt + (int64_t)wr * ldw + cg * 8;
    }
    auto stage = [&](int kt, int buf) {
        char* la = smem + buf * 2 * TILE_BYTES + wave * 32 * 128;
        char* lw = la + TILE_BYTES;
        const int koff = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(ga[i] + koff, la + i * 8 * 128);
            glds16(gw[i] + koff, lw + i * 8 * 128);
        }
    };

    // ---- fragment read offsets ---------------------------------------------------------
    const int sw = (l31 >> 1) & 7;
    const int t3 = g ^ sw;  // chunk = t3 ^ (kk<<1)
    const int a_row_off = (wm * 64 + l31) * 128;  // token rows (MFMA B operand)
    const int w_row_off = (wn * 64 + l31) * 128;  // feature rows (MFMA A operand)

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* la = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* lw = la + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int coff = (t3 ^ (kk << 1)) << 4;
            bf16x8_t fa[2], fw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fa[j] = *(const bf16x8_t*)(la + a_row_off + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fw[i] = *(const bf16x8_t*)(lw + w_row_off + i * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue (gemm_epilogue.h): lane owns token row m, 4 features per accumulator quad; batched loads ----
    mg_gemm_epilogue<EPI, 2, 2>(acc, m0 + wm * 64, n0 + wn * 64, l31, g, M, N, bias, gate, out, ldo);
}

int mg_gemm_v12_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                       int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);       // gemm_bf16_v12.hip
int mg_gemm_v2_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);

// Tile schedules in the PRODUCT library (libmoviigen_hip.so), chosen by shape alone — no switch, no process-global state:
//   12 = 256x256x64 tile, 16x16x32 MFMA, one wave per SIMD, persistent, loads two k-tiles ahead (gemm_bf16_v12.hip): M > 256 and N > 128;
//    2 = 256x128x64, 8 waves, 3 stages (gemm_bf16_v2.hip): everything narrower, and what variant 12's launcher hands back (one k-tile,
//        bf16 pitches that only allow 8-byte stores, strides past its 32-bit tile offsets);   1 = 128x128x64, 2 stages (this file): M <= 128.
// The A/B library (libmoviigen_hip_ab.so, built with -DMG_AB_BUILD: mg_selftest, the variant-agreement tests, tools/) adds the earlier
// large-shape kernels — 11 (one barrier per k-tile, generated schedule), 8 (eight waves in two ping-pong groups), 7 (one wave per SIMD,
// compiler-scheduled) — behind mg_gemm_set_variant, a process-global MEASUREMENT switch, and the s_memtime hooks.  Archived under
// experiments/: 3, 4, 5, 6, 9, 10.
#ifdef MG_AB_BUILD
int mg_gemm_v7_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);
int mg_gemm_v8_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                      int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);
int mg_gemm_v11_launch(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw, const float* bias, int64_t M,
                       int N, int K, int epilogue, void* out, int64_t ldo, const float* gate, hipStream_t st);
unsigned long long* g_gemm5_prof = nullptr;   // s_memtime hook of the 256x256 kernels (7, 8, 11, 12)
extern "C" void mg_gemm5_debug_profile(unsigned long long* dev_buf) { g_gemm5_prof = dev_buf; }
static int g_gemm_variant = 0;   // 0 = the product's rule
void mg_gemm_v11_set_flags(int f);
void mg_gemm_v12_set_flags(int f);
extern "C" int mg_gemm_set_variant(int v) {      // 110 + f / 200 + f: variant 11 / 12 with measurement flags f (gemm_bf16_v11.hip, gemm_bf16_v12.hip)
    const int base = v >= 200 ? 12 : v >= 110 ? 11 : v;
    if (base != 0 && base != 1 && base != 2 && base != 7 && base != 8 && base != 11 && base != 12) return MG_ERR_ARG;      // no silent aliases
    if ((v >= 110 && v - 110 >= 64 && v < 200) || v >= 200 + 8192) return MG_ERR_ARG;
    if (v >= 200 && (((v - 200) >> 5) & 3) == 2) return MG_ERR_ARG;      // variant 12: generated bodies 0 (bits 5-6 = 0 / 1) and 2 (= 3) are compiled; 264 + f named body 1, which is not
    g_gemm_variant = base;
    mg_gemm_v11_set_flags(v >= 110 && v < 200 ? v - 110 : 0);
    mg_gemm_v12_set_flags(v >= 200 ? v - 200 : 0);
    return MG_OK;
}
#endif

extern "C" int mg_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* Wt, int64_t ldw,
                            const float* bias, int64_t M, int N, int K, int epilogue, void* out,
                            int64_t ldo, const float* gate, void* stream) {
    if (!A || !Wt || !out) return MG_ERR_ARG;
    if (epilogue < 0 || epilogue > 3) return MG_ERR_ARG;
    if (M < 0 || N <= 0 || K <= 0 || (K % BK) || (lda & 7) || (ldw & 7) || (ldo & 3)) return MG_ERR_SHAPE;
    if (((uintptr_t)A & 15) || ((uintptr_t)Wt & 15) || ((uintptr_t)out & 15)) return MG_ERR_SHAPE;
    if (bias && ((uintptr_t)bias & 15)) return MG_ERR_SHAPE;
    if (gate && ((uintptr_t)gate & 15)) return MG_ERR_SHAPE;
    if (M == 0) return MG_OK;
#ifdef MG_AB_BUILD
    const int variant = g_gemm_variant ? g_gemm_variant : 12;
    if (variant == 11 && M > 256 && N > 128)
        return mg_gemm_v11_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, (hipStream_t)stream);
    if (variant == 8 && M > 256 && N > 128)
        return mg_gemm_v8_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, (hipStream_t)stream);
    if (variant == 7 && M > 256 && N > 128)
        return mg_gemm_v7_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, (hipStream_t)stream);
#else
    const int variant = 12;
#endif
    if (variant == 12 && M > 256 && N > 128)
        return mg_gemm_v12_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, (hipStream_t)stream);
    if (variant >= 2 && M > 128)  // tiny M: the 128-row tile wastes less (the 256x256 variants fall through to here for narrow shapes)
        return mg_gemm_v2_launch(A, lda, Wt, ldw, bias, M, N, K, epilogue, out, ldo, gate, (hipStream_t)stream);
    const int64_t tiles_m64 = (M + BM - 1) / BM;
    const int tiles_n = (N + BN - 1) / BN;
    if (tiles_m64 * tiles_n > 0x7fffffffLL) return MG_ERR_SHAPE;
    const int tiles_m = (int)tiles_m64;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(GEMM_THREADS);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(E)                                                                                      \
    hipLaunchKernelGGL(gemm_bf16_kernel<E>, grid, block, 0, st, A, lda, Wt, ldw, bias, M, N, K, out, ldo, \
                       gate, tiles_m, tiles_n)
    switch (epilogue) {
        case MG_EPI_BIAS_BF16: LAUNCH(MG_EPI_BIAS_BF16); break;
        case MG_EPI_BIAS_GELU_BF16: LAUNCH(MG_EPI_BIAS_GELU_BF16); break;
        case MG_EPI_GATE_RESID_F32: LAUNCH(MG_EPI_GATE_RESID_F32); break;
        default: LAUNCH(MG_EPI_BIAS_F32); break;
    }
#undef LAUNCH
    return mg_check_launch();
}
