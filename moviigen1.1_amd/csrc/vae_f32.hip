// WanVAE decode kernels, fp32-exact mode, for gfx950.
//
// The reference runs the whole VAE in fp32 (wan/modules/vae.py:623,658): 1.1 PFLOP of 3x3x3
// causal convolutions at 1920x832x81.  gfx950 has an exact-f32 MFMA (v_mfma_f32_32x32x2_f32,
// bitwise an fmaf chain) at the f32 vector peak (157 TF) — so the convolutions run as an
// IMPLICIT GEMM on that instruction: M = output voxels, N = Cout, K = taps x Cin.
//
// Layout: activations are channels-last [T][H][W][C] (the reference is NCTHW): every tap of every
// voxel is then a K-contiguous run of Cin floats, exactly like a GEMM row, and the output row of
// a voxel is N-contiguous.  Weights are repacked once to [Cout][kt][kh][kw][Cin].
//
// Tile: 128 voxels x (32*NB) couts x 32 channels per step, 4 waves, wave w owns voxels
// [32w,32w+32) x all NB cout blocks (NB x f32x16 accumulators).  MFMA A-operand = weights
// (row = cout), B-operand = voxels, so a lane owns ONE voxel and 4 consecutive couts per
// accumulator quad: bias + residual + store are 16-byte row-local accesses.
// LDS rows are 36 floats (144 B): 16-byte aligned for ds_write_b128 staging and conflict-free
// for the ds_read_b128 fragment reads (row*36 mod 64 hits 16 distinct 4-bank groups).
// Each lane reads 4 consecutive k per ds_read_b128 and feeds them to 4 successive MFMAs; A and B
// use the same k permutation so the contraction is complete.
// The causal temporal padding (2 frames of cache / zeros, vae.py:28-36), the spatial zero padding
// and the nearest-exact 2x upsample of Resample (vae.py:66-83) are all folded into the gather of
// the A tile: no padded or upsampled tensor is ever materialised.
#include "common.h"
#include "../../include/moviigen_hip.h"

#define CV_THREADS 256
#define CV_BM 128
#define CV_BK 32
#define CV_LDS 36  // floats per LDS row

struct ConvArgs {
    const float* x; const float* cache; int tc; int T, H, W, Cin; int64_t ldx;
    const float* w; int64_t ldw; const float* bias; int Cout; int kt, kh, kw; int up2;
    const float* residual; float* out; int64_t ldo; int Ho, Wo; int64_t M; float out_scale;
};

// Output epilogue shared by both conv kernels: lane (l31, g) owns voxel row m and, per cout block nb and quad rq, the four
// consecutive couts n = n0 + 32 nb + 8 rq + 4 g .. +3.  Bias and residual are LOADED IN BATCHES (all NB*4 float4 of a voxel
// row before any is used): one load at a time, hipcc puts an `s_waitcnt vmcnt(0)` behind every load — NB*4 serialized
// HBM round trips per tile with the matrix pipe idle.
template <int NB, int GRP>      // GRP = cout blocks whose loads are in flight together (register budget of the caller)
MG_DEV void cv_epilogue(const ConvArgs& a, const f32x16_t (&acc)[NB], int64_t m, int n0, int g) {
    if (m >= a.M) return;
    const bool fullw = n0 + NB * 32 <= a.Cout;           // whole tile inside Cout: vector path without per-quad checks
#pragma unroll
    for (int nb0 = 0; nb0 < NB; nb0 += GRP) {
        constexpr int NC = GRP * 4;
        float4 b4[NC], r4[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = n0 + (nb0 + (c >> 2)) * 32 + (c & 3) * 8 + g * 4;
            b4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            r4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nb0 + (c >> 2) >= NB) continue;
            if (fullw || n + 3 < a.Cout) {
                if (a.bias) b4[c] = *(const float4*)(a.bias + n);
                if (a.residual) r4[c] = *(const float4*)(a.residual + m * a.ldo + n);
            } else {
                float* bb = (float*)&b4[c];
                float* rp = (float*)&r4[c];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a.Cout) {
                        if (a.bias) bb[e] = a.bias[n + e];
                        if (a.residual) rp[e] = a.residual[m * a.ldo + n + e];
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int nb = nb0 + (c >> 2), rq = c & 3;
            if (nb >= NB) continue;
            const int n = n0 + nb * 32 + rq * 8 + g * 4;
            if (n >= a.Cout) continue;
            // same operation order as before: (acc * scale + bias) + residual
            const float4 v = make_float4((acc[nb][rq * 4 + 0] * a.out_scale + b4[c].x) + r4[c].x, (acc[nb][rq * 4 + 1] * a.out_scale + b4[c].y) + r4[c].y,
                                         (acc[nb][rq * 4 + 2] * a.out_scale + b4[c].z) + r4[c].z, (acc[nb][rq * 4 + 3] * a.out_scale + b4[c].w) + r4[c].w);
            if (fullw || n + 3 < a.Cout) *(float4*)(a.out + m * a.ldo + n) = v;
            else {
                const float* vv = (const float*)&v;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < a.Cout) a.out[m * a.ldo + n + e] = vv[e];
            }
        }
    }
}

template <int NB>
__global__ __launch_bounds__(CV_THREADS) void vae_conv_kernel(const ConvArgs a) {
    constexpr int BN = 32 * NB;
    __shared__ __attribute__((aligned(16))) float smem[2 * (CV_BM + BN) * CV_LDS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * CV_BM;
    const int n0 = blockIdx.y * BN;

    // ---- gather bookkeeping: this thread stages rows (tid>>3)+32i, float4 column tid&7 -----------
    const int ch4 = tid & 7;
    int vt_[4], vy_[4], vx_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + (tid >> 3) + 32 * i;
        if (m < a.M) {
            const int64_t hw = (int64_t)a.Ho * a.Wo;
            vt_[i] = (int)(m / hw);
            const int rem = (int)(m - (int64_t)vt_[i] * hw);
            vy_[i] = rem / a.Wo;
            vx_[i] = rem - vy_[i] * a.Wo;
        } else {
            vt_[i] = -1000000; vy_[i] = 0; vx_[i] = 0;
        }
    }
    const int ncc = (a.Cin + CV_BK - 1) / CV_BK;      // channel chunks per tap
    const int ntap = a.kt * a.kh * a.kw;
    const int nchunk = ntap * ncc;

    // The gather is BRANCH-FREE: every lane always loads from a clamped, valid address and the result is zeroed by a
    // select when the tap falls into the zero padding / before the first cached frame / past M (r02: the guarded
    // version compiled to ~16 nested execz branches per chunk, which pinned all address arithmetic and loads in
    // front of the MFMA block of the iteration — the matrix pipe idled 34 % of the time at full clock).  The chunk
    // coordinates (tap -> dt, dy, dx; channel offset) advance by counters instead of divisions.
    float4 ra[4], rw[NB];
    float ka[4], kw_[NB];                                            // 0/1 masks of the loads in flight
    int ld_cc = 0, ld_dt = 0, ld_dy = 0, ld_dx = 0, ld_tap = 0;     // coordinates of the NEXT chunk to load (wave-uniform)
    const float* const base_neg = a.cache ? a.cache : a.x;          // frames before the chunk: the cache, if any
    const int has_cache = a.cache != nullptr;
    auto load_chunk = [&]() __attribute__((always_inline)) {
        const int c_raw = ld_cc * CV_BK + ch4 * 4;
        const int c_ok = c_raw < a.Cin;
        const int c = c_ok ? c_raw : 0;
        const int dt = ld_dt, dy = ld_dy, dx = ld_dx, tap = ld_tap;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ti = vt_[i] + dt - (a.kt - 1);
            int yy = vy_[i] + dy - a.kh / 2, xx = vx_[i] + dx - a.kw / 2;
            // bitwise on purpose: `&&` compiles to nested exec-mask branches here
            const int ok = c_ok & (vt_[i] >= 0) & (yy >= 0) & (yy < a.Ho) & (xx >= 0) & (xx < a.Wo) &
                           ((ti >= 0) | (has_cache & (a.tc + ti >= 0)));
            yy = min(max(yy, 0), a.Ho - 1);
            xx = min(max(xx, 0), a.Wo - 1);
            if (a.up2) { yy >>= 1; xx >>= 1; }
            const int tt = max(ti >= 0 ? ti : a.tc + ti, 0);          // (ti <= vt < T always: the taps only reach back in time)
            const int vox = (tt * a.H + yy) * a.W + xx;               // < 2^31 voxels per launch (checked on the host)
            const float* src = (ti >= 0 ? a.x : base_neg) + (int64_t)vox * a.ldx + c;
            ra[i] = *(const float4*)src;
            ka[i] = ok ? 1.f : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = n0 + (tid >> 3) + 32 * i;
            rw[i] = *(const float4*)(a.w + (int64_t)min(row, a.Cout - 1) * a.ldw + (int64_t)tap * a.Cin + c);
            kw_[i] = (c_ok & (row < a.Cout)) ? 1.f : 0.f;
        }
        if (++ld_cc == ncc) {
            ld_cc = 0;
            ++ld_tap;
            if (++ld_dx == a.kw) {
                ld_dx = 0;
                if (++ld_dy == a.kh) { ld_dy = 0; ++ld_dt; }
            }
        }
    };
    // the mask is applied HERE, behind the MFMAs of the iteration, so the loads have the whole compute phase to land.
    // Multiplying (instead of selecting) keeps LLVM from sinking the loads back under the condition; the loaded values
    // are finite tensor data from valid addresses, so v * 0 == 0 and v * 1 == v exactly.
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        float* sa = smem + buf * (CV_BM + BN) * CV_LDS;
        float* sw = sa + CV_BM * CV_LDS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(float4*)(sa + ((tid >> 3) + 32 * i) * CV_LDS + ch4 * 4) =
                make_float4(ra[i].x * ka[i], ra[i].y * ka[i], ra[i].z * ka[i], ra[i].w * ka[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            *(float4*)(sw + ((tid >> 3) + 32 * i) * CV_LDS + ch4 * 4) =
                make_float4(rw[i].x * kw_[i], rw[i].y * kw_[i], rw[i].z * kw_[i], rw[i].w * kw_[i]);
    };

    f32x16_t acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    load_chunk();
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        if (kc + 1 < nchunk) load_chunk();
        const float* sa = smem + (kc & 1) * (CV_BM + BN) * CV_LDS + (wave * 32 + l31) * CV_LDS + g * 4;
        const float* sw = smem + (kc & 1) * (CV_BM + BN) * CV_LDS + CV_BM * CV_LDS + l31 * CV_LDS + g * 4;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const float4 xa = *(const float4*)(sa + k8 * 8);
            const float xv[4] = {xa.x, xa.y, xa.z, xa.w};
            float4 wv4[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wv4[nb] = *(const float4*)(sw + nb * 32 * CV_LDS + k8 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float wv = s == 0 ? wv4[nb].x : s == 1 ? wv4[nb].y : s == 2 ? wv4[nb].z : wv4[nb].w;
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv[s], acc[nb], 0, 0, 0);
                }
            }
        }
        if (kc + 1 < nchunk) store_chunk((kc + 1) & 1);
        __syncthreads();
    }

    cv_epilogue<NB, 1>(a, acc, m0 + wave * 32 + l31, n0, g);
}

// ---------------------------------------------------------------------------------------------
// vae_conv2_kernel — the same implicit GEMM with ONE wave per SIMD (r02).
//
// PMC of the kernel above at 1920x832 (profiles/r02c_pmc_vae_conv.txt): shader clock 2.38 GHz (NOT power-limited),
// matrix pipe busy 66 %, 8.7 non-MFMA instructions per MFMA.  Two waves per SIMD do not hide each other's VALU work
// (experiments/mfma_probe.hip, DESIGN.md 3.1: their MFMA and VALU streams serialize): 2 x (48 x 64 + 418 x 4)
// cycles per chunk pair = the measured time.  A wave ALONE on its SIMD hides the instructions it issues between its
// own MFMAs, so here:
//   * tile 256 voxels x 32*NB couts, 4 waves, wave w owns 64 voxels (two 32-blocks) x all couts: 2*NB accumulators,
//     8*NB MFMAs per k-step, a weight fragment feeds two MFMAs; LDS 2 x (256 + 32 NB) rows x 144 B = 99-108 KiB ->
//     one workgroup per CU;
//   * the loop body is ONE basic block (clamped addresses, 0/1 masks, counters advanced with selects) cut into 16
//     slots (k-step, sub-step); every slot = 2*NB MFMAs with one "piece" of staging work interleaved between them
//     by sched_group_barrier: store piece p of chunk kc+1 (registers -> LDS, mask applied) and load piece p of chunk
//     kc+2 (global -> the same registers), i.e. the global loads have a whole chunk (> 6000 cycles) to land;
//   * same chunk order, same accumulation order: the same bits as the 128-voxel kernel.
// ---------------------------------------------------------------------------------------------
#define CV2_BM 256

template <int NB>
__global__ __launch_bounds__(CV_THREADS, 1) void vae_conv2_kernel(const ConvArgs a) {
    constexpr int BN = 32 * NB;
    constexpr int ROWS = CV2_BM + BN;
    constexpr int NP = 8 + NB;                       // staging pieces per chunk and thread: 8 voxel rows + NB weight rows
    __shared__ __attribute__((aligned(16))) float smem[2 * ROWS * CV_LDS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * CV2_BM;
    const int n0 = blockIdx.y * BN;
    const int ch4 = tid & 7, r0 = tid >> 3;
    int vt_[8], vy_[8], vx_[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + r0 + 32 * i;
        if (m < a.M) {
            const int64_t hw = (int64_t)a.Ho * a.Wo;
            vt_[i] = (int)(m / hw);
            const int rem = (int)(m - (int64_t)vt_[i] * hw);
            vy_[i] = rem / a.Wo;
            vx_[i] = rem - vy_[i] * a.Wo;
        } else {
            vt_[i] = -1000000; vy_[i] = 0; vx_[i] = 0;
        }
    }
    const int ncc = (a.Cin + CV_BK - 1) / CV_BK;
    const int ntap = a.kt * a.kh * a.kw;
    const int nchunk = ntap * ncc;
    const float* const base_neg = a.cache ? a.cache : a.x;
    const int has_cache = a.cache != nullptr;

    float4 rr[NP];          // staged rows in flight (pieces 0-7: voxels, 8..: weights)
    float km[NP];           // their 0/1 masks
    int ld_cc = 0, ld_dt = 0, ld_dy = 0, ld_dx = 0, ld_tap = 0;      // coordinates of the chunk being loaded
    auto load_piece = [&](int p) __attribute__((always_inline)) {
        const int c_raw = ld_cc * CV_BK + ch4 * 4;
        const int c_ok = c_raw < a.Cin;
        const int c = c_ok ? c_raw : 0;
        if (p < 8) {
            const int ti = vt_[p] + ld_dt - (a.kt - 1);
            int yy = vy_[p] + ld_dy - a.kh / 2, xx = vx_[p] + ld_dx - a.kw / 2;
            const int ok = c_ok & (vt_[p] >= 0) & (yy >= 0) & (yy < a.Ho) & (xx >= 0) & (xx < a.Wo) &
                           ((ti >= 0) | (has_cache & (a.tc + ti >= 0)));
            yy = min(max(yy, 0), a.Ho - 1);
            xx = min(max(xx, 0), a.Wo - 1);
            if (a.up2) { yy >>= 1; xx >>= 1; }
            const int tt = max(ti >= 0 ? ti : a.tc + ti, 0);
            const int vox = (tt * a.H + yy) * a.W + xx;
            rr[p] = *(const float4*)((ti >= 0 ? a.x : base_neg) + (int64_t)vox * a.ldx + c);
            km[p] = ok ? 1.f : 0.f;
        } else {
            const int row = n0 + r0 + 32 * (p - 8);
            rr[p] = *(const float4*)(a.w + (int64_t)min(row, a.Cout - 1) * a.ldw + (int64_t)ld_tap * a.Cin + c);
            km[p] = (c_ok & (row < a.Cout)) ? 1.f : 0.f;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {      // next chunk, branch-free; sticks at the last one
        const int last = (ld_tap == ntap - 1) & (ld_cc == ncc - 1);
        const int cc = ld_cc + 1;
        const int w1 = cc == ncc;
        const int dx = ld_dx + w1;
        const int w2 = dx == a.kw;
        const int dy = ld_dy + w2;
        const int w3 = dy == a.kh;
        ld_cc = last ? ld_cc : (w1 ? 0 : cc);
        ld_tap = last ? ld_tap : ld_tap + w1;
        ld_dx = last ? ld_dx : (w2 ? 0 : dx);
        ld_dy = last ? ld_dy : (w3 ? 0 : dy);
        ld_dt = last ? ld_dt : ld_dt + w3;
    };
    auto store_piece = [&](int p, int buf) __attribute__((always_inline)) {
        float* dst = smem + buf * ROWS * CV_LDS + ((p < 8 ? 0 : CV2_BM) + r0 + 32 * (p < 8 ? p : p - 8)) * CV_LDS + ch4 * 4;
        *(float4*)dst = make_float4(rr[p].x * km[p], rr[p].y * km[p], rr[p].z * km[p], rr[p].w * km[p]);
    };

    f32x16_t acc[2][NB];      // [voxel block][cout block]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // prologue: chunk 0 -> LDS buffer 0, chunk 1 -> registers
#pragma unroll
    for (int p = 0; p < NP; ++p) load_piece(p);
    advance();
#pragma unroll
    for (int p = 0; p < NP; ++p) store_piece(p, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) load_piece(p);
    advance();
    __syncthreads();

    for (int kc = 0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        const float* sa = smem + buf * ROWS * CV_LDS + (wave * 64 + l31) * CV_LDS + g * 4;
        const float* sw = smem + buf * ROWS * CV_LDS + CV2_BM * CV_LDS + l31 * CV_LDS + g * 4;
        float4 xa[2][2], wf[2][NB];          // fragment double buffer over the k-steps
#pragma unroll
        for (int j = 0; j < 2; ++j) xa[0][j] = *(const float4*)(sa + j * 32 * CV_LDS);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[0][nb] = *(const float4*)(sw + nb * 32 * CV_LDS);
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const int cur = k8 & 1, nxt = cur ^ 1;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int slot = k8 * 4 + s4;
                if (s4 == 0 && k8 < 3) {     // fragments of the next k-step
#pragma unroll
                    for (int j = 0; j < 2; ++j) xa[nxt][j] = *(const float4*)(sa + j * 32 * CV_LDS + (k8 + 1) * 8);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) wf[nxt][nb] = *(const float4*)(sw + nb * 32 * CV_LDS + (k8 + 1) * 8);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float4 w4 = wf[cur][nb];
                    const float wv = s4 == 0 ? w4.x : s4 == 1 ? w4.y : s4 == 2 ? w4.z : w4.w;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 x4 = xa[cur][j];
                        const float xv = s4 == 0 ? x4.x : s4 == 1 ? x4.y : s4 == 2 ? x4.z : x4.w;
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv, acc[j][nb], 0, 0, 0);
                    }
                }
                if (slot < NP) {             // piece `slot`: registers of chunk kc+1 -> LDS, then chunk kc+2 -> the same registers
                    store_piece(slot, buf ^ 1);
                    load_piece(slot);
                }
                // interleave: after every MFMA of the slot a share of the slot's other instructions
#pragma unroll
                for (int mm = 0; mm < 2 * NB; ++mm) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                    if (mm == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NB, 0);
                    if (mm == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (mm == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        advance();
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) cv_epilogue<NB, NB>(a, acc[j], m0 + wave * 64 + j * 32 + l31, n0, g);
}

static int g_vae_conv_variant = 2;   // 2 = one wave per SIMD (vae_conv2_kernel) where the tile fits, 1 = the 128-voxel kernel
extern "C" void mg_vae_set_conv_variant(int v) { g_vae_conv_variant = v; }

static int launch_conv(const ConvArgs& a, hipStream_t st) {
    if ((int64_t)(a.T > a.tc ? a.T : a.tc) * a.H * a.W > 0x7fffffffLL) return MG_ERR_SHAPE;   // 32-bit voxel index in the gather
    const int64_t tiles_m = (a.M + CV_BM - 1) / CV_BM;
    if (tiles_m > 0x7fffffffLL) return MG_ERR_SHAPE;
    int nb;
    if (a.Cout <= 32) nb = 1;
    else if (a.Cout % 128 == 0) nb = 4;
    else if (a.Cout % 96 == 0) nb = 3;
    else nb = 4;
    const int bn = 32 * nb;
    if (g_vae_conv_variant == 2 && nb >= 3 && a.M >= 4 * CV2_BM) {
        const dim3 grid2((unsigned)((a.M + CV2_BM - 1) / CV2_BM), (unsigned)((a.Cout + bn - 1) / bn)), block2(CV_THREADS);
        if (nb == 3) hipLaunchKernelGGL(vae_conv2_kernel<3>, grid2, block2, 0, st, a);
        else hipLaunchKernelGGL(vae_conv2_kernel<4>, grid2, block2, 0, st, a);
        return mg_check_launch();
    }
    const dim3 grid((unsigned)tiles_m, (unsigned)((a.Cout + bn - 1) / bn)), block(CV_THREADS);
    if (nb == 1) hipLaunchKernelGGL(vae_conv_kernel<1>, grid, block, 0, st, a);
    else if (nb == 3) hipLaunchKernelGGL(vae_conv_kernel<3>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(vae_conv_kernel<4>, grid, block, 0, st, a);
    return mg_check_launch();
}

extern "C" int mg_vae_conv_f32(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                               const float* w, const float* bias, int Cout, int kt, int kh, int kw, int up2,
                               const float* residual, float* out, void* stream) {
    if (!x || !w || !out) return MG_ERR_ARG;
    if (T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || kt < 1 || kh < 1 || kw < 1 ||
        !(kh & 1) || !(kw & 1) || tc < 0 || tc > kt - 1 || (tc > 0 && !cache))
        return MG_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || (cache && ((uintptr_t)cache & 15)) ||
        (bias && ((uintptr_t)bias & 15)) || (residual && ((uintptr_t)residual & 15)))
        return MG_ERR_SHAPE;
    ConvArgs a;
    a.x = x; a.cache = cache; a.tc = tc; a.T = T; a.H = H; a.W = W; a.Cin = Cin; a.ldx = Cin;
    a.w = w; a.ldw = (int64_t)kt * kh * kw * Cin; a.bias = bias; a.Cout = Cout; a.kt = kt; a.kh = kh; a.kw = kw;
    a.up2 = up2 ? 1 : 0; a.residual = residual; a.out = out; a.ldo = Cout;
    a.Ho = up2 ? 2 * H : H; a.Wo = up2 ? 2 * W : W; a.M = (int64_t)T * a.Ho * a.Wo; a.out_scale = 1.f;
    return launch_conv(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// RMS_norm over channels (+ SiLU) — vae.py:39-54: F.normalize(x, dim=C) * sqrt(C) * gamma
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_rmsnorm_silu_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               float* __restrict__ out, int64_t rows, int C,
                                                               int do_silu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float sc = sqrtf((float)C);
    const int nv = C >> 2;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float4* xr = (const float4*)(x + row * C);
        float4 v[2];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        ss = wave_sum(ss);
        const float inv = sc / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 gm = ((const float4*)gamma)[c];
                float4 y = make_float4(v[i].x * inv * gm.x, v[i].y * inv * gm.y, v[i].z * inv * gm.z,
                                       v[i].w * inv * gm.w);
                if (do_silu) { y.x = silu(y.x); y.y = silu(y.y); y.z = silu(y.z); y.w = silu(y.w); }
                ((float4*)(out + row * C))[c] = y;
            }
        }
    }
}

extern "C" int mg_vae_rmsnorm_silu_f32(const float* x, const float* gamma, float* out, int64_t rows, int C,
                                       int do_silu, void* stream) {
    if (!x || !gamma || !out) return MG_ERR_ARG;
    if (C <= 0 || (C & 3) || C > 512 || rows < 0) return MG_ERR_SHAPE;
    if (rows == 0) return MG_OK;
    int64_t g = (rows + 3) / 4;
    if (g > 65536 * 8) g = 65536 * 8;
    hipLaunchKernelGGL(vae_rmsnorm_silu_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       out, rows, C, do_silu);
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// AttentionBlock — vae.py:247-256.  S = q k^T / sqrt(C) and o = softmax(S) v as two implicit-GEMM
// launches around a row softmax; S (L x L fp32) lives in caller workspace.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int64_t L, int64_t ld) {
    __shared__ float red[4];
    float* row = s + (int64_t)blockIdx.x * ld;
    for (int64_t i = L + threadIdx.x; i < ld; i += 256) row[i] = 0.f;  // zero the row padding
    float mx = -3.0e38f;
    for (int64_t i = threadIdx.x; i < L; i += 256) mx = fmaxf(mx, row[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int64_t i = threadIdx.x; i < L; i += 256) {
        const float e = expf(row[i] - mx);
        row[i] = e;
        sum += e;
    }
    sum = block_sum<256>(sum, red);
    const float inv = 1.f / sum;
    for (int64_t i = threadIdx.x; i < L; i += 256) row[i] *= inv;
}

__global__ void transpose_f32_kernel(const float* __restrict__ in, int64_t ldin, float* __restrict__ out,
                                     int64_t rows, int cols, int64_t ldout) {
    // in [rows][cols] (row stride ldin) -> out [cols][ldout], zero for r in [rows, ldout)
    __shared__ float tile[32][33];
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int64_t r = r0 + i;
        const int c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? in[r * ldin + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = c0 + i;
        const int64_t r = r0 + threadIdx.x;
        if (c < cols && r < ldout) out[(int64_t)c * ldout + r] = r < rows ? tile[threadIdx.x][i] : 0.f;
    }
}

// queries are processed in blocks of VAE_ATTN_QB rows: the score block S [QB][L] (204 MB at L = 24 960) stays inside the
// 256 MB Infinity Cache between the GEMM that writes it, the three softmax sweeps and the GEMM that reads it, and the
// workspace is (QB + C) * L floats instead of the full L x L matrix (2.5 GB at 1920x832).  Same arithmetic per row.
#define VAE_ATTN_QB 2048

extern "C" int64_t mg_vae_attn_workspace_floats(int64_t L, int C) {
    const int64_t Lp = (L + 3) & ~(int64_t)3;
    const int64_t qb = L < VAE_ATTN_QB ? L : VAE_ATTN_QB;
    return (qb + C) * Lp;
}

extern "C" int mg_vae_attn_f32(const float* qkv, float* out, int frames, int64_t L, int C, float* workspace,
                               void* stream) {
    if (!qkv || !out || !workspace) return MG_ERR_ARG;
    if (frames <= 0 || L <= 0 || C <= 0 || (C & 3) || L > 0x7ffffff0LL) return MG_ERR_SHAPE;
    if ((uintptr_t)workspace & 15) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t Lp = (L + 3) & ~(int64_t)3;  // row stride of S / V^T, 16-byte aligned rows
    const int64_t QB = L < VAE_ATTN_QB ? L : VAE_ATTN_QB;
    float* S = workspace;                 // [QB][Lp]
    float* vT = workspace + QB * Lp;      // [C][Lp]
    for (int f = 0; f < frames; ++f) {
        const float* base = qkv + (int64_t)f * L * 3 * C;
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((Lp + 31) / 32), (unsigned)((C + 31) / 32)),
                           dim3(32, 8), 0, st, base + 2 * C, (int64_t)3 * C, vT, L, C, Lp);
        for (int64_t q0 = 0; q0 < L; q0 += QB) {
            const int64_t nq = L - q0 < QB ? L - q0 : QB;
            ConvArgs a;
            // S[nq][L] = q[q0.. ][C] . k[L][C]^T * C^-1/2
            a.x = base + q0 * 3 * C; a.cache = nullptr; a.tc = 0; a.T = 1; a.H = 1; a.W = (int)nq; a.Cin = C; a.ldx = 3 * C;
            a.w = base + C; a.ldw = 3 * C; a.bias = nullptr; a.Cout = (int)L; a.kt = a.kh = a.kw = 1; a.up2 = 0;
            a.residual = nullptr; a.out = S; a.ldo = Lp; a.Ho = 1; a.Wo = (int)nq; a.M = nq;
            a.out_scale = 1.f / sqrtf((float)C);
            int rc = launch_conv(a, st);
            if (rc) return rc;
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)nq), dim3(256), 0, st, S, L, Lp);
            // out[nq][C] = P[nq][Lp] . vT[C][Lp]^T   (padding columns are zero on both sides)
            a.x = S; a.ldx = Lp; a.Cin = (int)Lp; a.w = vT; a.ldw = Lp; a.Cout = C; a.out = out + ((int64_t)f * L + q0) * C;
            a.ldo = C; a.out_scale = 1.f;
            rc = launch_conv(a, st);
            if (rc) return rc;
        }
    }
    return mg_check_launch();
}

// ---------------------------------------------------------------------------------------------
// layout / glue kernels
// ---------------------------------------------------------------------------------------------
__global__ void latent_in_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                 const float* __restrict__ inv_std, int C, int64_t thw, float* __restrict__ out) {
    const int64_t total = thw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        out[i] = z[(int64_t)c * thw + v] / inv_std[c] + mean[c];  // z / scale[1] + scale[0], vae.py:546-551
    }
}

extern "C" int mg_vae_latent_in_f32(const float* z, const float* mean, const float* inv_std, int C, int T, int H,
                                    int W, float* out, void* stream) {
    if (!z || !mean || !inv_std || !out) return MG_ERR_ARG;
    if (C <= 0 || T <= 0 || H <= 0 || W <= 0) return MG_ERR_SHAPE;
    const int64_t thw = (int64_t)T * H * W;
    int64_t g = (thw * C + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(latent_in_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, z, mean, inv_std, C,
                       thw, out);
    return mg_check_launch();
}

__global__ void video_out_kernel(const float* __restrict__ x, int C, int T, int64_t hw, float* __restrict__ out,
                                 int t_off, int T_total) {
    // x [T][HW][C] channels-last -> out[c][t_off + t][hw], clamped to [-1, 1] (vae.py:661)
    const int64_t total = (int64_t)T * hw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % hw;
        const int64_t r = i / hw;
        const int t = (int)(r % T), c = (int)(r / T);
        const float v = x[((int64_t)t * hw + p) * C + c];
        out[((int64_t)c * T_total + t_off + t) * hw + p] = fminf(1.f, fmaxf(-1.f, v));
    }
}

extern "C" int mg_vae_video_out_f32(const float* x, int C, int T, int H, int W, float* out, int t_off, int T_total,
                                    void* stream) {
    if (!x || !out) return MG_ERR_ARG;
    if (C <= 0 || T <= 0 || H <= 0 || W <= 0 || t_off < 0 || t_off + T > T_total) return MG_ERR_SHAPE;
    const int64_t hw = (int64_t)H * W;
    int64_t g = ((int64_t)T * hw * C + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(video_out_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, C, T, hw, out,
                       t_off, T_total);
    return mg_check_launch();
}

// decoded video [3][T][H][W] fp32 -> uint8 frames [T][H][W][3], the arithmetic of the reference's
// cache_video for one video (wan/utils/utils.py:39-47: clamp to the value range, torchvision
// make_grid normalisation (x - lo) / max(hi - lo, 1e-5), * 255, truncating cast)
template <bool ROUND>
__global__ void video_to_u8_kernel(const float* __restrict__ v, int T, int64_t hw, float lo, float hi,
                                   uint8_t* __restrict__ out) {
    const int64_t total = (int64_t)T * hw;
    const float span = fmaxf(hi - lo, 1e-5f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t px[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = fminf(hi, fmaxf(lo, v[(int64_t)c * total + i]));
            const float y = ((x - lo) / span) * 255.f;
            // ROUND: torchvision save_image (mul 255, add 0.5, clamp 0..255, truncate) — the reference's cache_image
            px[c] = (uint8_t)(int)(ROUND ? fminf(255.f, fmaxf(0.f, y + 0.5f)) : y);
        }
        out[i * 3 + 0] = px[0];
        out[i * 3 + 1] = px[1];
        out[i * 3 + 2] = px[2];
    }
}

extern "C" int mg_video_to_u8(const float* video, int T, int H, int W, float lo, float hi, uint8_t* frames, void* stream) {
    if (!video || !frames) return MG_ERR_ARG;
    if (T <= 0 || H <= 0 || W <= 0 || !(hi >= lo)) return MG_ERR_SHAPE;
    const int64_t hw = (int64_t)H * W;
    int64_t g = ((int64_t)T * hw + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(video_to_u8_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, video, T, hw, lo, hi, frames);
    return mg_check_launch();
}

extern "C" int mg_image_to_u8(const float* image, int H, int W, float lo, float hi, uint8_t* pixels, void* stream) {
    if (!image || !pixels) return MG_ERR_ARG;
    if (H <= 0 || W <= 0 || !(hi >= lo)) return MG_ERR_SHAPE;
    const int64_t hw = (int64_t)H * W;
    int64_t g = (hw + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(video_to_u8_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, image, 1, hw, lo, hi, pixels);
    return mg_check_launch();
}

__global__ void time_interleave_kernel(const float* __restrict__ x, int T, int64_t hw, int C, float* __restrict__ out) {
    // x [T][hw][2C] -> out [2T][hw][C]: frame 2t <- channels [0,C), frame 2t+1 <- [C,2C)  (vae.py:133-137)
    const int64_t total = (int64_t)2 * T * hw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int64_t p = r % hw;
        const int t2 = (int)(r / hw);
        out[i] = x[(((int64_t)(t2 >> 1)) * hw + p) * 2 * C + (t2 & 1) * C + c];
    }
}

extern "C" int mg_vae_time_interleave_f32(const float* x, int T, int64_t HW, int C, float* out, void* stream) {
    if (!x || !out) return MG_ERR_ARG;
    if (T <= 0 || HW <= 0 || C <= 0) return MG_ERR_SHAPE;
    int64_t g = ((int64_t)2 * T * HW * C + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(time_interleave_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, T, HW, C, out);
    return mg_check_launch();
}
