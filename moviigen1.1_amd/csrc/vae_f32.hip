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
// NB = 0: the Cout <= 4 form (the decoder head, 96 -> 3): same 128-voxel tile and gather, 4 weight rows, and the
// contraction on v_mfma_f32_4x4x1_16B_f32 — 16 independent 4 couts x 4 voxels outer products per instruction: lanes 0-31
// carry the wave's 32 voxels for the even float4 of every 8 channels, lanes 32-63 the same voxels for the odd one, and the
// two halves are added once at the end.  8 cycles per k instead of 64: the 32x32 tile burned 29 of its 32 cout columns.
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

// a page of zeros: what a tap in the zero padding reads (indexed by the channel offset, < 1024 floats — checked on the host)
__device__ __attribute__((aligned(16))) float g_cv_zero_page[1024 + 32];

struct ConvArgs {
    const float* x; const float* cache; int tc; int T, H, W, Cin; int64_t ldx;
    const float* w; int64_t ldw; const float* bias; int Cout; int kt, kh, kw; int up2;
    const float* residual; float* out; int64_t ldo; int Ho, Wo; int64_t M; float out_scale;
    int phases = 0; int64_t w_phase_stride = 0;      // phases: blockIdx.z = 2 py + px is one of the four 2x2 phase convolutions (below)
    int ksplit = 0; int64_t part_stride = 0;         // ksplit > 1 (1x1x1 GEMMs only): blockIdx.z = K slice, raw partial sums go to out + z * part_stride
    int xw0 = 0;                                     // column window (mg_vae_conv_cols_f32 / mg_vae_upconv_phases_cols_f32: a rank's W band of a decode split over GPUs):
                                                     // the conv grid is Ho x Wo with Wo <= W columns, its column c reads input column xw0 + c (+ tap offset); the zero
                                                     // padding begins outside [0, W) of the INPUT, whose columns left and right of the window are the neighbours' halos;
                                                     // out / residual are compact ([.][Ho][Wo], phases: [.][2 H][2 Wo]).  Not with up2.
};

// Output epilogue shared by both conv kernels: lane (l31, g) owns voxel row m and, per cout block nb and quad rq, the four
// consecutive couts n = n0 + 32 nb + 8 rq + 4 g .. +3.  Bias and residual are LOADED IN BATCHES (all NB*4 float4 of a voxel
// row before any is used): one load at a time, hipcc puts an `s_waitcnt vmcnt(0)` behind every load — NB*4 serialized
// HBM round trips per tile with the matrix pipe idle.
template <int NB, int GRP>      // GRP = cout blocks whose loads are in flight together (register budget of the caller)
MG_DEV void cv_epilogue(const ConvArgs& a, const f32x16_t (&acc)[NB], int64_t m_in, int64_t m, int n0, int g) {
    if (m_in >= a.M) return;      // m_in: the voxel of the conv grid, m: its row in out / residual
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

// FAST = the opt-in split-bf16 mode (mode = MG_VAE_BF16X3): every fp32 operand is split once, when its tile is staged, into
// hi = bf16(x) and lo = bf16(x - hi) — 16 mantissa bits — and the product runs as W_hi.X_hi + W_hi.X_lo + W_lo.X_hi on
// v_mfma_f32_32x32x16_bf16 with the same fp32 accumulators (the dropped lo.lo term is 2^-18 of a product): 6 MFMAs of 32
// cycles per 32-channel chunk and cout block instead of 16 of 64.  LDS rows keep their 144 bytes: 32 hi (64 B) | 32 lo
// (64 B) | 16 B pad — fragment reads of 16 consecutive rows still land on 16 distinct 16-byte slots (9 r mod 16).
// NOT the reference's arithmetic: results agree with the exact mode to ~1e-5 relative (test_vae_fast_mode), never the default.
// MB = 128-voxel sub-tiles per workgroup (1 = what runs; 2 = measurement variant, see launch_conv): with MB = 2 a wave owns
// 2 x 32 voxels, every staged weight row and every weight fragment read feeds twice the MFMAs (non-MFMA instructions per MFMA:
// 0.63 -> 0.44 at NB = 3, 0.56 -> 0.38 at NB = 4); the price is LDS for one workgroup per CU instead of two — and it loses.
template <int NB, bool FAST = false, int MB = 1>
__global__ __launch_bounds__(CV_THREADS) void vae_conv_kernel(const ConvArgs a) {
    constexpr int BN = NB ? 32 * NB : 4;
    constexpr int NW = NB ? NB : 1;                   // weight rows a thread stages per chunk (NB = 0: threads 0-31 stage the 4 rows)
    constexpr int BM = CV_BM * MB;                    // voxels per workgroup
    constexpr int NR = 4 * MB;                        // voxel rows a thread stages per chunk: rows (tid >> 3) + 32 i
    static_assert(NB || !FAST, "the Cout <= 4 form is exact only");
    static_assert(NB || MB == 1, "the Cout <= 4 form uses the 128-voxel tile");
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * CV_LDS];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l31 = lane & 31, g = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // Phase mode (mg_vae_upconv_phases_f32): "3x3 conv of the nearest-2x upsampled image" = four 2x2 convs of the image
    // itself, one per output parity (py, px): output row 2y+py reads upsampled rows 2y+py-1 .. 2y+py+1 = image rows
    // {y-1, y, y} (py = 0) or {y, y, y+1} (py = 1) — taps that share an image row share a pre-summed weight
    // (mg_vae_upconv_fold_weights_f32).  The conv grid is the INPUT grid, the tap origin is (py-1, px-1) and the result is
    // scattered to (2y+py, 2x+px): 4 instead of 9 taps per output, and the taps go through the cheap pointer path.
    const int ph = a.phases ? (int)blockIdx.z : 0;
    const int oy = a.phases ? (ph >> 1) - 1 : -(a.kh / 2), ox = a.phases ? (ph & 1) - 1 : -(a.kw / 2);
    const float* const wbase = a.w + ph * a.w_phase_stride;

    // ---- gather bookkeeping: this thread stages rows (tid>>3)+32i, float4 column tid&7 -----------
    const int ch4 = tid & 7;
    int vt_[NR], vy_[NR], vx_[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
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

    // Staging (r02).  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate: VALU instructions do not hide behind it, whichever
    // wave issues them (PMC, profiles/r02c_pmc_vae_conv.txt: matrix pipe busy 66 % at full clock with 5.0 VALU + 3.1 SALU
    // per MFMA; a one-wave-per-SIMD variant with interleaved staging measured the same 66 %,
    // experiments/vae_conv2_one_wave_per_simd.hip) — so the staging work itself is cut down:
    //   * the row pointers of a tap are computed ONCE per tap (clamp, padding test, cache / frame select, 64-bit address:
    //     ~25 VALU per row) and only advanced by the channel offset for the Cin/32 chunks of that tap;
    //   * a tap that falls into the zero padding (space, or time before the first cached frame) points at a page of zeros
    //     instead of being masked: no select, no multiply, no branch around the load;
    //   * chunk coordinates advance by counters, not divisions.
    float4 ra[NR], rw[NW];
    const float* pa[NR];                    // voxel rows of the tap being loaded (or the zero page)
    const float* pw[NW];                    // weight rows, advanced by Cin per tap
    int ld_cc = 0, ld_dt = 0, ld_dy = 0, ld_dx = 0;                 // coordinates of the NEXT chunk to load (wave-uniform)
    const float* const base_neg = a.cache ? a.cache : a.x;          // frames before the chunk: the cache, if any
    const int has_cache = a.cache != nullptr;
#pragma unroll
    for (int i = 0; i < NW; ++i) pw[i] = wbase + (int64_t)min(n0 + (tid >> 3) + 32 * i, a.Cout - 1) * a.ldw;   // rows >= Cout: never stored
    // Per row, once per tile: which temporal / vertical / horizontal tap offsets stay inside the tensor (bits dt | dy<<3 |
    // dx<<6); a tap is valid when its three bits are set.  For the convolutions without the folded 2x upsample the row
    // pointer of a tap is then `frame base of the row` (recomputed when dt changes: every kh*kw taps) + a WAVE-UNIFORM
    // offset ((dy - kh/2) W + (dx - kw/2)) ldx — ~10 VALU per row and tap instead of ~40.
    unsigned vmask[NR];
    const float* fbc[NR];
    const float* zp[NR];                   // what an invalid tap reads: the zero page — or, for rows past M (never stored), row 0 of x:
                                            // the 1x1 GEMMs of the attention block index it with channel offsets far beyond the page
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        unsigned mk = 0;
        const int valid = vt_[i] >= 0;                               // rows past M: every tap reads the zero page (never stored)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ti = vt_[i] + d - (a.kt - 1), yy = vy_[i] + d + oy, xx = vx_[i] + a.xw0 + d + ox;
            mk |= (unsigned)(valid & (d < a.kt) & ((ti >= 0) | (has_cache & (a.tc + ti >= 0)))) << d;
            mk |= (unsigned)(valid & (d < a.kh) & (yy >= 0) & (yy < a.Ho)) << (3 + d);
            mk |= (unsigned)(valid & (d < a.kw) & (xx >= 0) & (xx < a.W)) << (6 + d);      // (this mask serves the paths without up2: input width = W)
        }
        vmask[i] = mk;
        fbc[i] = a.x;
        zp[i] = valid ? g_cv_zero_page : a.x;
    }
    auto tap_pointers = [&]() __attribute__((always_inline)) {
        if (!a.up2) {
            if ((ld_dy | ld_dx) == 0) {                              // wave-uniform: first tap of a temporal offset
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int ti = vt_[i] + ld_dt - (a.kt - 1);
                    const int tt = max(ti >= 0 ? ti : a.tc + ti, 0);
                    const int vox = (tt * a.H + vy_[i]) * a.W + vx_[i] + a.xw0;       // only dereferenced when the row's bits are set
                    fbc[i] = (ti >= 0 ? a.x : base_neg) + (int64_t)max(vox, 0) * a.ldx;
                }
            }
            const int64_t off = (int64_t)((ld_dy + oy) * a.W + (ld_dx + ox)) * a.ldx;   // wave-uniform
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const unsigned ok = (vmask[i] >> ld_dt) & (vmask[i] >> (3 + ld_dy)) & (vmask[i] >> (6 + ld_dx)) & 1u;
                pa[i] = ok ? fbc[i] + off : zp[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ti = vt_[i] + ld_dt - (a.kt - 1);
            int yy = vy_[i] + ld_dy - a.kh / 2, xx = vx_[i] + ld_dx - a.kw / 2;
            const int ok = (vt_[i] >= 0) & (yy >= 0) & (yy < a.Ho) & (xx >= 0) & (xx < a.Wo) & ((ti >= 0) | (has_cache & (a.tc + ti >= 0)));
            yy = min(max(yy, 0), a.Ho - 1) >> 1;                     // nearest-exact 2x folded into the gather
            xx = min(max(xx, 0), a.Wo - 1) >> 1;
            const int tt = max(ti >= 0 ? ti : a.tc + ti, 0);          // (ti <= vt < T always: the taps only reach back in time)
            const int vox = (tt * a.H + yy) * a.W + xx;               // < 2^31 voxels per launch (checked on the host)
            pa[i] = ok ? (ti >= 0 ? a.x : base_neg) + (int64_t)vox * a.ldx : zp[i];
        }
    };
    auto load_chunk = [&]() __attribute__((always_inline)) {
        if (ld_cc == 0) tap_pointers();                              // wave-uniform: first chunk of a tap
        const int c_raw = ld_cc * CV_BK + ch4 * 4;
        const bool c_ok = c_raw < a.Cin;                             // false only in the last chunk of a tap when Cin % 32 != 0
        const int c = c_ok ? c_raw : 0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {         // (loaded as an ext-vector: a float4 struct copy from a selected pointer kept ra / rw in memory)
            const f32x4_t t = *(const f32x4_t*)(pa[i] + c);                   // k >= Cin: finite data x zero weights
            ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const f32x4_t t = *(const f32x4_t*)(c_ok ? pw[i] + c : g_cv_zero_page);
            rw[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
        if (++ld_cc == ncc) {
            ld_cc = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) pw[i] += a.Cin;
            if (++ld_dx == a.kw) {
                ld_dx = 0;
                if (++ld_dy == a.kh) { ld_dy = 0; ++ld_dt; }
            }
        }
    };
    auto split_store = [&](float* row, const float4& v) __attribute__((always_inline)) {      // FAST: 4 fp32 -> 4 hi + 4 lo bf16
        const float h0 = round_bf(v.x), h1 = round_bf(v.y), h2 = round_bf(v.z), h3 = round_bf(v.w);
        uint2 hi, lo;
        hi.x = pack_bf2(v.x, v.y); hi.y = pack_bf2(v.z, v.w);
        lo.x = pack_bf2(v.x - h0, v.y - h1); lo.y = pack_bf2(v.z - h2, v.w - h3);
        char* r = (char*)row;
        *(uint2*)(r + ch4 * 8) = hi;
        *(uint2*)(r + 64 + ch4 * 8) = lo;
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
        float* sa = smem + buf * (BM + BN) * CV_LDS;
        float* sw = sa + BM * CV_LDS;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < NR; ++i) split_store(sa + ((tid >> 3) + 32 * i) * CV_LDS, ra[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) split_store(sw + ((tid >> 3) + 32 * i) * CV_LDS, rw[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) *(float4*)(sa + ((tid >> 3) + 32 * i) * CV_LDS + ch4 * 4) = ra[i];
        if (NB == 0) {
            if (tid < 32) *(float4*)(sw + (tid >> 3) * CV_LDS + ch4 * 4) = rw[0];
            return;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) *(float4*)(sw + ((tid >> 3) + 32 * i) * CV_LDS + ch4 * 4) = rw[i];
    };

    f32x16_t acc[MB][NW];                             // [voxel sub-tile][cout block]: wave w owns rows 128 mb + 32 w + l31
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mb][i][e] = 0.f;
    f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f};              // NB = 0: couts 0..3 of the lane's voxel, this lane half's share of k

    // K slice of this workgroup (ksplit: 1x1x1 GEMMs with few output tiles — P.V of the attention block): channel chunks
    // [kc0, kc1) of the single tap
    int kc0 = 0, kc1 = nchunk;
    if (a.ksplit > 1) {
        kc0 = (int)((int64_t)nchunk * blockIdx.z / a.ksplit);
        kc1 = (int)((int64_t)nchunk * (blockIdx.z + 1) / a.ksplit);
        ld_cc = kc0;
        if (kc0) tap_pointers();                                      // load_chunk() computes them for chunk 0 of a tap only
    }
    load_chunk();
    store_chunk(kc0 & 1);
    __syncthreads();
    for (int kc = kc0; kc < kc1; ++kc) {
        if (kc + 1 < kc1) load_chunk();
        // pin the order loads | MFMAs | LDS stores: without the fences hipcc moves the ds_writes of the staged rows up in
        // front of the MFMA block (it can prove they touch the other buffer) and then waits for the loads right away
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const float* sa = smem + (kc & 1) * (BM + BN) * CV_LDS + (wave * 32 + l31) * CV_LDS + g * 4;      // sub-tile mb: + mb * 128 rows
        const float* sw = smem + (kc & 1) * (BM + BN) * CV_LDS + BM * CV_LDS + l31 * CV_LDS + g * 4;
        if (NB == 0) {
            // A = weights: lane l holds w[cout l & 3][k]; B = voxels: lane l holds x[voxel l31][k]; block l >> 2 pairs them
            const float* s4 = smem + (kc & 1) * (BM + BN) * CV_LDS + BM * CV_LDS + (lane & 3) * CV_LDS + g * 4;
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                const float4 xa = *(const float4*)(sa + k8 * 8);
                const float4 wa = *(const float4*)(s4 + k8 * 8);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.x, xa.x, acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.y, xa.y, acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.z, xa.z, acc4, 0, 0, 0);
                acc4 = __builtin_amdgcn_mfma_f32_4x4x1f32(wa.w, xa.w, acc4, 0, 0, 0);
            }
        } else if (FAST) {
            // lane (l31, g): row l31, k = 16 s + 8 g .. + 7 of k-step s: 16 bytes at byte 32 s + 16 g of the hi plane (lo: + 64)
            const char* xa = (const char*)(sa - g * 4) + g * 16;
            const char* wa = (const char*)(sw - g * 4) + g * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t xh[MB], xl[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    xh[mb] = *(const bf16x8_t*)(xa + mb * 128 * CV_LDS * 4 + ks * 32);
                    xl[mb] = *(const bf16x8_t*)(xa + mb * 128 * CV_LDS * 4 + 64 + ks * 32);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8_t wh = *(const bf16x8_t*)(wa + nb * 32 * CV_LDS * 4 + ks * 32);
                    const bf16x8_t wl = *(const bf16x8_t*)(wa + nb * 32 * CV_LDS * 4 + 64 + ks * 32);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[mb], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[mb], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[mb], acc[mb][nb], 0, 0, 0);
                    }
                }
            }
        } else
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            float xv[MB][4];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const float4 xa = *(const float4*)(sa + mb * 128 * CV_LDS + k8 * 8);
                xv[mb][0] = xa.x; xv[mb][1] = xa.y; xv[mb][2] = xa.z; xv[mb][3] = xa.w;
            }
            float4 wv4[NW];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wv4[nb] = *(const float4*)(sw + nb * 32 * CV_LDS + k8 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float wv = s == 0 ? wv4[nb].x : s == 1 ? wv4[nb].y : s == 2 ? wv4[nb].z : wv4[nb].w;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, xv[mb][s], acc[mb][nb], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < kc1) store_chunk((kc + 1) & 1);
        __syncthreads();
    }

    auto out_row = [&](int64_t m_in) __attribute__((always_inline)) {       // row of out / residual of conv-grid voxel m_in
        if (!a.phases || m_in >= a.M) return m_in;
        const int64_t hw = (int64_t)a.H * a.Wo;                                   // the conv grid: H x Wo (Wo = W unless a column window)
        const int t = (int)(m_in / hw);
        const int rem = (int)(m_in - (int64_t)t * hw);
        const int y = rem / a.Wo, x = rem - y * a.Wo;
        return ((int64_t)t * 2 * a.H + 2 * y + (ph >> 1)) * (2 * a.Wo) + 2 * x + (ph & 1);
    };
    if (NB == 0) {
        // the two lane halves hold the even / odd float4 of every 8 channels of the same voxel: add them, then lanes 0-31
        // apply (acc * scale + bias) + residual as cv_epilogue does and store the row's <= 4 couts
        const int64_t m_in = m0 + wave * 32 + l31;
        const int64_t m_out = out_row(m_in);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[i] += __shfl_xor(acc4[i], 32, 64);
        if (g == 0 && m_in < a.M) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < a.Cout) {
                    float v = acc4[i] * a.out_scale + (a.bias ? a.bias[i] : 0.f);
                    if (a.residual) v += a.residual[m_out * a.ldo + i];
                    a.out[m_out * a.ldo + i] = v;
                }
        }
        return;
    }
    ConvArgs b = a;
    if (a.ksplit > 1) {     // raw partial sums of this K slice; vae_ksplit_reduce_kernel adds the slices in a fixed order
        b.out = a.out + (int64_t)blockIdx.z * a.part_stride;
        b.bias = nullptr; b.residual = nullptr; b.out_scale = 1.f;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int64_t m_in = m0 + mb * 128 + wave * 32 + l31;
        cv_epilogue<NW, 1>(b, acc[mb], m_in, out_row(m_in), n0, g);
    }
}

// out[m][n] = (sum over the K slices z of part[z][m][n]) (fixed order: deterministic), rows of n floats, n % 4 == 0
__global__ __launch_bounds__(256) void vae_ksplit_reduce_kernel(const float* __restrict__ part, int64_t part_stride, int Z,
                                                                float* __restrict__ out, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s4 = ((const float4*)part)[i];
        for (int z = 1; z < Z; ++z) {
            const float4 p = ((const float4*)(part + (int64_t)z * part_stride))[i];
            s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w;
        }
        ((float4*)out)[i] = s4;
    }
}

// mode: MG_VAE_EXACT = fp32 MFMA (the reference's arithmetic), MG_VAE_BF16X3 = split-bf16 x 3 (opt-in fast mode) — an
// argument of every call (ABI 7): two decodes on two streams or threads cannot change each other's arithmetic
static int launch_conv(const ConvArgs& a, hipStream_t st, int mode) {
    if ((int64_t)(a.T > a.tc ? a.T : a.tc) * a.H * a.W > 0x7fffffffLL) return MG_ERR_SHAPE;   // 32-bit voxel index in the gather
    if (a.kt * a.kh * a.kw > 1 && a.Cin > 1024) return MG_ERR_SHAPE;                           // padding taps index the zero page by channel
    int nb;
    if (a.Cout <= 4 && !a.phases && a.ksplit <= 1) nb = 0;        // the decoder head (96 -> 3): v_mfma_f32_4x4x1, exact in either mode
    else if (a.Cout <= 32) nb = 1;
    else if (a.Cout % 128 == 0) nb = 4;
    else if (a.Cout % 96 == 0) nb = 3;
    else nb = 4;
    const int bn = nb ? 32 * nb : 4;
    if (a.ksplit > 1 && (a.phases || a.kt * a.kh * a.kw != 1)) return MG_ERR_ARG;
    // voxel tile: 128 per workgroup.  The 256-voxel form (MB = 2: weight staging and weight fragment reads shared by twice the
    // MFMAs, 0.63 -> 0.44 non-MFMA instructions per MFMA at NB = 3) is kept behind MG_VAE_TILE_256 for A/B runs only: it needs the
    // LDS of a whole CU, and with ONE workgroup per CU nothing fills the barrier and load-latency bubbles that a second
    // workgroup fills today — the 1920x832x81f decode takes 9.59 s with it against 8.93 s (profiles/r04e_vae_tiles.log).
    const int tile_flag = mode >> 8;        // bits 8-9 of `mode`: 0 / MG_VAE_TILE_128 = 128 voxels, MG_VAE_TILE_256 = 256 (measurements)
    mode &= 0xff;
    if (mode != MG_VAE_EXACT && mode != MG_VAE_BF16X3) return MG_ERR_ARG;
    const int mbt = (tile_flag == 2 && nb >= 3 && mode == MG_VAE_EXACT && a.ksplit <= 1) ? 2 : 1;
    const int bm = CV_BM * mbt;
    const int64_t tiles_m = (a.M + bm - 1) / bm;
    if (tiles_m > 0x7fffffffLL) return MG_ERR_SHAPE;
    const dim3 grid((unsigned)tiles_m, (unsigned)((a.Cout + bn - 1) / bn), a.phases ? 4u : a.ksplit > 1 ? (unsigned)a.ksplit : 1u), block(CV_THREADS);
    if (nb == 0) {
        hipLaunchKernelGGL((vae_conv_kernel<0, false>), grid, block, 0, st, a);
        return mg_check_launch();
    }
    if (mode == MG_VAE_BF16X3) {
        if (nb == 1) hipLaunchKernelGGL((vae_conv_kernel<1, true>), grid, block, 0, st, a);
        else if (nb == 3) hipLaunchKernelGGL((vae_conv_kernel<3, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((vae_conv_kernel<4, true>), grid, block, 0, st, a);
        return mg_check_launch();
    }
    if (mbt == 2) {
        if (nb == 3) hipLaunchKernelGGL((vae_conv_kernel<3, false, 2>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((vae_conv_kernel<4, false, 2>), grid, block, 0, st, a);
        return mg_check_launch();
    }
    if (nb == 1) hipLaunchKernelGGL((vae_conv_kernel<1, false>), grid, block, 0, st, a);
    else if (nb == 3) hipLaunchKernelGGL((vae_conv_kernel<3, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((vae_conv_kernel<4, false>), grid, block, 0, st, a);
    return mg_check_launch();
}

static int vae_conv_impl(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                         const float* w, const float* bias, int Cout, int kt, int kh, int kw, int up2,
                         const float* residual, float* out, int col0, int cols, int mode, void* stream) {
    if (!x || !w || !out) return MG_ERR_ARG;
    if (col0 < 0 || cols <= 0 || col0 + cols > W || (up2 && (col0 || cols != W))) return MG_ERR_SHAPE;
    if (T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || kt < 1 || kh < 1 || kw < 1 ||
        !(kh & 1) || !(kw & 1) || tc < 0 || tc > kt - 1 || (tc > 0 && !cache))
        return MG_ERR_SHAPE;
    // the tile gather keeps one validity bit per tap offset and axis in 3-bit fields: extents above 3 would alias
    // (WanVAE uses 1 and 3 only, reference vae.py:17-36)
    if (kt > 3 || kh > 3 || kw > 3) return MG_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)out & 15) || (cache && ((uintptr_t)cache & 15)) ||
        (bias && ((uintptr_t)bias & 15)) || (residual && ((uintptr_t)residual & 15)))
        return MG_ERR_SHAPE;
    ConvArgs a;
    a.x = x; a.cache = cache; a.tc = tc; a.T = T; a.H = H; a.W = W; a.Cin = Cin; a.ldx = Cin;
    a.w = w; a.ldw = (int64_t)kt * kh * kw * Cin; a.bias = bias; a.Cout = Cout; a.kt = kt; a.kh = kh; a.kw = kw;
    a.up2 = up2 ? 1 : 0; a.residual = residual; a.out = out; a.ldo = Cout;
    a.Ho = up2 ? 2 * H : H; a.Wo = up2 ? 2 * W : cols; a.M = (int64_t)T * a.Ho * a.Wo; a.out_scale = 1.f;
    a.phases = 0; a.w_phase_stride = 0; a.xw0 = col0;
    return launch_conv(a, (hipStream_t)stream, mode);
}

extern "C" int mg_vae_conv_f32(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                               const float* w, const float* bias, int Cout, int kt, int kh, int kw, int up2,
                               const float* residual, float* out, int mode, void* stream) {
    return vae_conv_impl(x, cache, tc, T, H, W, Cin, w, bias, Cout, kt, kh, kw, up2, residual, out, 0, W, mode, stream);
}

// the same convolution for the output columns [col0, col0 + cols) only, written compactly (out / residual [T][H][cols][Cout]): one rank's band of a
// decode split along W over several GPUs — x and cache are the band WITH its halo columns, the zero padding starts outside [0, W) of them
extern "C" int mg_vae_conv_cols_f32(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                                    const float* w, const float* bias, int Cout, int kt, int kh, int kw,
                                    const float* residual, float* out, int col0, int cols, int mode, void* stream) {
    return vae_conv_impl(x, cache, tc, T, H, W, Cin, w, bias, Cout, kt, kh, kw, 0, residual, out, col0, cols, mode, stream);
}

// w [Cout][1][3][3][Cin] -> wp [4 phases = 2 py + px][Cout][2][2][Cin]: the taps of a 3x3 kernel that fall on the same image
// pixel under the nearest-2x upsample, summed (py = 0: {w0 | w1 + w2}, py = 1: {w0 + w1 | w2}; the same along x)
__global__ void upconv_fold_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin) {
    const int64_t total = (int64_t)4 * Cout * 4 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        int64_t r = i / Cin;
        const int dx = (int)(r & 1), dy = (int)((r >> 1) & 1);
        r >>= 2;
        const int co = (int)(r % Cout), ph = (int)(r / Cout);
        const int py = ph >> 1, px = ph & 1;
        const int y0 = py == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), y1 = py == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
        const int x0 = px == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), x1 = px == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
        float acc = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) acc += w[(((int64_t)co * 3 + yy) * 3 + xx) * Cin + ci];
        wp[i] = acc;
    }
}

extern "C" int mg_vae_upconv_fold_weights_f32(const float* w, int Cout, int Cin, float* wp, void* stream) {
    if (!w || !wp) return MG_ERR_ARG;
    if (Cout <= 0 || Cin <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(upconv_fold_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin);
    return mg_check_launch();
}

static int vae_upconv_phases_impl(const float* x, int T, int H, int W, int Cin, const float* wp, const float* bias,
                                  int Cout, float* out, int col0, int cols, int mode, void* stream) {
    if (!x || !wp || !out) return MG_ERR_ARG;
    if (T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || Cout <= 0 || col0 < 0 || cols <= 0 || col0 + cols > W) return MG_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)wp & 15) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15))) return MG_ERR_SHAPE;
    if ((int64_t)T * 4 * H * W > 0x7fffffffLL) return MG_ERR_SHAPE;      // output voxels stay 32-bit like every other conv's
    ConvArgs a;
    a.x = x; a.cache = nullptr; a.tc = 0; a.T = T; a.H = H; a.W = W; a.Cin = Cin; a.ldx = Cin;
    a.w = wp; a.ldw = (int64_t)4 * Cin; a.bias = bias; a.Cout = Cout; a.kt = 1; a.kh = 2; a.kw = 2; a.up2 = 0;
    a.residual = nullptr; a.out = out; a.ldo = Cout; a.Ho = H; a.Wo = cols; a.M = (int64_t)T * H * cols; a.out_scale = 1.f;
    a.phases = 1; a.w_phase_stride = (int64_t)Cout * 4 * Cin; a.xw0 = col0;
    return launch_conv(a, (hipStream_t)stream, mode);
}

extern "C" int mg_vae_upconv_phases_f32(const float* x, int T, int H, int W, int Cin, const float* wp, const float* bias,
                                        int Cout, float* out, int mode, void* stream) {
    return vae_upconv_phases_impl(x, T, H, W, Cin, wp, bias, Cout, out, 0, W, mode, stream);
}

// the same for the image columns [col0, col0 + cols) only: out [T][2 H][2 cols][Cout] (a rank's band; x carries the halo columns)
extern "C" int mg_vae_upconv_phases_cols_f32(const float* x, int T, int H, int W, int Cin, const float* wp, const float* bias,
                                             int Cout, float* out, int col0, int cols, int mode, void* stream) {
    return vae_upconv_phases_impl(x, T, H, W, Cin, wp, bias, Cout, out, col0, cols, mode, stream);
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

// P.V of a query block is a GEMM with M = QB rows, N = C and K = L: 16 x 3 output tiles for 256 CUs.  Its K range is cut
// into VAE_ATTN_KSPLIT slices (one more grid dimension: 384 workgroups), each writes raw partial sums and a small kernel
// adds the slices in a fixed order — deterministic, unlike atomics.  (Measured before: 116 ms per 4-frame chunk at
// 104 x 240 for the attention block = 3.7x the cost per MAC of the 3x3x3 convolutions, most of it this launch at 19 %
// of the CUs; profiles/r04a_vae_stages.txt.)
#define VAE_ATTN_KSPLIT 8

static int64_t vae_attn_ksplit(int64_t nq, int C) {
    const int64_t tiles = ((nq + CV_BM - 1) / CV_BM) * ((C + 127) / 128);
    return tiles >= 256 ? 1 : VAE_ATTN_KSPLIT;
}

extern "C" int64_t mg_vae_attn_workspace_floats(int64_t L, int C) {
    const int64_t Lp = (L + 3) & ~(int64_t)3;
    const int64_t qb = L < VAE_ATTN_QB ? L : VAE_ATTN_QB;
    return (qb + C) * Lp + VAE_ATTN_KSPLIT * qb * C;
}

// General form: Lq query rows (row stride ldq, frame stride q_fs) against Lk keys / values (row stride ldkv, frame stride kv_fs), out [frames][Lq][C].
// A row's result does not depend on which other rows are computed with it (one dot product per score in a fixed order, a row-local softmax, P.V in
// VAE_ATTN_KSPLIT K slices of the same Lk added in a fixed order): a rank of a decode split along W computes the rows of its own pixels against the
// gathered keys / values and gets the bits of the one-GPU launch.
static int vae_attn_impl(const float* q, int64_t ldq, int64_t q_fs, const float* k, const float* v, int64_t ldkv, int64_t kv_fs, float* out,
                         int frames, int64_t Lq, int64_t L, int C, float* workspace, void* stream) {
    if (!q || !k || !v || !out || !workspace) return MG_ERR_ARG;
    if (frames <= 0 || L <= 0 || Lq <= 0 || Lq > L || C <= 0 || (C & 3) || L > 0x7ffffff0LL || (ldq & 3) || (ldkv & 3)) return MG_ERR_SHAPE;
    if (((uintptr_t)workspace & 15) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15)) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t Lp = (L + 3) & ~(int64_t)3;  // row stride of S / V^T, 16-byte aligned rows
    const int64_t QB = L < VAE_ATTN_QB ? L : VAE_ATTN_QB;
    float* S = workspace;                 // [QB][Lp]
    float* vT = workspace + QB * Lp;      // [C][Lp]
    float* part = vT + (int64_t)C * Lp;   // [KSPLIT][QB][C] partial sums of P.V
    for (int f = 0; f < frames; ++f) {
        const float* qf = q + (int64_t)f * q_fs;
        const float* kf = k + (int64_t)f * kv_fs;
        hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((Lp + 31) / 32), (unsigned)((C + 31) / 32)),
                           dim3(32, 8), 0, st, v + (int64_t)f * kv_fs, ldkv, vT, L, C, Lp);
        for (int64_t q0 = 0; q0 < Lq; q0 += QB) {
            const int64_t nq = Lq - q0 < QB ? Lq - q0 : QB;
            ConvArgs a;
            // S[nq][L] = q[q0.. ][C] . k[L][C]^T * C^-1/2
            a.x = qf + q0 * ldq; a.cache = nullptr; a.tc = 0; a.T = 1; a.H = 1; a.W = (int)nq; a.Cin = C; a.ldx = ldq;
            a.w = kf; a.ldw = ldkv; a.bias = nullptr; a.Cout = (int)L; a.kt = a.kh = a.kw = 1; a.up2 = 0;
            a.residual = nullptr; a.out = S; a.ldo = Lp; a.Ho = 1; a.Wo = (int)nq; a.M = nq;
            a.out_scale = 1.f / sqrtf((float)C);
            int rc = launch_conv(a, st, MG_VAE_EXACT);     // the attention block's two GEMMs are exact in either mode
            if (rc) return rc;
            hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)nq), dim3(256), 0, st, S, L, Lp);
            // out[nq][C] = P[nq][Lp] . vT[C][Lp]^T   (padding columns are zero on both sides)
            float* dst = out + ((int64_t)f * Lq + q0) * C;
            a.x = S; a.ldx = Lp; a.Cin = (int)Lp; a.w = vT; a.ldw = Lp; a.Cout = C; a.out = dst;
            a.ldo = C; a.out_scale = 1.f;
            const int Z = (int)vae_attn_ksplit(nq, C);
            if (Z > 1) {
                a.out = part; a.ksplit = Z; a.part_stride = nq * C;
            }
            rc = launch_conv(a, st, MG_VAE_EXACT);
            if (rc) return rc;
            if (Z > 1) {
                const int64_t total4 = nq * C / 4;
                hipLaunchKernelGGL(vae_ksplit_reduce_kernel, dim3((unsigned)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048)), dim3(256), 0,
                                   st, part, nq * C, Z, dst, total4);
                a.ksplit = 0; a.part_stride = 0;
            }
        }
    }
    return mg_check_launch();
}

extern "C" int mg_vae_attn_f32(const float* qkv, float* out, int frames, int64_t L, int C, float* workspace,
                               void* stream) {
    if (!qkv) return MG_ERR_ARG;
    return vae_attn_impl(qkv, 3 * (int64_t)C, L * 3 * C, qkv + C, qkv + 2 * C, 3 * (int64_t)C, L * 3 * C, out, frames, L, L, C, workspace, stream);
}

// rows of ONE band: q [frames][Lq][>= C] (row stride ldq), k / v [frames][Lk][..] (row stride ldkv, e.g. the two halves of a gathered k|v tensor),
// out [frames][Lq][C]; workspace: mg_vae_attn_workspace_floats(Lk, C) floats
extern "C" int mg_vae_attn_rows_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, int frames,
                                    int64_t Lq, int64_t Lk, int C, float* workspace, void* stream) {
    return vae_attn_impl(q, ldq, Lq * ldq, k, v, ldkv, Lk * ldkv, out, frames, Lq, Lk, C, workspace, stream);
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
