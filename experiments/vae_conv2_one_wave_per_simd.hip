// ARCHIVED EXPERIMENT (round 2) — not built.  The WanVAE implicit-GEMM convolution with a 256-voxel tile and ONE wave per
// SIMD, its staging work interleaved between the MFMAs by sched_group_barrier (the structure that pays for the bf16
// kernels).  Measured (profiles/r02e_pmc_vae_conv2.txt): the 96-channel 1920x832 convolutions take exactly as long as
// with the 128-voxel / two-waves-per-SIMD kernel (933.6 vs 933.5 ms for 45 launches, matrix pipe busy 66.1 vs 65.9 %), the
// low-resolution stages lose to the coarser tile grid (293 workgroups on 256 CUs).  Reason: v_mfma_f32_32x32x2_f32 runs
// at the fp32 VECTOR rate — VALU instructions do not hide behind it, whichever wave they come from; time = MFMA cycles
// + VALU cycles.  The lever is FEWER VALU per MFMA (csrc/vae_f32.hip: per-tap row pointers, zero page), not hiding them.
// Depends on ConvArgs / cv_epilogue / CV_* of csrc/vae_f32.hip.
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

