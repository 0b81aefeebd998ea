// Shared epilogue of the bf16 GEMM variants (gemm_bf16*.hip).  MFMA roles are swapped, so lane
// (l31, g) owns token row m = m_wave + 32*j + l31 and, per 32-feature block i and quad rq, the four
// consecutive features n = n_wave + 32*i + 8*rq + 4*g .. +3 (accumulator registers 4*rq .. 4*rq+3).
//
// Everything a lane needs besides its accumulators is LOADED IN BATCHES: bias and gate for all NI*4
// column groups once, and for the residual epilogue the NI*4 float4 of x of a token block before any of
// them is used.  The first version read-modified-wrote one float4 at a time; hipcc must assume the
// store may alias the next load, so every one of the 64 loads per lane (4-wave kernel) was followed by
// `s_waitcnt vmcnt(0)`: ~100 us of serialized HBM round trips per tile with the matrix pipes idle
// (the residual GEMM ran 13 % below the same shape with a plain bf16 store).
#pragma once
#include "common.h"
#include "../../include/moviigen_hip.h"

template <int EPI, int NI, int NJ>
MG_DEV void mg_gemm_epilogue(const f32x16_t (&acc)[NI][NJ], int64_t m_wave, int n_wave, int l31, int g, int64_t M, int N,
                             const float* __restrict__ bias, const float* __restrict__ gate, void* __restrict__ out,
                             int64_t ldo) {
    constexpr int NC = NI * 4;                     // column groups of 4 features per lane
    int ncol[NC];
    float4 b4[NC], g4[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int n = n_wave + (c >> 2) * 32 + (c & 3) * 8 + g * 4;
        ncol[c] = n;
        b4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        g4[c] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (n + 3 < N) {
            if (bias) b4[c] = *(const float4*)(bias + n);
            if (EPI == MG_EPI_GATE_RESID_F32 && gate) g4[c] = *(const float4*)(gate + n);
        } else {
            float* bb = (float*)&b4[c];
            float* gg = (float*)&g4[c];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < N) {
                    if (bias) bb[e] = bias[n + e];
                    if (EPI == MG_EPI_GATE_RESID_F32 && gate) gg[e] = gate[n + e];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int64_t m = m_wave + j * 32 + l31;
        if (m >= M) continue;
        float4 x4[NC];
        if (EPI == MG_EPI_GATE_RESID_F32) {        // all residual loads of this token block first
            const float* xr = (const float*)out + m * ldo;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                x4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ncol[c] + 3 < N) x4[c] = *(const float4*)(xr + ncol[c]);
                else {
                    float* xx = (float*)&x4[c];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ncol[c] + e < N) xx[e] = xr[ncol[c] + e];
                }
            }
            // the loads stay a batch: without this fence hipcc sinks each one to its use, behind the previous column
            // group's store (which may alias it for all it knows) — load, vmcnt(0), store, 64 times per lane
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int n = ncol[c];
            if (n >= N) continue;
            const int i = c >> 2, rq = c & 3;
            float v[4] = {acc[i][j][rq * 4 + 0] + b4[c].x, acc[i][j][rq * 4 + 1] + b4[c].y, acc[i][j][rq * 4 + 2] + b4[c].z,
                          acc[i][j][rq * 4 + 3] + b4[c].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);          // the reference's bf16 Linear output
            if (EPI == MG_EPI_BIAS_GELU_BF16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            }
            const bool full = n + 3 < N;
            if (EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16) {
                uint16_t* o = (uint16_t*)out + m * ldo + n;
                if (full) {
                    uint2 p;
                    p.x = pack_bf2(v[0], v[1]);
                    p.y = pack_bf2(v[2], v[3]);
                    *(uint2*)o = p;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) o[e] = f2bf(v[e]);
                }
            } else {
                float* o = (float*)out + m * ldo + n;
                float4 r = make_float4(v[0], v[1], v[2], v[3]);
                if (EPI == MG_EPI_GATE_RESID_F32) {
                    // torch evaluates `x + y * e` as a rounded product and a rounded sum (two kernels): no fma here
#pragma clang fp contract(off)
                    r.x = x4[c].x + v[0] * g4[c].x;
                    r.y = x4[c].y + v[1] * g4[c].y;
                    r.z = x4[c].z + v[2] * g4[c].z;
                    r.w = x4[c].w + v[3] * g4[c].w;
                }
                if (full) *(float4*)o = r;
                else {
                    const float* rr = (const float*)&r;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) o[e] = rr[e];
                }
            }
        }
    }
}


// The same epilogue for the 16x16x32 kernels (gemm_bf16_v7.hip): accumulator block [i][j] is 16 features x 16 tokens,
// lane (r16, G) owns token row m = m_wave + 16*j + r16 and the four consecutive features n = n_wave + 16*i + 4*G .. +3
// of every feature block (registers 0..3).  Same arithmetic, same batching of the bias / gate / residual loads.
// FULL / FULLM (wave-uniform: all NI*16 feature columns / all NJ*16 token rows of the wave exist) compile the guards
// away — with them in, hipcc turned the residual epilogue into ~1000 basic blocks of one guarded load each (the GEMM
// ran 6x slower than its own k-loop).
template <int EPI, int NI, int NJ, bool FULL, bool FULLM>
MG_DEV void mg_gemm_epilogue16_impl(const f32x4_t (&acc)[NI][NJ], int64_t m_wave, int n_wave, int r16, int G, int64_t M, int N,
                                    const float* __restrict__ bias, const float* __restrict__ gate, void* __restrict__ out,
                                    int64_t ldo) {
    // feature block outer, token block inner: bias / gate of ONE column group are live at a time (8 registers instead
    // of 64 — the 512-register kernels have none to spare here: with all column groups resident hipcc reused one
    // register quad for several residual loads and put a vmcnt(0) between them), and the NJ residual loads of a column
    // group (NJ different token rows) are issued as a batch before any of them is used.
#pragma unroll
    for (int c = 0; c < NI; ++c) {
        const int n = n_wave + c * 16 + G * 4;
        if (!FULL && n >= N) continue;
        const bool full = FULL || n + 3 < N;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (full) {
            if (bias) b4 = *(const float4*)(bias + n);
            if (EPI == MG_EPI_GATE_RESID_F32 && gate) g4 = *(const float4*)(gate + n);
        } else {
            float* bb = (float*)&b4;
            float* gg = (float*)&g4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < N) {
                    if (bias) bb[e] = bias[n + e];
                    if (EPI == MG_EPI_GATE_RESID_F32 && gate) gg[e] = gate[n + e];
                }
        }
        f32x4_t x4[NJ];
        if (EPI == MG_EPI_GATE_RESID_F32) {        // all residual loads of this column group first
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t m = m_wave + j * 16 + r16;
                x4[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (FULLM || m < M) {
                    const float* xr = (const float*)out + m * ldo + n;
                    if (full) x4[j] = *(const f32x4_t*)xr;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < N) x4[j][e] = xr[e];
                    }
                }
            }
            // the loads stay a BATCH: every loaded value passes through one opaque statement, so no use can be hoisted
            // between the loads (hipcc otherwise issues load, vmcnt(0), use, load, ... — one memory round trip per
            // 16 bytes — or sinks each load behind the previous store, which may alias it for all it knows)
            static_assert(NJ == 8, "the fence below names 8 registers quads");
            asm volatile("" : "+v"(x4[0]), "+v"(x4[1]), "+v"(x4[2]), "+v"(x4[3]), "+v"(x4[4]), "+v"(x4[5]), "+v"(x4[6]), "+v"(x4[7])
                         :: "memory");
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t m = m_wave + j * 16 + r16;
            if (!FULLM && m >= M) continue;
            float v[4] = {acc[c][j][0] + b4.x, acc[c][j][1] + b4.y, acc[c][j][2] + b4.z, acc[c][j][3] + b4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = round_bf(v[e]);          // the reference's bf16 Linear output
            if (EPI == MG_EPI_BIAS_GELU_BF16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
            }
            if (EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16) {
                uint16_t* o = (uint16_t*)out + m * ldo + n;
                if (full) {
                    uint2 p;
                    p.x = pack_bf2(v[0], v[1]);
                    p.y = pack_bf2(v[2], v[3]);
                    *(uint2*)o = p;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) o[e] = f2bf(v[e]);
                }
            } else {
                float* o = (float*)out + m * ldo + n;
                float4 r = make_float4(v[0], v[1], v[2], v[3]);
                if (EPI == MG_EPI_GATE_RESID_F32) {
                    // torch evaluates `x + y * e` as a rounded product and a rounded sum (two kernels): no fma here
#pragma clang fp contract(off)
                    r.x = x4[j][0] + v[0] * g4.x;
                    r.y = x4[j][1] + v[1] * g4.y;
                    r.z = x4[j][2] + v[2] * g4.z;
                    r.w = x4[j][3] + v[3] * g4.w;
                }
                if (full) *(float4*)o = r;
                else {
                    const float* rr = (const float*)&r;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) o[e] = rr[e];
                }
            }
        }
    }
}

template <int EPI, int NI, int NJ>
MG_DEV void mg_gemm_epilogue16(const f32x4_t (&acc)[NI][NJ], int64_t m_wave, int n_wave, int r16, int G, int64_t M, int N,
                               const float* __restrict__ bias, const float* __restrict__ gate, void* __restrict__ out,
                               int64_t ldo) {
    // wave-uniform fast path: every feature column and every token row of the wave's 128 x 128 block exists -> no
    // guards at all (a guard per token block splits the stores into basic blocks with a vmcnt(0) — which also waits
    // for the previous STORE — in front of each)
    if (n_wave + NI * 16 <= N && m_wave + NJ * 16 <= M)
        mg_gemm_epilogue16_impl<EPI, NI, NJ, true, true>(acc, m_wave, n_wave, r16, G, M, N, bias, gate, out, ldo);
    else if (n_wave + NI * 16 <= N)
        mg_gemm_epilogue16_impl<EPI, NI, NJ, true, false>(acc, m_wave, n_wave, r16, G, M, N, bias, gate, out, ldo);
    else
        mg_gemm_epilogue16_impl<EPI, NI, NJ, false, false>(acc, m_wave, n_wave, r16, G, M, N, bias, gate, out, ldo);
}
