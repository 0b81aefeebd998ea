// Shared device helpers of GEMM variants 11 and 12 (gemm_bf16_v11.hip, gemm_bf16_v12.hip): tile constants, the inline-assembly
// primitives (fragment read, counted wait, accumulator-pinned MFMA, LDS-DMA buffer load, M0 write) and the two epilogues shaped for the
// CU's memory pipe (16-byte bf16 stores; fp32 rows through LDS).  Arithmetic = gemm_epilogue.h's, element for element.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"
#include "../../include/moviigen_hip.h"

#define V11_BM 256
#define V11_BN 256
#define V11_BK 64
#define V11_THREADS 256
#define V11_A_BYTES (V11_BM * V11_BK * 2)  // 32 KiB
#define V11_W_BYTES (V11_BN * V11_BK * 2)  // 32 KiB
#define V11_STAGE (V11_A_BYTES + V11_W_BYTES)

typedef __attribute__((address_space(3))) void* v11_lptr_t;

template <int OFF>
MG_DEV void v11_rd(bf16x8_t& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
MG_DEV void v11_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
// acc += a . b with the accumulator PINNED to the accumulation register file ("+a"): given the choice, hipcc kept 15 of the 64
// accumulator quads of this kernel in arch VGPRs and copied each into an AGPR quad in front of its MFMA (60 v_accvgpr_write and 43
// s_nop per k-tile)
MG_DEV void v11_mfma(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// a raw buffer over [base, base + 2 GiB) as four SGPR words (base, stride 0, num_records, format): range checking plays no role
// (rows are clamped in the offsets), the resource only carries the 48-bit base
MG_DEV u32x4_t v11_rsrc(const void* base) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    return (u32x4_t){(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu)), 0x7fffffffu, 0x00020000u};
}
// one LDS-DMA load: 64 lanes x 16 bytes from rsrc.base + voff (per lane) + soff (scalar) to LDS at M0 + 16 * lane; M0 is written by
// v11_set_m0 one MFMA earlier (no s_nop between the M0 write and the load)
MG_DEV void v11_dma(int voff, const u32x4_t& rsrc, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int IMM>
MG_DEV void v11_set_m0(unsigned lds_base) {
    asm volatile("s_add_u32 m0, %0, %1" ::"s"(lds_base), "n"(IMM) : "scc", "memory");
}

// ---- epilogue of the bf16 outputs: 16-byte stores --------------------------------------------------------------------------------
// The MFMA leaves lane (G, r16) with FOUR consecutive features of token r16 per 16-feature block: 8 bytes of bf16 per store and
// lane, 64 stores per wave and tile, and the stores cost 16 k of a tile's 238 k cycles at N = K = 5120 (s_memtime with and without
// them, profiles/r04p_gemm_v11.log; skewing the workgroups so that their epilogues do not coincide changed nothing, it is not an HBM
// burst; a transposition through LDS to whole 256-byte row segments was SLOWER, profiles/r04q_gemm_v11.log).  The vendor library's
// kernel of the same tile stores 16 bytes per lane (its disassembly: 32 buffer_store_dwordx4 per wave).  Same here, at no cost in
// the k-loop: the W rows of a tile are staged in LDS in a PERMUTED order (v11_feature_of_row, only the per-piece source offsets of
// the LDS-DMA loads change), so that feature blocks 2p and 2p+1 of a lane are the two halves of EIGHT consecutive features
// 32p + 8G .. + 7.  Same arithmetic as mg_gemm_epilogue16 (gemm_epilogue.h), element for element.
MG_DEV int v11_feature_of_row(int row) {       // LDS row 16i + 4g + e of the W tile holds feature 32 (i >> 1) + 8g + 4 (i & 1) + e
    return (row & ~31) | ((row & 12) << 1) | ((row & 16) >> 2) | (row & 3);
}
template <int EPI, bool FULL, bool FULLM, bool NT = false>
MG_DEV void v11_epilogue_pair_impl(const f32x4_t (&acc)[8][8], int64_t m_wave, int n_wave, int r16, int G, int64_t M, int N,
                                   const float* __restrict__ bias, void* __restrict__ out, int64_t ldo) {
    static_assert(EPI == MG_EPI_BIAS_BF16 || EPI == MG_EPI_BIAS_GELU_BF16, "bf16 outputs only");
    // the four bias octets of the lane first, as ONE batch of loads: inside the loop below each of them was a separate round trip to the L2
    // with the matrix pipe idle (4 x ~0.5 us per tile)
    // (one opaque asm statement consumes all eight vectors: without it hipcc sinks each load to its first use, gemm_epilogue.h)
    f32x4_t b4a[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n_wave + p * 32 + G * 8;
        b4a[p][0] = b4a[p][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (bias && (FULL || n < N)) {
            if (FULL || n + 7 < N) {
                b4a[p][0] = *(const f32x4_t*)(bias + n);
                b4a[p][1] = *(const f32x4_t*)(bias + n + 4);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < N) b4a[p][e >> 2][e & 3] = bias[n + e];
            }
        }
    }
    asm volatile("" : "+v"(b4a[0][0]), "+v"(b4a[0][1]), "+v"(b4a[1][0]), "+v"(b4a[1][1]), "+v"(b4a[2][0]), "+v"(b4a[2][1]), "+v"(b4a[3][0]), "+v"(b4a[3][1]));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n_wave + p * 32 + G * 8;
        if (!FULL && n >= N) continue;
        const bool full = FULL || n + 7 < N;
        const float b8[8] = {b4a[p][0][0], b4a[p][0][1], b4a[p][0][2], b4a[p][0][3], b4a[p][1][0], b4a[p][1][1], b4a[p][1][2], b4a[p][1][3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t m = m_wave + j * 16 + r16;
            if (!FULLM && m >= M) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[2 * p][j][e] + b8[e], v[4 + e] = acc[2 * p + 1][j][e] + b8[4 + e];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = round_bf(v[e]);          // the reference's bf16 Linear output
            if (EPI == MG_EPI_BIAS_GELU_BF16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
            }
            uint16_t* o = (uint16_t*)out + m * ldo + n;
            if (full) {
                uint4 q;
                q.x = pack_bf2(v[0], v[1]);
                q.y = pack_bf2(v[2], v[3]);
                q.z = pack_bf2(v[4], v[5]);
                q.w = pack_bf2(v[6], v[7]);
                if (NT) __builtin_nontemporal_store((u32x4_t){q.x, q.y, q.z, q.w}, (u32x4_t*)o);      // the non-temporal hint (variant 12; variant 11 stores plainly)
                else *(uint4*)o = q;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < N) o[e] = f2bf(v[e]);
            }
        }
    }
}
template <int EPI, bool NT = false>
MG_DEV void v11_epilogue_pair(const f32x4_t (&acc)[8][8], int64_t m_wave, int n_wave, int r16, int G, int64_t M, int N,
                              const float* __restrict__ bias, void* __restrict__ out, int64_t ldo) {
    // wave-uniform fast paths, as in mg_gemm_epilogue16
    if (n_wave + 128 <= N && m_wave + 128 <= M)
        v11_epilogue_pair_impl<EPI, true, true, NT>(acc, m_wave, n_wave, r16, G, M, N, bias, out, ldo);
    else if (n_wave + 128 <= N)
        v11_epilogue_pair_impl<EPI, true, false>(acc, m_wave, n_wave, r16, G, M, N, bias, out, ldo);
    else
        v11_epilogue_pair_impl<EPI, false, false>(acc, m_wave, n_wave, r16, G, M, N, bias, out, ldo);
}

// ---- epilogue of the fp32 outputs: whole row segments through LDS ---------------------------------------------------------------
// experiments/store_probe.hip (profiles/r04r_store_probe.log), one CU writing the GEMM's own output tiles: the memory pipe takes ONE
// REQUEST PER LANE when adjacent lanes are different rows — the MFMA's layout, 16 rows x 64 bytes per instruction: 7.5 us per
// 256 x 256 fp32 tile, 13.6 us with the residual read in front — and 64 bytes per clock when a wave covers whole row segments
// (2 rows x 512 bytes per instruction: 1.9 us, 5.4 us with the residual).  The gated-residual epilogue is the one of the three largest
// GEMMs of a block.  So the wave transposes its 128 x 128 block through the LDS stage the k-loop has just released, two passes of 64
// tokens x 128 features of bf16 (16 KiB per wave; the Linear output is rounded to bf16 BEFORE gate and residual in the reference, so
// the staging is exact), and reads it back row-wise:
//   write:  lane (G, r16), feature block c, token block jj: 8 bytes at row jj*16 + r16, 16-byte chunk (2c + (G >> 1)) ^ r16, half G & 1
//           (a half-wave = 16 rows x 2 halves: every bank once);
//   read:   instruction q, lane l: row 2q + (l >> 5), features 4 (l & 31) .. + 3 -> residual dwordx4 load, dwordx4 store, 16 of them
//           in flight per wave; buffer addressing (block base in SGPRs, one 32-bit lane offset, the row in the scalar offset).
// Same arithmetic as mg_gemm_epilogue16 (gemm_epilogue.h), element for element.
typedef unsigned v11_u4 __attribute__((ext_vector_type(4)));
typedef unsigned v11_u2 __attribute__((ext_vector_type(2)));
// A hazard hipcc does not cover (found with `mg_selftest gemmdiff`, which bins the mismatches by position in the wave's block: ONE
// element of a dwordx4, in lanes 12-15 of every 16, only where the compiler had happened to put such an instruction next): behind a
// buffer_store_dwordx4 with a scalar offset REGISTER (LLVM's hazard table exempts that form) a PACKED-fp32 VALU instruction
// (v_pk_add_f32 / v_pk_mul_f32 of the next row) that overwrites the store's data registers gets the hi half of its result into the
// store data of the last four lanes of every row of 16.  The epilogue below therefore keeps the 16 results of a batch in DISTINCT
// registers until all 16 stores are issued (the asm statement behind them, which also holds the wait states behind the last one); the
// LDS writes — wide stores too, though no failure was traced to them — are inline assembly with a wait state behind them as a precaution.
// tools/audit_hot_loops.py scans every kernel of the library for the pattern.  (An inline-assembly STORE is no way out: the compiler then does not see that a VMEM instruction reads the scalar
// offset, and put the v_readlane that reloads a spilled offset directly in front of it — 5 wait states short.)
// two 8-byte LDS writes 4096 bytes apart (token blocks jj and jj + 1 of one feature block): `ds_write2st64_b64 ... offset0:OFF/512 offset1:OFF/512+8`
template <int OFF>
MG_DEV void v11_lds_write2(unsigned addr, const v11_u2& a, const v11_u2& b) {
    asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4\n\ts_nop 1" ::"v"(addr), "v"(a), "v"(b), "n"(OFF / 512), "n"(OFF / 512 + 8) : "memory");
}
template <int EPI>
MG_DEV void v11_epilogue_rows(const f32x4_t (&acc)[8][8], char* __restrict__ sp, int lane, int r16, int G, int64_t m_wave, int n_wave,
                              const float* __restrict__ bias, const float* __restrict__ gate, void* __restrict__ out, int64_t ldo) {
    static_assert(EPI == MG_EPI_GATE_RESID_F32 || EPI == MG_EPI_BIAS_F32, "fp32 outputs only");
    const int c8 = lane & 31, half = lane >> 5;
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (EPI == MG_EPI_GATE_RESID_F32 && gate) g4 = *(const float4*)(gate + n_wave + 4 * c8);
    float4 b4[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b4[c] = bias ? *(const float4*)(bias + n_wave + c * 16 + G * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((float*)out + m_wave * ldo + n_wave, 0, 0x7fffffff, 0x00020000);
    const int row_bytes = (int)ldo * 4;
    const int voff = half * row_bytes + c8 * 16;
    const unsigned wr = (unsigned)(uintptr_t)(v11_lptr_t)sp + r16 * 256 + (G & 1) * 8;
    char* const rd = sp + half * 256 + (c8 & 1) * 8;
    // The exchange is between the LANES of one wave: the hardware executes a wave's LDS instructions in order, but to the compiler
    // a lane's reads and another lane's writes are unrelated — without the compiler-level fences below it moved a read-back in front
    // of the last write of its pass.
#define V11_WAVE_LDS_FENCE asm volatile("" ::: "memory")
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        v11_u4 x4[16];
        if (EPI == MG_EPI_GATE_RESID_F32) {         // the first 32 residual rows of the pass: on their way during the transposition
#pragma unroll
            for (int u = 0; u < 16; ++u) x4[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (pass * 64 + u * 2) * row_bytes, 0);
        }
        V11_WAVE_LDS_FENCE;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chunk = ((2 * c + (G >> 1)) ^ r16) << 4;
            v11_u2 p[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = pass * 4 + jj;
                // the reference's bf16 Linear output
                p[jj].x = pack_bf2(acc[c][j][0] + b4[c].x, acc[c][j][1] + b4[c].y);
                p[jj].y = pack_bf2(acc[c][j][2] + b4[c].z, acc[c][j][3] + b4[c].w);
            }
            v11_lds_write2<0>(wr + chunk, p[0], p[1]);
            v11_lds_write2<8192>(wr + chunk, p[2], p[3]);
        }
        V11_WAVE_LDS_FENCE;
#pragma unroll
        for (int qb = 0; qb < 32; qb += 16) {
            v11_u2 d[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int row2 = (qb + u) * 2;                  // this lane's row: row2 + half
                if (EPI == MG_EPI_GATE_RESID_F32 && qb) x4[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (pass * 64 + row2) * row_bytes, 0);
                d[u] = *(const v11_u2*)(rd + row2 * 256 + (((c8 >> 1) ^ ((row2 & 15) | half)) << 4));
            }
            if (EPI == MG_EPI_GATE_RESID_F32)       // the residual loads stay a batch (gemm_epilogue.h)
                asm volatile("" : "+v"(x4[0]), "+v"(x4[1]), "+v"(x4[2]), "+v"(x4[3]), "+v"(x4[4]), "+v"(x4[5]), "+v"(x4[6]), "+v"(x4[7]),
                                  "+v"(x4[8]), "+v"(x4[9]), "+v"(x4[10]), "+v"(x4[11]), "+v"(x4[12]), "+v"(x4[13]), "+v"(x4[14]), "+v"(x4[15])
                             :: "memory");
            v11_u4 rr[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int row2 = (qb + u) * 2;
                const float v[4] = {__uint_as_float(d[u].x << 16), __uint_as_float(d[u].x & 0xffff0000u),
                                    __uint_as_float(d[u].y << 16), __uint_as_float(d[u].y & 0xffff0000u)};
                v11_u4 r = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                if (EPI == MG_EPI_GATE_RESID_F32) {
                    // torch evaluates `x + y * e` as a rounded product and a rounded sum (two kernels): no fma here
#pragma clang fp contract(off)
                    r.x = __float_as_uint(__uint_as_float(x4[u].x) + v[0] * g4.x);
                    r.y = __float_as_uint(__uint_as_float(x4[u].y) + v[1] * g4.y);
                    r.z = __float_as_uint(__uint_as_float(x4[u].z) + v[2] * g4.z);
                    r.w = __float_as_uint(__uint_as_float(x4[u].w) + v[3] * g4.w);
                }
                rr[u] = r;
                __builtin_amdgcn_raw_buffer_store_b128(rr[u], rs, voff, (pass * 64 + row2) * row_bytes, 0);
            }
            // no result register is reused before all 16 stores are out, and nothing overwrites one for two more cycles
            asm volatile("s_nop 1" ::"v"(rr[0]), "v"(rr[1]), "v"(rr[2]), "v"(rr[3]), "v"(rr[4]), "v"(rr[5]), "v"(rr[6]), "v"(rr[7]),
                         "v"(rr[8]), "v"(rr[9]), "v"(rr[10]), "v"(rr[11]), "v"(rr[12]), "v"(rr[13]), "v"(rr[14]), "v"(rr[15]));
        }
        V11_WAVE_LDS_FENCE;
    }
#undef V11_WAVE_LDS_FENCE
}

// The same epilogue with the residual loads as a ROLLING WINDOW (round 5; variant 12).  v11_epilogue_rows issues a batch of 16 row-segment loads
// and waits for it twice per pass with nothing else in flight — and its three 16-deep register arrays (loads, results, LDS read-back: 160 registers
// beside the kernel's state) spill; every scratch reload is a vmcnt(0), i.e. a full drain of whatever was in flight.  Of a tile's 28.9 us outside the
// k-loop (profiles/r05e_gemm_v12_tilecost.log) that is about four exposed round trips.  Here the tile's 128 row pairs are 8 groups of 8 instructions;
// two groups (16 loads, 16 KiB per wave) are always on their way: a group's results are computed IN PLACE in its load registers (8 distinct quads,
// kept until its 8 stores are issued: the store-data hazard above), then the registers take the loads of the group after next — across the pass
// boundary too.  80 registers instead of 160, no spill.  Same arithmetic, element for element.
// DBG (measurement builds only): 1 = no residual loads (x = 0; wrong results), 2 = no stores (wrong results), 4 = stores without the nt hint, 8 = residual
// loads without it.  (The tile of `out` is read once and written once: both carry the non-temporal hint — +0.5 % at K = 5120, +1.1 % at K = 13824,
// profiles/r05y3_gemm_epilogue_parts.log.)
template <int EPI, int DBG = 0>
MG_DEV void v11_epilogue_rows2(const f32x4_t (&acc)[8][8], char* __restrict__ sp, int lane, int r16, int G, int64_t m_wave, int n_wave,
                               const float* __restrict__ bias, const float* __restrict__ gate, void* __restrict__ out, int64_t ldo) {
    static_assert(EPI == MG_EPI_GATE_RESID_F32 || EPI == MG_EPI_BIAS_F32, "fp32 outputs only");
    constexpr bool RES = EPI == MG_EPI_GATE_RESID_F32;
    constexpr bool LOADS = RES && !(DBG & 1), STORES = !(DBG & 2);
    const int c8 = lane & 31, half = lane >> 5;
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (RES && gate) g4 = *(const float4*)(gate + n_wave + 4 * c8);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((float*)out + m_wave * ldo + n_wave, 0, 0x7fffffff, 0x00020000);
    const int row_bytes = (int)ldo * 4;
    const int voff = half * row_bytes + c8 * 16;
    const unsigned wr = (unsigned)(uintptr_t)(v11_lptr_t)sp + r16 * 256 + (G & 1) * 8;
    char* const rd = sp + half * 256 + (c8 & 1) * 8;
#define V11_FENCE asm volatile("" ::: "memory")      // compiler-level: LDS exchange between the lanes of a wave; and loads stay in front of it
    // the bias BEFORE the residual stream starts: a load behind it would be waited for with vmcnt counting every older load too
    f32x4_t b4[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b4[c] = bias ? *(const f32x4_t*)(bias + n_wave + c * 16 + G * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]), "+v"(b4[4]), "+v"(b4[5]), "+v"(b4[6]), "+v"(b4[7]) :: "memory");
    v11_u4 x[2][8];
    // group gid = 0..7: pass gid >> 2, row pairs 8 (gid & 3) .. + 7 of the pass (row pair q = rows 2q, 2q + 1 of the pass's 64 tokens)
    auto issue = [&](int gid) __attribute__((always_inline)) {
        if (LOADS) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                x[gid & 1][u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((gid >> 2) * 64 + ((gid & 3) * 8 + u) * 2) * row_bytes, (DBG & 8) ? 0 : 2);
        }
        V11_FENCE;
    };
    auto transpose = [&](int pass) __attribute__((always_inline)) {
        V11_FENCE;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int chunk = ((2 * c + (G >> 1)) ^ r16) << 4;
            v11_u2 p[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = pass * 4 + jj;
                p[jj].x = pack_bf2(acc[c][j][0] + b4[c][0], acc[c][j][1] + b4[c][1]);       // the reference's bf16 Linear output
                p[jj].y = pack_bf2(acc[c][j][2] + b4[c][2], acc[c][j][3] + b4[c][3]);
            }
            v11_lds_write2<0>(wr + chunk, p[0], p[1]);
            v11_lds_write2<8192>(wr + chunk, p[2], p[3]);
        }
        V11_FENCE;
    };
    auto group = [&](int gid) __attribute__((always_inline)) {
        v11_u4 (&xg)[8] = x[gid & 1];
        const int pass = gid >> 2, q0 = (gid & 3) * 8;
        v11_u2 d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row2 = (q0 + u) * 2;                  // this lane's row: row2 + half
            d[u] = *(const v11_u2*)(rd + row2 * 256 + (((c8 >> 1) ^ ((row2 & 15) | half)) << 4));
        }
        if (!LOADS && RES) {
#pragma unroll
            for (int u = 0; u < 8; ++u) xg[u] = (v11_u4){0u, 0u, 0u, 0u};
        }
        if (LOADS)     // this group's loads are waited for HERE, as one batch; the younger group stays in flight (the compiler counts vmcnt)
            asm volatile("" : "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3]), "+v"(xg[4]), "+v"(xg[5]), "+v"(xg[6]), "+v"(xg[7]) :: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row2 = (q0 + u) * 2;
            const float v[4] = {__uint_as_float(d[u].x << 16), __uint_as_float(d[u].x & 0xffff0000u),
                                __uint_as_float(d[u].y << 16), __uint_as_float(d[u].y & 0xffff0000u)};
            v11_u4 r = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            if (RES) {
                // torch evaluates `x + y * e` as a rounded product and a rounded sum (two kernels): no fma here
#pragma clang fp contract(off)
                r.x = __float_as_uint(__uint_as_float(xg[u].x) + v[0] * g4.x);
                r.y = __float_as_uint(__uint_as_float(xg[u].y) + v[1] * g4.y);
                r.z = __float_as_uint(__uint_as_float(xg[u].z) + v[2] * g4.z);
                r.w = __float_as_uint(__uint_as_float(xg[u].w) + v[3] * g4.w);
            }
            xg[u] = r;
            if (STORES) __builtin_amdgcn_raw_buffer_store_b128(xg[u], rs, voff, (pass * 64 + row2) * row_bytes, (DBG & 4) ? 0 : 2);
        }
        // no result register is reused before all 8 stores are out, and nothing overwrites one for two more cycles
        asm volatile("s_nop 1" ::"v"(xg[0]), "v"(xg[1]), "v"(xg[2]), "v"(xg[3]), "v"(xg[4]), "v"(xg[5]), "v"(xg[6]), "v"(xg[7]) : "memory");
    };
    issue(0);
    issue(1);
    transpose(0);
#pragma unroll
    for (int gid = 0; gid < 8; ++gid) {
        if (gid == 4) transpose(1);          // every read-back of pass 0 has been issued: a wave's LDS instructions execute in order
        group(gid);
        if (gid + 2 < 8) issue(gid + 2);
    }
    V11_FENCE;
#undef V11_FENCE
}
