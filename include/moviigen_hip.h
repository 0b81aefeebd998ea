/*
 * moviigen_hip.h — C-ABI of libmoviigen_hip.so, the MI355X (gfx950) kernels behind the
 * MoviiGen1.1 / Wan2.1-14B T2V denoising hot path.
 *
 * The reference (ZulutionAI/MoviiGen1.1) is pure Python + torch and has no FFI of its own;
 * every arithmetic step of its hot path is a call into a third-party native kernel
 * (cuBLAS / flash_attn / ATen elementwise / cuDNN).  Each entry point below names the
 * reference call site (file:line under the reference tree) whose arithmetic it replaces.
 * The Python side that binds these with ctypes is moviigen1.1_amd/wan/backend/lib.py; the
 * binding a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 (MG_OK) or a negative error code; nothing throws, nothing
 *     allocates, nothing synchronises: work is enqueued on `stream` (a hipStream_t passed as
 *     void*; NULL = the legacy default stream).
 *   - all pointers are DEVICE pointers owned by the caller.  bf16 travels as uint16_t.
 *   - `ld*` arguments are row strides in ELEMENTS.  Row-major everywhere.
 *   - no torch types, no C++ types in any signature.
 */
#ifndef MOVIIGEN_HIP_H
#define MOVIIGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_ARG (-1)    /* null pointer / bad enum */
#define MG_ERR_SHAPE (-2)  /* unsupported shape or alignment */
#define MG_ERR_LAUNCH (-3) /* hipGetLastError() != hipSuccess after launch */
#define MG_ERR_UNAVAILABLE (-4) /* librccl could not be bound at run time (collective entry points only) */
#define MG_ERR_COMM (-5) /* an RCCL call returned an error */

/* library identification: returns a static string "moviigen_hip <abi> gfx950" */
const char* mg_version(void);
/* ABI revision, bumped on any signature change */
int mg_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * DiT — token-wise HBM-bound kernels
 * ---------------------------------------------------------------------------------------- */

/* WanLayerNorm (eps, no affine) in fp32 followed by the AdaLN modulation, or by an affine
 * weight/bias (norm3).  Replaces wan/modules/model.py:89-99 + :299,:307 (norm1/norm2 +
 * `*(1+scale)+shift`), :306 (norm3, elementwise_affine) and :340-342 (head norm+modulate).
 *   y = (x-mean)*rsqrt(var+eps);  if round_norm_bf16: y = bf16(y)   [block 0: x is bf16 there,
 *   model.py:99 `.type_as(x)`];  out = add_one ? y*(1+scale)+shift : y*scale+shift
 *   scale/shift: fp32 [dim] or NULL (=> 1 / 0).  out: bf16 (out_f32=0) or fp32 (out_f32=1).
 * dim % 4 == 0, dim <= 8192. */
int mg_ln_modulate(const float* x, int64_t ldx, int64_t rows, int dim, const float* scale,
                   const float* shift, int add_one, float eps, int round_norm_bf16, void* out,
                   int out_f32, int64_t ldo, void* stream);

/* WanRMSNorm over the whole `dim` vector (fp32 math, result rounded to bf16, then * weight in
 * fp32) and optional 3-axis RoPE on adjacent pairs, output bf16.
 * Replaces wan/modules/model.py:70-86 (+:139-140, :168-169) and rope_apply model.py:39-67 /
 * wan/distributed/xdit_context_parallel.py:23-62 (rank slice = pos0).
 *   rope_cs: NULL (no RoPE: cross-attention q/k) or float2 (cos,sin) tables laid out as
 *            [F][c0] ++ [H][c1] ++ [W][c1] with c = head_dim/2, c1 = c/3, c0 = c - 2*c1
 *   token index of row r is pos0 + r, decomposed (f,h,w) row-major over the F*H*W grid;
 *   rows with token index >= F*H*W are passed through un-rotated (padding, model.py:61).
 *   out_scale: the fp32 result is multiplied by it before the (single) rounding to bf16; 1 = the reference value.
 *            WanModel.forward passes softmax_scale*log2(e) for q, so that mg_attn_fwd_bf16_hd128_prescaled needs no
 *            per-score multiply (flash_attn applies softmax_scale to the fp32 scores: same product, one rounding
 *            either way, taken at a different point).
 * dim % 8 == 0, dim <= 8192, head_dim % 2 == 0, dim % head_dim == 0, ldx/ldo % 8 == 0. */
int mg_rmsnorm_rope_bf16(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo,
                         int64_t rows, int dim, const float* weight, float eps, int head_dim,
                         const float* rope_cs, int F, int H, int W, int64_t pos0, float out_scale, void* stream);

/* Pack K and/or V (row-major [L][>=heads*128], row strides ldk/ldv; either may be NULL) into the
 * per-64-key-tile operand layout of mg_attn_fwd_bf16_hd128*, nt = ceil(L/64) tiles per head:
 *     kp[head][tile][c = d/8 (16)][row (64)][8]      key 32u + 8a + 4b + j of the tile sits in row 32u + 16b + 4a + j
 *     vp[head][tile][kc = (key%64)/8 (8)][d (128)][8 keys]
 * each tile = one contiguous 16 KiB block = the LDS image (pure LDS-DMA staging, conflict-free
 * ds_read_b128 with immediate offsets); keys >= L are zero.  The K row order is the one the 16x16x32 attention kernel
 * wants (its scores come out as the B operand of P.V without a shuffle); under the A/B library's mg_attn_set_variant(3) the rows are
 * in natural order (row = key % 64) for the A/B partner kernel.  kp/vp: heads*nt*8192 elements each,
 * 16-byte aligned.  No reference counterpart (flash_attn stages K/V inside its kernel). */
int mg_pack_kv_bf16(const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv, int64_t L,
                    int heads, int head_dim, uint16_t* kp, uint16_t* vp, void* stream);

/* ------------------------------------------------------------------------------------------
 * DiT — MFMA kernels
 * ---------------------------------------------------------------------------------------- */

enum {
    MG_EPI_BIAS_BF16 = 0,      /* out_bf16 = bf16(acc + bias)                         nn.Linear */
    MG_EPI_BIAS_GELU_BF16 = 1, /* out_bf16 = bf16(gelu_tanh(bf16(acc + bias)))        ffn.0+ffn.1 */
    MG_EPI_GATE_RESID_F32 = 2, /* out_f32 += bf16(acc + bias) * gate[n] (gate NULL=>1) x + y*e */
    MG_EPI_BIAS_F32 = 3        /* out_f32 = float(bf16(acc + bias))                   patch embed */
};

/* out[M][N] = A[M][K] . W[N][K]^T (+bias, epilogue) — bf16 operands, fp32 MFMA accumulate
 * (v_mfma_f32_32x32x16_bf16).  Replaces every nn.Linear executed under autocast(bf16) on the
 * path: wan/modules/model.py:139-141,155 (q,k,v,o), :168-170,180 (cross q,k,v,o), :267-269
 * (ffn), :451-453 (text_embedding), :445-450 (patch_embedding as a [L,64]x[64,dim] GEMM), with
 * the residual/gate updates of model.py:301-302,306,308-309 fused as epilogue 2.
 * K % 64 == 0, lda/ldw % 8 == 0, A/W 16-byte aligned; ldo % 4 == 0. */
int mg_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                 const float* bias, int64_t M, int N, int K, int epilogue, void* out,
                 int64_t ldo, const float* gate, void* stream);

/* softmax(q k^T * scale) v, non-causal, keys >= Lk masked; bf16 in/out, fp32 accumulate,
 * head_dim 128.  Replaces flash_attn_varlen_func as called from
 * wan/modules/attention.py:96-127 (self-attention model.py:146-151, k_lens=seq_lens;
 * cross-attention model.py:176, Lk = 512 unmasked).
 *   q [Lq][>=heads*128] row stride ldq, head h at column h*128; o likewise (row stride ldo)
 *   kp, vp: K and V of the Lk keys packed by mg_pack_kv_bf16(…, L = Lk, …).
 *   workspace: NULL, or mg_attn_workspace_bytes() device bytes OWNED BY THE CALLER, 8-byte aligned, zeroed once before
 *     their first use.  A persistent launch with >= 32 rounds of work per compute unit hands its (query block, head)
 *     items out by ticket through it (+1.35 % at the metric's launch) and leaves it zeroed when it ends, so launches
 *     that are ordered with respect to one another (one stream, or event-ordered) share one workspace; launches that may
 *     overlap need one each.  With NULL the launch uses the static per-XCD partition (same bits).  The library holds no
 *     device memory of its own and never allocates or synchronises inside a launch; a launch with a
 *     workspace is capturable in a hipGraph from its first call. */
int64_t mg_attn_workspace_bytes(void);
int mg_attn_fwd_bf16_hd128(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                           uint16_t* o, int64_t ldo, int64_t Lq, int64_t Lk, int heads, float scale,
                           void* workspace, void* stream);

/* Same contract for any head_dim <= 256 (head_dim % 8 == 0): the correctness path for model
 * sizes whose head_dim is not 128 (BASELINE.json configs[0]: head_dim 32).  v is NOT transposed:
 * v [Lk][>=heads*head_dim] row stride ldv. */
/* Same, plus the log-sum-exp of the scaled scores per (head, query): lse[head*Lq + q] fp32 (may be
 * NULL = mg_attn_fwd_bf16_hd128).  Ring attention merges per-block results with it. */
int mg_attn_fwd_bf16_hd128_lse(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                               uint16_t* o, int64_t ldo, float* lse, int64_t Lq, int64_t Lk, int heads,
                               float scale, void* workspace, void* stream);

/* Same operator for a q that ALREADY carries the factor scale*log2(e) (mg_rmsnorm_rope_bf16 with out_scale: the factor
 * enters before q's one rounding to bf16, so nothing is rounded twice) — the form WanModel.forward uses.  With the
 * default kernel a score then IS its base-2 exponent (no per-score multiply-add).  lse may be NULL; when given it is
 * the natural-log log-sum-exp of the true scaled scores, as above.
 * reserve_cus >= 0: compute units the (persistent, one-workgroup-per-CU) launch leaves free, rounded up to a multiple of
 * 8 = one per XCD — room for a kernel on another stream (RCCL's all-to-all of the next head group under sequence
 * parallelism: a persistent workgroup holds its CU's whole register file until the launch ends).  0 = all CUs. */
int mg_attn_fwd_bf16_hd128_prescaled(const uint16_t* q, int64_t ldq, const uint16_t* kp, const uint16_t* vp,
                                     uint16_t* o, int64_t ldo, float* lse, int64_t Lq, int64_t Lk, int heads,
                                     int reserve_cus, void* workspace, void* stream);

/* x[r][c] += float(y[r][c]) * gate[c]: the gated residual update of wan/modules/model.py:301-302,306,308-309 as a
 * stand-alone kernel (x fp32 [rows][dim] row stride ldx; y bf16 row stride ldy; gate fp32 [dim] or NULL = 1).
 * The fused form is MG_EPI_GATE_RESID_F32; this one serves a caller-replaced attention operator
 * (`types.MethodType(fn, block.self_attn)`, wan/text2video.py:97-100).  dim, ldx, ldy % 4 == 0. */
int mg_gate_residual_f32(float* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate, int64_t rows,
                         int dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ulysses sequence-parallel exchange layouts (the transposes xfuser / FastVideo's all_to_all_4D do
 * around their all-to-alls: wan/distributed/xdit_context_parallel.py:185-190,
 * scripts/train/model/model_seq.py:232-234,256).  The collective itself is issued by the host
 * (RCCL all_to_all_single on a communication stream); these two kernels produce / consume its
 * contiguous buffers so that ONE exchange carries q, k and v of a head group.
 * ---------------------------------------------------------------------------------------- */

/* send[p][t][i*w + c] = x_i[t][p*cols_per_dest + col0 + c]   (i = 0,1,2 for q,k,v; p < P; t < Lloc; c < w)
 *   q, k, v: [Lloc][>= P*cols_per_dest] bf16, row strides ldq/ldk/ldv (column slices of a fused buffer are fine)
 *   cols_per_dest = (heads / P) * head_dim; [col0, col0 + w) = the head group inside a destination's slice.
 * After all_to_all_single(recv, send) the receive buffer is the row-major [P*Lloc][3w] matrix of ALL tokens
 * (rank order) x (q | k | v) of this rank's heads of the group.  All widths/strides % 8 == 0, 16-byte aligned. */
int mg_sp_pack_qkv_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v,
                        int64_t ldv, int64_t Lloc, int P, int cols_per_dest, int col0, int w, uint16_t* send,
                        void* stream);

/* the way back: recv[p][t][c] (attention output of source rank p's heads for MY tokens) ->
 * o[t][p*cols_per_src + col0 + c], o row stride ldo. */
int mg_sp_unpack_o_bf16(const uint16_t* recv, int64_t Lloc, int P, int cols_per_src, int col0, int w, uint16_t* o,
                        int64_t ldo, void* stream);

/* the strided block copy both are built on (one-tensor seq<->head exchanges, all_to_all_4D's reshape):
 * dst[b*d_blk + r*d_row + c] = src[b*s_blk + r*s_row + c], b < blocks, r < rows, c < width; strides in elements,
 * everything % 8 == 0, 16-byte aligned. */
int mg_sp_copy_blocks_bf16(const uint16_t* src, int64_t s_blk, int64_t s_row, uint16_t* dst, int64_t d_blk,
                           int64_t d_row, int blocks, int64_t rows, int width, void* stream);

/* ------------------------------------------------------------------------------------------
 * Collectives on an RCCL communicator.  librccl is bound at run time (dlopen, preferring the
 * copy already loaded in the process); without it these return MG_ERR_UNAVAILABLE and nothing else is affected.
 * `comm` is an ncclComm_t.  Everything is enqueued on `stream`; no host synchronisation.
 * Replaces: xfuser get_sp_group().all_gather / xFuserLongContextAttention's all-to-alls
 * (wan/distributed/xdit_context_parallel.py:148,185-190), FastVideo all_to_all_4D / all_gather
 * (scripts/train/model/model_seq.py:232-234,256,780), torch FSDP's per-block parameter all-gather
 * (wan/distributed/fsdp.py:20-31).
 * ---------------------------------------------------------------------------------------- */

/* communicator bootstrap: rank 0 calls mg_comm_unique_id (128 bytes), hands the bytes to every rank by any side
 * channel (the host code uses torch.distributed's object broadcast), every rank calls mg_comm_create. */
int mg_comm_unique_id(void* id128);
int mg_comm_create(const void* id128, int nranks, int rank, void** comm);
int mg_comm_destroy(void* comm);

/* all-to-all of equal chunks: bytes [p*B, (p+1)*B) of `send` go to rank p, the same range of `recv` comes from rank p
 * (B = bytes_per_peer).  Grouped ncclSend/ncclRecv: on the xGMI mesh every pair uses its own link. */
int mg_sp_all_to_all(void* comm, const void* send, void* recv, int64_t bytes_per_peer, void* stream);

/* FastVideo's all_to_all_4D on one bf16 tensor, layout changes included:
 *   seq_to_head != 0 (scatter_dim=2, gather_dim=1): x [rows = L/P][heads*head_dim] (row stride ldx)
 *        -> out [L][(heads/P)*head_dim] contiguous (ldo must equal (heads/P)*head_dim), tokens in rank order;
 *   seq_to_head == 0 (scatter_dim=1, gather_dim=2): x [rows = L][(heads/P)*head_dim] contiguous
 *        -> out [L/P][heads*head_dim] (row stride ldo).
 * workspace: rows * (heads/P or heads... i.e. as many elements as x) bf16, 16-byte aligned. */
int mg_sp_all_to_all_4d_bf16(void* comm, const uint16_t* x, int64_t ldx, int64_t rows, int heads, int head_dim,
                             int seq_to_head, uint16_t* out, int64_t ldo, uint16_t* workspace, void* stream);

/* rank-order concatenation of `bytes` bytes per rank (ncclAllGather): the head-output gather of the SP forward and the
 * per-block parameter gather of the block-sharded (FSDP-style) weights. */
int mg_sp_all_gather(void* comm, const void* send, void* recv, int64_t bytes, void* stream);
int mg_shard_all_gather(void* comm, const void* shard, void* full, int64_t shard_bytes, void* stream);

/* Ring attention (the reference delegates to yunchang inside xFuserLongContextAttention,
 * generate.py:225-229): fold one block's normalised bf16 result `part` [Lq][heads*128] and its lse into
 * the running fp32 result: lse' = logaddexp(lse, lse_j), acc' = acc*exp(lse-lse') + part*exp(lse_j-lse').
 * first != 0 initialises acc/lse_acc from the block; out (bf16, may be NULL) receives acc'. */
int mg_attn_merge_f32(float* acc, int64_t lda, float* lse_acc, const uint16_t* part, int64_t ldp,
                      const float* lse_part, uint16_t* out, int64_t ldo, int64_t Lq, int heads, int first,
                      void* stream);

int mg_attn_fwd_bf16_generic(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                             const uint16_t* v, int64_t ldv, uint16_t* o, int64_t ldo, int64_t Lq,
                             int64_t Lk, int heads, int head_dim, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * DiT — small fp32 pieces (time embedding, head, patch gather, latent algebra)
 * ---------------------------------------------------------------------------------------- */

/* sinusoidal_embedding_1d, wan/modules/model.py:15-25: out[i][0:half]=cos(t_i*w_j),
 * out[i][half:]=sin(..), w_j = 10000^(-j/half), evaluated in fp64, stored fp32.
 * t_dtype: 0 = int64, 1 = float32, 2 = float64 (device pointer, n entries). */
int mg_sinusoid_embed(const void* t, int t_dtype, int n, int dim, float* out, void* stream);

/* y[N] = W[N][K] . f(x[K]) + bias,  f = SiLU if silu_in else identity; all fp32.
 * Replaces time_embedding / time_projection, wan/modules/model.py:455-457,541-545. */
int mg_gemv_f32(const float* W, const float* bias, const float* x, float* y, int N, int K,
                int silu_in, void* stream);

/* out[r][:] = a[r][:] + b[r % period][:]  (modulation + e0, model.py:292-295; head :340). */
int mg_add_rows_f32(const float* a, const float* b, float* out, int rows, int dim, int period,
                    void* stream);

/* fp32 out[M][N] = x[M][K] . W[N][K]^T + bias (Head.head Linear, model.py:342; N <= 64). */
int mg_head_gemm_f32(const float* x, int64_t ldx, const float* W, const float* bias, float* out,
                     int64_t M, int N, int K, void* stream);

/* latent [C][F][H][W] fp32 -> tokens [F*(H/ph)*(W/pw)][C*ph*pw] bf16 with k = (c,i,j), the
 * im2col of the k=s=(1,ph,pw) Conv3d at model.py:445-450,529-531. */
int mg_patchify_bf16(const float* lat, int C, int F, int H, int W, int ph, int pw,
                     uint16_t* out, int64_t ldo, void* stream);

/* tokens [F*Hg*Wg][ph*pw*C] fp32 (channel fastest) -> latent [C][F][Hg*ph][Wg*pw];
 * WanModel.unpatchify, model.py:581-609. */
int mg_unpatchify_f32(const float* tok, int64_t ldt, int C, int F, int Hg, int Wg, int ph, int pw,
                      float* lat, void* stream);

/* out[i] = c0*x0[i] + c1*x1[i] + c2*x2[i] + c3*x3[i]  (NULL terms skipped).  CFG combine
 * (wan/text2video.py:245-246) and every UniPC / DPM-Solver++ update
 * (wan/utils/fm_solvers_unipc.py:315-332,455-485,600-627; fm_solvers.py:461-470,533-541) are
 * linear combinations with host-side scalar coefficients. */
int mg_lincomb4_f32(float* out, int64_t n, const float* x0, float c0, const float* x1, float c1,
                    const float* x2, float c2, const float* x3, float c3, void* stream);

/* Classifier-free guidance in the reference's operation order: out = uncond + g*(cond - uncond)
 * (wan/text2video.py:245-246). */
int mg_cfg_combine_f32(float* out, const float* uncond, const float* cond, float guide_scale,
                       int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * umT5 text encoder glue (the GEMMs are mg_gemm_bf16, the T5LayerNorm is
 * mg_rmsnorm_rope_bf16 without RoPE).  Reference wan/modules/t5.py, model in bf16.
 * ---------------------------------------------------------------------------------------- */

/* token_embedding lookup, t5.py:273,289: out[i][:] = table[clamp(ids[i])][:]; dim % 8 == 0. */
int mg_embed_rows_bf16(const uint16_t* table, int64_t vocab, int dim, const int64_t* ids, int n,
                       uint16_t* out, void* stream);

/* bf16 elementwise, n % 8 == 0: mode 0 out = a + b (residual add of bf16 tensors, t5.py:165-166);
 * mode 1 out = a * gelu_tanh(b) (T5FeedForward fc1(x) * gate(x), t5.py:40-44,135). */
int mg_ew_bf16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, int mode, void* stream);

/* T5Attention, t5.py:82-113: softmax(q k^T + emb[bucket(j-i)][head]) v, NO 1/sqrt(d) scaling,
 * keys >= Lk masked.  q,k,v [L][heads*head_dim] (row stride ld), rel_emb [num_buckets][heads] bf16,
 * rel_bucket[(j - i) + Lq - 1] (int32, Lq + Lk - 1 entries) = T5RelativeEmbedding bucket of the
 * relative position (t5.py:242-263), tabulated by the host.  head_dim <= 128. */
int mg_t5_attn_bf16(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld,
                    const uint16_t* rel_emb, const int* rel_bucket, uint16_t* o, int64_t ldo,
                    int64_t Lq, int64_t Lk, int heads, int head_dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * WanVAE decode (fp32, channels-last activations [T][H][W][C])
 * ---------------------------------------------------------------------------------------- */

/* Generic causal conv as an implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32).
 * x: [T][H][W][Cin] channels-last;  w: [Cout][kt][kh][kw][Cin];  out: [T][Ho][Wo][Cout].
 * Temporal taps reach back kt-1 frames: frame index t-(kt-1)+dt; frames < 0 come from
 * `cache` ([tc][H][W][Cin], the last tc <= kt-1 frames of the previous chunk, may be NULL) and
 * are zero before that (CausalConv3d, wan/modules/vae.py:17-36).  Spatial zero padding
 * (kh/2, kw/2).  `up2`: the input is read through a nearest-exact 2x upsample (Ho=2H, Wo=2W;
 * Resample upsample2d/3d, vae.py:66-83,138-141).  residual (same shape as out, may be NULL) is
 * added (ResidualBlock `x + h`, vae.py:220).  kt <= 3, kh and kw in {1, 3} (what WanVAE uses); larger extents return
 * MG_ERR_SHAPE. */
int mg_vae_conv_f32(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                    const float* w, const float* bias, int Cout, int kt, int kh, int kw, int up2,
                    const float* residual, float* out, int mode, void* stream);

/* The 3x3 conv behind a nearest-exact 2x upsample (Resample upsample2d/3d, vae.py:66-83,138-141) as FOUR 2x2 convs of the
 * image itself, one per output parity (py, px): out[t][2y+py][2x+px] reads image rows {y-1, y} (py = 0) or {y, y+1}
 * (py = 1), likewise along x — the 3x3 taps that land on the same image pixel are summed into one weight beforehand.
 * 4/9 of the multiply-adds of mg_vae_conv_f32(up2 = 1) and no per-tap index arithmetic; results agree with it to fp32
 * rounding of the weight sums (not bit for bit).  x [T][H][W][Cin], out [T][2H][2W][Cout], zero padding as the reference's.
 * mg_vae_upconv_fold_weights_f32: w [Cout][1][3][3][Cin] -> wp [4 = 2 py + px][Cout][2][2][Cin] (once per checkpoint). */
int mg_vae_upconv_fold_weights_f32(const float* w, int Cout, int Cin, float* wp, void* stream);
int mg_vae_upconv_phases_f32(const float* x, int T, int H, int W, int Cin, const float* wp, const float* bias, int Cout,
                             float* out, int mode, void* stream);

/* Column-window forms (ABI 9) — one rank's W band of a decode split along W over several GPUs (no reference counterpart: the
 * reference decodes on rank 0 alone, wan/text2video.py:260-261; SURVEY.md 8(e) names the spatial split with one halo pixel per
 * convolution as the natural sharding).  x and cache are the band WITH the neighbours' halo columns: [.][H][W][Cin] where W counts
 * the halos; the launch computes the output columns [col0, col0 + cols) only and writes them compactly — out / residual
 * [T][H][cols][Cout] (phases: out [T][2H][2 cols][Cout]); the zero padding of the reference begins outside [0, W) of x, i.e. at
 * the true image border of an edge rank.  Every output voxel is the same dot product in the same order as in the full launch: the
 * bands of P ranks, side by side, are bit-identical to mg_vae_conv_f32 / mg_vae_upconv_phases_f32 on the whole image. */
int mg_vae_conv_cols_f32(const float* x, const float* cache, int tc, int T, int H, int W, int Cin,
                         const float* w, const float* bias, int Cout, int kt, int kh, int kw,
                         const float* residual, float* out, int col0, int cols, int mode, void* stream);
int mg_vae_upconv_phases_cols_f32(const float* x, int T, int H, int W, int Cin, const float* wp, const float* bias, int Cout,
                                  float* out, int col0, int cols, int mode, void* stream);

/* `mode` of mg_vae_conv_f32 / mg_vae_upconv_phases_f32 — the arithmetic of THAT call (ABI 7: an argument, not a
 * process-global switch; anything else returns MG_ERR_ARG):
 *   MG_VAE_EXACT  = v_mfma_f32_32x32x2_f32, bitwise an fmaf chain — the reference's fp32 arithmetic (vae.py:623,658);
 *   MG_VAE_BF16X3 = split bf16 x 3: every fp32 operand as hi + lo bf16 (16 mantissa bits), W_hi.X_hi + W_hi.X_lo + W_lo.X_hi
 *       on the bf16 MFMA with fp32 accumulation: ~1e-5 relative to the exact mode per convolution, NOT the reference's
 *       arithmetic, opt-in (WanVAE(mode='bf16x3')) and never what bench.py measures.  mg_vae_attn_f32 and convolutions
 *       with Cout <= 4 (the decoder head, on v_mfma_f32_4x4x1_16B_f32) are exact in either mode. */
#define MG_VAE_EXACT 0
#define MG_VAE_BF16X3 1
/* bits 8-9 of `mode` (measurement override, same bits out): the voxel tile of the wide exact convolutions — 0 or
 * MG_VAE_TILE_128 = 128 voxels per workgroup (what the library runs), MG_VAE_TILE_256 = 256 (one workgroup per CU: measured
 * 7 % slower on the 1920x832x81f decode, kept for A/B runs). */
#define MG_VAE_TILE_128 (1 << 8)
#define MG_VAE_TILE_256 (2 << 8)

/* RMS_norm over channels (F.normalize(x, dim=C) * sqrt(C) * gamma, vae.py:39-54), optional SiLU
 * (vae.py:193-197, 466-468).  x,out [rows][C] channels-last. */
int mg_vae_rmsnorm_silu_f32(const float* x, const float* gamma, float* out, int64_t rows, int C,
                            int do_silu, void* stream);

/* Single-head attention over one frame's h*w tokens with head dim C (<=512, %32==0), fp32:
 * AttentionBlock, vae.py:247-256 (scaled_dot_product_attention, scale 1/sqrt(C)).
 * qkv [frames][L][3C] (q|k|v), out [frames][L][C]; workspace: caller-owned, mg_vae_attn_workspace_floats(L, C)
 * floats = (min(L, 2048) + C) * roundup(L, 4) — one block of 2048 query rows of the score matrix, which stays
 * inside the Infinity Cache between the three passes, plus V^T — 16-byte aligned (the library never allocates). */
int64_t mg_vae_attn_workspace_floats(int64_t L, int C);
int mg_vae_attn_f32(const float* qkv, float* out, int frames, int64_t L, int C, float* workspace,
                    void* stream);
/* The same attention for Lq <= Lk query rows of a frame against all Lk keys / values held elsewhere (ABI 9; the W-band decode: q =
 * the band's own pixels, k / v = the two halves of the all-gathered k|v tensor): q [frames][Lq][..] row stride ldq, k / v
 * [frames][Lk][..] row stride ldkv, out [frames][Lq][C]; workspace mg_vae_attn_workspace_floats(Lk, C).  A row's bits do not
 * depend on which other rows are computed with it. */
int mg_vae_attn_rows_f32(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, float* out, int frames,
                         int64_t Lq, int64_t Lk, int C, float* workspace, void* stream);

/* z[C][T][H][W] (NCTHW, reference layout) -> channels-last with the latent un-normalisation
 * z/scale1[c] + scale0[c] of vae.py:546-551; and the inverse layout change with clamp(-1,1) for
 * the decoded video (vae.py:661). */
int mg_vae_latent_in_f32(const float* z, const float* mean, const float* inv_std, int C, int T,
                         int H, int W, float* out, void* stream);
int mg_vae_video_out_f32(const float* x, int C, int T, int H, int W, float* out, int t_off,
                         int T_total, void* stream);

/* Decoded video [3][T][H][W] fp32 -> uint8 frames [T][H][W][3] as the reference's cache_video writes
 * them (wan/utils/utils.py:39-47): clamp(lo,hi), (x-lo)/max(hi-lo,1e-5), *255, truncating cast. */
int mg_video_to_u8(const float* video, int T, int H, int W, float lo, float hi, uint8_t* frames,
                   void* stream);

/* One decoded frame [3][H][W] fp32 -> uint8 pixels [H][W][3] as the reference's cache_image writes a t2i result
 * (wan/utils/utils.py:64-91 -> torchvision save_image): clamp(lo,hi), (x-lo)/max(hi-lo,1e-5), *255, +0.5,
 * clamp(0,255), truncating cast. */
int mg_image_to_u8(const float* image, int H, int W, float lo, float hi, uint8_t* pixels, void* stream);

/* time_conv channel halves -> interleaved frames (vae.py:133-137):
 * x [T][H][W][2C] -> out [2T][H][W][C], frame 2t from channels [0,C), 2t+1 from [C,2C). */
int mg_vae_time_interleave_f32(const float* x, int T, int64_t HW, int C, float* out, void* stream);

#ifdef MG_AB_BUILD
/* ------------------------------------------------------------------------------------------
 * A/B LIBRARY ONLY (libmoviigen_hip_ab.so, built from the same sources with -DMG_AB_BUILD; the product library
 * libmoviigen_hip.so exports NOTHING of this section).  Measurement partners and s_memtime hooks: used by
 * csrc/tools/selftest.cpp, tools/ and the variant-agreement tests to force a kernel schedule and to take cycle breakdowns
 * of the hot loops.  All are process-global switches, not thread-safe, not part of the drop-in contract; passing 0 / NULL
 * restores the default.  The product path never loads this library (wan/backend/lib.py: load vs load_ab).
 * ---------------------------------------------------------------------------------------- */
/* Tile schedule of mg_gemm_bf16 for M > 256 and N > 128 (same results bit for bit); 0 = the product's rule (12, else 2 / 1).
 * Returns MG_ERR_ARG for a number that names no variant (no silent aliases).
 * 12 = (product) 256x256x64 tile, v_mfma_f32_16x16x32_bf16, persistent, 4 waves = ONE per SIMD; a stage is refilled WHILE it is
 *     consumed — loads two k-tiles ahead, three barriers per k-tile, none behind a drained memory pipe; the k-tile body is a GENERATED
 *     schedule (tools/gen_gemm_v12_schedule.py) in a single self-looping block (200 + flags: 2 raster 0, 4 no stores, 16 direct fp32
 *     epilogue, 32 * (1 + s) generated body s);
 * 11 = the round-4 default: the same tile and epilogues, ONE barrier per k-tile behind vmcnt(0), the last 32 MFMAs of a k-tile behind
 *     the next barrier (tools/gen_gemm_v11_schedule.py; 110 + flags: 1 de-phased waves, 2 raster 0, 4 no stores, 8 skewed start,
 *     16 direct fp32 epilogue, 32 the every-4th-gap schedule);
 * 8 = the round-3 default: EIGHT waves in two ping-pong groups, direct epilogues (the reference the epilogue tests compare with);
 * 7 = one wave per SIMD, compiler-scheduled;   2 = 256x128x64, 8 waves, 3 stages;   1 = 128x128x64, 4 waves, 2 stages. */
int mg_gemm_set_variant(int variant);

/* Kernel behind mg_attn_fwd_bf16_hd128* and the matching K row order of mg_pack_kv_bf16 (pack and attend under the same setting):
 * 0 = "m16" (the product's kernel); 3 = "w64", the round-2 kernel (32x32x16 MFMA, per-row softmax reference; NOT the same bits).
 * Returns MG_ERR_ARG for anything else. */
int mg_attn_set_variant(int variant);

void mg_attn_w64_profile(unsigned long long* dev_buf);     /* both attention kernels: 4 waves x {fence, step A, step B, iterations} at [0, 16); the m16 kernel also stores every workgroup's {start, end} (s_memrealtime) of the last launch at [16 + 2 b]: dev_buf holds 16 + 2 x 512 + 8 entries; m16 also adds wave 0's per-item phases {first loads landed, first tile, refill wait, steady loop, drain, epilogue, items, epilogue up to its last store's issue} at [1040, 1048) */
void mg_attn_w64_debug(int flags);                         /* bit 0: keep the pipelined result of flagged blocks (no exact pass) */
void mg_attn_w64_flag_counter(unsigned* dev_counter);      /* dev_counter[2]: [0] += query blocks whose pipelined pass flagged (m16: repeated with swept row maxima; w64: redone by the exact loop), [1] += m16 blocks that went on to the exact loop */
void mg_gemm_debug_profile(unsigned long long* dev_buf);   /* GEMM variants 1/2: 8 waves x {wait+barrier, stage issue, MFMA, k-tiles} */
void mg_gemm5_debug_profile(unsigned long long* dev_buf);  /* GEMM variants 7 / 8 / 11 / 12: per-wave s_memtime sums at [0, 64) (csrc/tools/selftest.cpp gemmprof prints each layout); variant 12 also stores every workgroup's {start, end} (s_memrealtime) at [64 + 2 b]: dev_buf holds 64 + 2 x 512 entries */
#endif /* MG_AB_BUILD */

#ifdef __cplusplus
}
#endif
#endif /* MOVIIGEN_HIP_H */
