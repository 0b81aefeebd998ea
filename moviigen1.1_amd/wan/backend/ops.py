"""Thin typed wrappers: torch CUDA tensors in, C-ABI kernel launches out (on torch's current
stream).  torch is used here for device memory and streams only — every arithmetic step is a
libmoviigen_hip.so kernel; a failure raises, nothing falls back to torch math."""
import ctypes

import torch

from . import lib

BIAS_BF16, BIAS_GELU_BF16, GATE_RESID_F32, BIAS_F32 = 0, 1, 2, 3


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise lib.MoviigenHipError(f'{name}: expected a CUDA (HIP) tensor — the hot path has no CPU fallback')
    if t.dtype != dtype:
        raise lib.MoviigenHipError(f'{name}: expected {dtype}, got {t.dtype}')
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise lib.MoviigenHipError(f'{name}: innermost dimension must be contiguous')


def ln_modulate(x, scale, shift, add_one, eps, out, round_norm_bf16=False):
    """x [rows, dim] fp32 -> out [rows, dim] (bf16 or fp32)."""
    _chk(x, torch.float32, 'x'); _chk(scale, torch.float32, 'scale'); _chk(shift, torch.float32, 'shift')
    rows, dim = x.shape
    lib.call('mg_ln_modulate', _p(x), x.stride(0), rows, dim, _p(scale), _p(shift), int(add_one), float(eps),
             int(round_norm_bf16), _p(out), int(out.dtype == torch.float32), out.stride(0), _st())
    return out


ATTN_LOG2E = 1.4426950408889634


def rmsnorm_rope(x, weight, eps, head_dim, out, rope_cs=None, grid=(1, 1, 1), pos0=0, out_scale=1.0):
    """x [rows, dim] bf16 (may be a strided column slice) -> out bf16.  out_scale: factor applied before the one
    rounding to bf16 (the attention's scale*log2(e) for a q that goes to attention_hd128(..., prescaled=True))."""
    _chk(x, torch.bfloat16, 'x'); _chk(out, torch.bfloat16, 'out'); _chk(weight, torch.float32, 'weight')
    rows, dim = x.shape
    lib.call('mg_rmsnorm_rope_bf16', _p(x), x.stride(0), _p(out), out.stride(0), rows, dim, _p(weight), float(eps),
             int(head_dim), _p(rope_cs), int(grid[0]), int(grid[1]), int(grid[2]), int(pos0), float(out_scale), _st())
    return out


def packed_kv_numel(L, heads):
    """elements of one packed K (or V) buffer for L keys: heads * ceil(L/64) tiles of 8192."""
    return heads * ((L + 63) // 64) * 8192


def pack_kv(k, v, heads, kp, vp):
    """k, v [L, heads*128] bf16 (strided column slices ok; either may be None) -> packed 64-key tiles
    kp / vp (flat bf16 buffers of packed_kv_numel elements) for attention_hd128."""
    _chk(k, torch.bfloat16, 'k'); _chk(v, torch.bfloat16, 'v'); _chk(kp, torch.bfloat16, 'kp'); _chk(vp, torch.bfloat16, 'vp')
    ref = k if k is not None else v
    L = ref.shape[0]
    for name, buf in (('kp', kp if k is not None else None), ('vp', vp if v is not None else None)):
        if buf is not None and buf.numel() < packed_kv_numel(L, heads):
            raise lib.MoviigenHipError(f'{name} too small for {L} keys x {heads} heads')
    lib.call('mg_pack_kv_bf16', _p(k), 0 if k is None else k.stride(0), _p(v), 0 if v is None else v.stride(0), L,
             int(heads), 128, _p(kp), _p(vp), _st())


def gemm(a, w, bias, epilogue, out, gate=None):
    """out[M,N] (+)= a[M,K] @ w[N,K]^T (+bias ...) — see MG_EPI_* in include/moviigen_hip.h."""
    _chk(a, torch.bfloat16, 'a'); _chk(w, torch.bfloat16, 'w'); _chk(bias, torch.float32, 'bias')
    _chk(gate, torch.float32, 'gate')
    want = torch.bfloat16 if epilogue in (BIAS_BF16, BIAS_GELU_BF16) else torch.float32
    _chk(out, want, 'out')
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or out.shape[0] != M or out.shape[1] != N:
        raise lib.MoviigenHipError(f'gemm shape mismatch a{tuple(a.shape)} w{tuple(w.shape)} out{tuple(out.shape)}')
    lib.call('mg_gemm_bf16', _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), M, N, K, int(epilogue), _p(out),
             out.stride(0), _p(gate), _st())
    return out


_ATTN_WS = {}


def attention_workspace(device=None, stream=None):
    """The caller-owned workspace of mg_attn_fwd_bf16_hd128* for launches on `stream` of `device` (default: the current ones):
    mg_attn_workspace_bytes() zeroed bytes, allocated once per (device, stream) HERE — outside the library, which never
    allocates — and shared by that stream's launches (they are ordered; each leaves it zeroed)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    st = torch.cuda.current_stream(dev) if stream is None else stream
    key = (dev, st.cuda_stream)
    ws = _ATTN_WS.get(key)
    if ws is None:
        with torch.cuda.stream(st):
            ws = torch.zeros(int(lib.load().mg_attn_workspace_bytes()), dtype=torch.uint8, device=f'cuda:{dev}')
        _ATTN_WS[key] = ws
    return ws


def attention_hd128(q, kp, vp, out, lk, heads, scale, prescaled=False, reserve_cus=0, workspace='stream'):
    """q [Lq, >=heads*128] bf16; kp/vp from pack_kv for the same lk keys; out [Lq, >=heads*128].
    prescaled: q was produced with rmsnorm_rope(out_scale=scale * ATTN_LOG2E) — `scale` is then only documentation.
    reserve_cus (prescaled entry): CUs the persistent grid leaves free for a kernel on another stream (the exchange).
    workspace: 'stream' = attention_workspace() of the current stream; a uint8 tensor of the caller's; None = no tickets."""
    _chk(q, torch.bfloat16, 'q'); _chk(kp, torch.bfloat16, 'kp'); _chk(vp, torch.bfloat16, 'vp')
    _chk(out, torch.bfloat16, 'out')
    if min(kp.numel(), vp.numel()) < packed_kv_numel(int(lk), int(heads)):
        raise lib.MoviigenHipError('packed K/V buffers too small for lk keys')
    ws = attention_workspace(q.device) if isinstance(workspace, str) else workspace
    if prescaled:
        lib.call('mg_attn_fwd_bf16_hd128_prescaled', _p(q), q.stride(0), _p(kp), _p(vp), _p(out), out.stride(0), None,
                 q.shape[0], int(lk), int(heads), int(reserve_cus), _p(ws), _st())
    else:
        lib.call('mg_attn_fwd_bf16_hd128', _p(q), q.stride(0), _p(kp), _p(vp), _p(out), out.stride(0), q.shape[0],
                 int(lk), int(heads), float(scale), _p(ws), _st())
    return out


def attention_generic(q, k, v, out, lk, heads, head_dim, scale):
    _chk(q, torch.bfloat16, 'q'); _chk(k, torch.bfloat16, 'k'); _chk(v, torch.bfloat16, 'v')
    _chk(out, torch.bfloat16, 'out')
    lib.call('mg_attn_fwd_bf16_generic', _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out),
             out.stride(0), q.shape[0], int(lk), int(heads), int(head_dim), float(scale), _st())
    return out


def sinusoid_embed(t, dim, out):
    code = {torch.int64: 0, torch.float32: 1, torch.float64: 2}.get(t.dtype)
    if code is None or not t.is_cuda:
        raise lib.MoviigenHipError('timestep must be a CUDA int64/float32/float64 tensor')
    lib.call('mg_sinusoid_embed', _p(t), code, t.numel(), int(dim), _p(out), _st())
    return out


def gemv(w, bias, x, y, silu_in=False):
    _chk(w, torch.float32, 'w'); _chk(x, torch.float32, 'x'); _chk(y, torch.float32, 'y')
    lib.call('mg_gemv_f32', _p(w), _p(bias), _p(x), _p(y), w.shape[0], w.shape[1], int(silu_in), _st())
    return y


def add_rows(a, b, out, period):
    rows, dim = a.shape
    lib.call('mg_add_rows_f32', _p(a), _p(b), _p(out), rows, dim, int(period), _st())
    return out


def head_gemm(x, w, bias, out):
    _chk(x, torch.float32, 'x'); _chk(w, torch.float32, 'w'); _chk(out, torch.float32, 'out')
    lib.call('mg_head_gemm_f32', _p(x), x.stride(0), _p(w), _p(bias), _p(out), x.shape[0], w.shape[0], w.shape[1],
             _st())
    return out


def patchify(lat, ph, pw, out):
    _chk(lat, torch.float32, 'lat'); _chk(out, torch.bfloat16, 'out')
    C, F, H, W = lat.shape
    lib.call('mg_patchify_bf16', _p(lat), C, F, H, W, int(ph), int(pw), _p(out), out.stride(0), _st())
    return out


def unpatchify(tok, C, F, Hg, Wg, ph, pw, lat):
    _chk(tok, torch.float32, 'tok'); _chk(lat, torch.float32, 'lat')
    lib.call('mg_unpatchify_f32', _p(tok), tok.stride(0), C, F, Hg, Wg, int(ph), int(pw), _p(lat), _st())
    return lat


def lincomb(out, terms):
    """out = sum(c_i * x_i) for up to 4 (tensor, coef) terms."""
    terms = list(terms) + [(None, 0.0)] * (4 - len(terms))
    args = []
    for x, c in terms:
        _chk(x, torch.float32, 'x')
        args += [_p(x), float(c)]
    lib.call('mg_lincomb4_f32', _p(out), out.numel(), *args, _st())
    return out


def cfg_combine(out, uncond, cond, g):
    _chk(uncond, torch.float32, 'uncond'); _chk(cond, torch.float32, 'cond')
    lib.call('mg_cfg_combine_f32', _p(out), _p(uncond), _p(cond), float(g), out.numel(), _st())
    return out


# ---- umT5 encoder glue ---------------------------------------------------------------------------
def embed_rows(table, ids, out):
    _chk(table, torch.bfloat16, 'table'); _chk(ids, torch.int64, 'ids'); _chk(out, torch.bfloat16, 'out')
    lib.call('mg_embed_rows_bf16', _p(table), table.shape[0], table.shape[1], _p(ids), ids.numel(), _p(out), _st())
    return out


def ew_bf16(a, b, out, mode):
    """mode 0: a + b; mode 1: a * gelu_tanh(b) — contiguous bf16 tensors of equal size."""
    for n, t in (('a', a), ('b', b), ('out', out)):
        _chk(t, torch.bfloat16, n)
        if not t.is_contiguous():
            raise lib.MoviigenHipError(f'{n} must be contiguous')
    lib.call('mg_ew_bf16', _p(a), _p(b), _p(out), out.numel(), int(mode), _st())
    return out


def t5_attention(q, k, v, rel_emb, rel_bucket, out, lk, heads, head_dim):
    """q, k, v: column slices of one [L, 3*heads*head_dim] buffer (same row stride)."""
    _chk(q, torch.bfloat16, 'q'); _chk(k, torch.bfloat16, 'k'); _chk(v, torch.bfloat16, 'v')
    _chk(rel_emb, torch.bfloat16, 'rel_emb'); _chk(rel_bucket, torch.int32, 'rel_bucket'); _chk(out, torch.bfloat16, 'out')
    if not (q.stride(0) == k.stride(0) == v.stride(0)):
        raise lib.MoviigenHipError('q, k, v must share a row stride')
    if rel_bucket.numel() != q.shape[0] + int(lk) - 1:
        raise lib.MoviigenHipError('rel_bucket must have Lq + Lk - 1 entries')
    lib.call('mg_t5_attn_bf16', _p(q), _p(k), _p(v), q.stride(0), _p(rel_emb), _p(rel_bucket), _p(out), out.stride(0),
             q.shape[0], int(lk), int(heads), int(head_dim), _st())
    return out


# ---- VAE (fp32, channels-last) -------------------------------------------------------------------
VAE_EXACT, VAE_BF16X3 = 0, 1     # MG_VAE_EXACT / MG_VAE_BF16X3 (include/moviigen_hip.h): the arithmetic of one convolution call


def vae_conv(x, w, bias, out, kt, kh, kw, cache=None, up2=False, residual=None, mode=VAE_EXACT):
    """x [T,H,W,Cin]; w [Cout,kt,kh,kw,Cin]; out [T,Ho,Wo,Cout]."""
    for n, t in (('x', x), ('w', w), ('bias', bias), ('out', out), ('cache', cache), ('residual', residual)):
        _chk(t, torch.float32, n)
    T, H, W, Cin = x.shape
    tc = 0 if cache is None else cache.shape[0]
    lib.call('mg_vae_conv_f32', _p(x), _p(cache), tc, T, H, W, Cin, _p(w), _p(bias), w.shape[0], kt, kh, kw,
             int(up2), _p(residual), _p(out), int(mode), _st())
    return out


def vae_conv_cols(x, w, bias, out, kt, kh, kw, col0, cache=None, residual=None, mode=VAE_EXACT):
    """the W-band form: x [T,H,W,Cin] and cache carry the neighbours' halo columns; out (and residual) [T,H,cols,Cout] compact =
    the output columns [col0, col0 + cols) of vae_conv on the whole image."""
    for n, t in (('x', x), ('w', w), ('bias', bias), ('out', out), ('cache', cache), ('residual', residual)):
        _chk(t, torch.float32, n)
    T, H, W, Cin = x.shape
    cols = out.shape[2]
    if out.shape[0] != T or out.shape[1] != H or not out.is_contiguous() or not x.is_contiguous():
        raise lib.MoviigenHipError(f'vae_conv_cols: out {tuple(out.shape)} does not match x {tuple(x.shape)}')
    if cache is not None and (tuple(cache.shape[1:]) != tuple(x.shape[1:]) or not cache.is_contiguous()):
        raise lib.MoviigenHipError('vae_conv_cols: the cache must have the haloed geometry of x')
    tc = 0 if cache is None else cache.shape[0]
    lib.call('mg_vae_conv_cols_f32', _p(x), _p(cache), tc, T, H, W, Cin, _p(w), _p(bias), w.shape[0], kt, kh, kw,
             _p(residual), _p(out), int(col0), int(cols), int(mode), _st())
    return out


def vae_upconv_fold_weights(w):
    """w [Cout,1,3,3,Cin] (the conv behind a nearest-2x upsample) -> [4,Cout,2,2,Cin]: one 2x2 kernel per output parity."""
    _chk(w, torch.float32, 'w')
    Cout, kt, kh, kw, Cin = w.shape
    if (kt, kh, kw) != (1, 3, 3):
        raise ValueError(f'the phase decomposition is defined for 1x3x3 kernels, got {kt}x{kh}x{kw}')
    wp = torch.empty(4, Cout, 2, 2, Cin, dtype=torch.float32, device=w.device)
    lib.call('mg_vae_upconv_fold_weights_f32', _p(w), Cout, Cin, _p(wp), _st())
    return wp


def vae_upconv_phases(x, wp, bias, out, mode=VAE_EXACT):
    """x [T,H,W,Cin]; wp from vae_upconv_fold_weights; out [T,2H,2W,Cout] = conv3x3(nearest-2x(x))."""
    for n, t in (('x', x), ('wp', wp), ('bias', bias), ('out', out)):
        _chk(t, torch.float32, n)
    T, H, W, Cin = x.shape
    lib.call('mg_vae_upconv_phases_f32', _p(x), T, H, W, Cin, _p(wp), _p(bias), wp.shape[1], _p(out), int(mode), _st())
    return out


def vae_upconv_phases_cols(x, wp, bias, out, col0, mode=VAE_EXACT):
    """the W-band form of vae_upconv_phases: x [T,H,W,Cin] with halo columns, out [T,2H,2 cols,Cout] compact."""
    for n, t in (('x', x), ('wp', wp), ('bias', bias), ('out', out)):
        _chk(t, torch.float32, n)
    T, H, W, Cin = x.shape
    if out.shape[0] != T or out.shape[1] != 2 * H or out.shape[2] % 2 or not out.is_contiguous() or not x.is_contiguous():
        raise lib.MoviigenHipError(f'vae_upconv_phases_cols: out {tuple(out.shape)} does not match x {tuple(x.shape)}')
    lib.call('mg_vae_upconv_phases_cols_f32', _p(x), T, H, W, Cin, _p(wp), _p(bias), wp.shape[1], _p(out), int(col0), out.shape[2] // 2,
             int(mode), _st())
    return out


def vae_rmsnorm_silu(x, gamma, out, do_silu=True):
    C = x.shape[-1]
    lib.call('mg_vae_rmsnorm_silu_f32', _p(x), _p(gamma), _p(out), x.numel() // C, C, int(do_silu), _st())
    return out


def vae_attn_workspace_floats(L, C):
    return int(lib.load().mg_vae_attn_workspace_floats(int(L), int(C)))


def vae_attn(qkv, out, workspace):
    frames, L, C3 = qkv.shape
    if workspace.numel() < vae_attn_workspace_floats(L, C3 // 3):
        raise lib.MoviigenHipError('vae_attn workspace too small')
    lib.call('mg_vae_attn_f32', _p(qkv), _p(out), frames, L, C3 // 3, _p(workspace), _st())
    return out


def vae_attn_rows(q, kv, out, workspace):
    """q [frames, Lq, >= C] (a column slice of the band's q|k|v is fine), kv [frames, Lk, 2C] = k | v of ALL pixels of the frame in image
    order, out [frames, Lq, C]: the band's rows of vae_attn, same bits."""
    for n, t in (('q', q), ('kv', kv), ('out', out), ('workspace', workspace)):
        _chk(t, torch.float32, n)
    frames, Lq = q.shape[:2]
    Lk, C = kv.shape[1], kv.shape[2] // 2
    if kv.shape[0] != frames or not kv.is_contiguous() or not out.is_contiguous() or q.stride(0) != Lq * q.stride(1) or out.shape != (frames, Lq, C):
        raise lib.MoviigenHipError('vae_attn_rows: shapes')
    if workspace.numel() < vae_attn_workspace_floats(Lk, C):
        raise lib.MoviigenHipError('vae_attn workspace too small')
    lib.call('mg_vae_attn_rows_f32', _p(q), q.stride(1), _p(kv), ctypes.c_void_p(kv.data_ptr() + 4 * C), 2 * C, _p(out), frames, Lq, Lk, C,
             _p(workspace), _st())
    return out


def vae_latent_in(z, mean, inv_std, out):
    C, T, H, W = z.shape
    lib.call('mg_vae_latent_in_f32', _p(z), _p(mean), _p(inv_std), C, T, H, W, _p(out), _st())
    return out


def vae_video_out(x, out, t_off):
    T, H, W, C = x.shape
    lib.call('mg_vae_video_out_f32', _p(x), C, T, H, W, _p(out), int(t_off), out.shape[1], _st())
    return out


def vae_time_interleave(x, out):
    T, H, W, C2 = x.shape
    lib.call('mg_vae_time_interleave_f32', _p(x), T, H * W, C2 // 2, _p(out), _st())
    return out


def video_to_u8(video, lo=-1.0, hi=1.0):
    """[3,T,H,W] fp32 -> uint8 frames [T,H,W,3] (reference cache_video arithmetic)."""
    _chk(video, torch.float32, 'video')
    if video.dim() != 4 or video.shape[0] != 3 or not video.is_contiguous():
        raise lib.MoviigenHipError('video must be a contiguous [3, T, H, W] tensor')
    _, T, H, W = video.shape
    out = torch.empty(T, H, W, 3, dtype=torch.uint8, device=video.device)
    lib.call('mg_video_to_u8', _p(video), T, H, W, float(lo), float(hi), _p(out), _st())
    return out


def gate_residual(x, y, gate=None):
    """x [rows, dim] fp32 += y [rows, dim] bf16 * gate [dim] fp32 (None = 1)."""
    _chk(x, torch.float32, 'x'); _chk(y, torch.bfloat16, 'y'); _chk(gate, torch.float32, 'gate')
    if x.shape != y.shape:
        raise lib.MoviigenHipError(f'gate_residual shape mismatch x{tuple(x.shape)} y{tuple(y.shape)}')
    lib.call('mg_gate_residual_f32', _p(x), x.stride(0), _p(y), y.stride(0), _p(gate), x.shape[0], x.shape[1], _st())
    return x


def sp_pack_qkv(q, k, v, P, cols_per_dest, col0, w, send):
    """q, k, v [Lloc, >= P*cols_per_dest] bf16 (column slices ok) -> send [P, Lloc, 3w] (see moviigen_hip.h)."""
    for n, t in (('q', q), ('k', k), ('v', v), ('send', send)):
        _chk(t, torch.bfloat16, n)
    Lloc = q.shape[0]
    if send.numel() < P * Lloc * 3 * w or not send.is_contiguous():
        raise lib.MoviigenHipError('send buffer too small / not contiguous')
    lib.call('mg_sp_pack_qkv_bf16', _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), Lloc, int(P),
             int(cols_per_dest), int(col0), int(w), _p(send), _st())
    return send


def sp_unpack_o(recv, P, cols_per_src, col0, w, o):
    """recv [P, Lloc, w] bf16 -> o[:, p*cols_per_src + col0 : +w] for every source rank p."""
    _chk(recv, torch.bfloat16, 'recv'); _chk(o, torch.bfloat16, 'o')
    Lloc = o.shape[0]
    if recv.numel() < P * Lloc * w or not recv.is_contiguous():
        raise lib.MoviigenHipError('recv buffer too small / not contiguous')
    lib.call('mg_sp_unpack_o_bf16', _p(recv), Lloc, int(P), int(cols_per_src), int(col0), int(w), _p(o), o.stride(0), _st())
    return o


def sp_copy_blocks(src, s_blk, s_row, dst, d_blk, d_row, blocks, rows, width):
    """dst[b][r][:width] = src[b][r][:width] with element strides (*_blk, *_row): the reshapes around an all-to-all."""
    _chk(src, torch.bfloat16, 'src'); _chk(dst, torch.bfloat16, 'dst')
    lib.call('mg_sp_copy_blocks_bf16', _p(src), int(s_blk), int(s_row), _p(dst), int(d_blk), int(d_row), int(blocks),
             int(rows), int(width), _st())
    return dst


def image_to_u8(image, lo=-1.0, hi=1.0):
    """[3,H,W] fp32 -> uint8 pixels [H,W,3] (reference cache_image / torchvision save_image arithmetic: rounds)."""
    _chk(image, torch.float32, 'image')
    if image.dim() != 3 or image.shape[0] != 3 or not image.is_contiguous():
        raise lib.MoviigenHipError('image must be a contiguous [3, H, W] tensor')
    _, H, W = image.shape
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=image.device)
    lib.call('mg_image_to_u8', _p(image), H, W, float(lo), float(hi), _p(out), _st())
    return out


def attention_hd128_lse(q, kp, vp, out, lse, lk, heads, scale, prescaled=False):
    """attention_hd128 that also writes lse [heads, Lq] fp32 (log-sum-exp of the scaled scores)."""
    _chk(q, torch.bfloat16, 'q'); _chk(kp, torch.bfloat16, 'kp'); _chk(vp, torch.bfloat16, 'vp')
    _chk(out, torch.bfloat16, 'out'); _chk(lse, torch.float32, 'lse')
    if min(kp.numel(), vp.numel()) < packed_kv_numel(int(lk), int(heads)):
        raise lib.MoviigenHipError('packed K/V buffers too small for lk keys')
    if lse.numel() < int(heads) * q.shape[0] or not lse.is_contiguous():
        raise lib.MoviigenHipError('lse must be a contiguous [heads, Lq] fp32 tensor')
    ws = attention_workspace(q.device)
    if prescaled:
        lib.call('mg_attn_fwd_bf16_hd128_prescaled', _p(q), q.stride(0), _p(kp), _p(vp), _p(out), out.stride(0), _p(lse),
                 q.shape[0], int(lk), int(heads), 0, _p(ws), _st())
    else:
        lib.call('mg_attn_fwd_bf16_hd128_lse', _p(q), q.stride(0), _p(kp), _p(vp), _p(out), out.stride(0), _p(lse),
                 q.shape[0], int(lk), int(heads), float(scale), _p(ws), _st())
    return out, lse


def attention_merge(acc, lse_acc, part, lse_part, heads, first, out=None):
    """ring attention: fold block result (part bf16, lse_part) into (acc fp32, lse_acc); out = bf16 copy of acc."""
    _chk(acc, torch.float32, 'acc'); _chk(lse_acc, torch.float32, 'lse_acc'); _chk(part, torch.bfloat16, 'part')
    _chk(lse_part, torch.float32, 'lse_part'); _chk(out, torch.bfloat16, 'out')
    lib.call('mg_attn_merge_f32', _p(acc), acc.stride(0), _p(lse_acc), _p(part), part.stride(0), _p(lse_part), _p(out),
             out.stride(0) if out is not None else 0, acc.shape[0], int(heads), int(bool(first)), _st())
    return acc
