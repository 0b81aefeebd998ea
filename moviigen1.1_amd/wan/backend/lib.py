"""ctypes binding of libmoviigen_hip.so (C-ABI declared in include/moviigen_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel returns an error
code this module raises — it never routes to torch ops or to the oracle.

Two builds of the same sources exist (csrc/Makefile): the PRODUCT library libmoviigen_hip.so — one kernel per shape
class, no kernel-selection switch, no profiling hook — is what `load()` returns and what every `wan` module runs on.
libmoviigen_hip_ab.so (-DMG_AB_BUILD) adds the measurement partners (GEMM variants 7 / 8 / 11, the w64 attention kernel),
`mg_gemm_set_variant` / `mg_attn_set_variant` and the s_memtime hooks; only tests/, tools/ and bench.py's measurement
flags ask for it, through `load_ab()` or the `ab_library()` scope."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libmoviigen_hip.so'))
LIB_AB_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'lib', 'libmoviigen_hip_ab.so'))

c_i64, c_int, c_f32, c_vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/moviigen_hip.h
SIGNATURES = {
    'mg_version': [],
    'mg_abi_version': [],
    'mg_ln_modulate': [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_int, c_f32, c_int, c_vp, c_int, c_i64, c_vp],
    'mg_rmsnorm_rope_bf16': [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_f32, c_int, c_vp, c_int, c_int,
                             c_int, c_i64, c_f32, c_vp],
    'mg_pack_kv_bf16': [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_vp],
    'mg_gemm_bf16': [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp],
    'mg_attn_workspace_bytes': [],
    'mg_attn_fwd_bf16_hd128': [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_f32, c_vp, c_vp],
    'mg_attn_fwd_bf16_hd128_lse': [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_f32, c_vp, c_vp],
    'mg_attn_fwd_bf16_hd128_prescaled': [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_vp],
    'mg_attn_merge_f32': [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp],
    'mg_attn_fwd_bf16_generic': [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_int, c_int,
                                 c_f32, c_vp],
    'mg_sinusoid_embed': [c_vp, c_int, c_int, c_int, c_vp, c_vp],
    'mg_gemv_f32': [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    'mg_add_rows_f32': [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    'mg_head_gemm_f32': [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    'mg_patchify_bf16': [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp],
    'mg_unpatchify_f32': [c_vp, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    'mg_lincomb4_f32': [c_vp, c_i64, c_vp, c_f32, c_vp, c_f32, c_vp, c_f32, c_vp, c_f32, c_vp],
    'mg_cfg_combine_f32': [c_vp, c_vp, c_vp, c_f32, c_i64, c_vp],
    'mg_embed_rows_bf16': [c_vp, c_i64, c_int, c_vp, c_int, c_vp, c_vp],
    'mg_ew_bf16': [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    'mg_t5_attn_bf16': [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp],
    'mg_vae_conv_f32': [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                        c_int, c_vp, c_vp, c_int, c_vp],
    'mg_vae_upconv_fold_weights_f32': [c_vp, c_int, c_int, c_vp, c_vp],
    'mg_vae_upconv_phases_f32': [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp],
    'mg_vae_conv_cols_f32': [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int,
                             c_vp, c_vp, c_int, c_int, c_int, c_vp],
    'mg_vae_upconv_phases_cols_f32': [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp],
    'mg_vae_rmsnorm_silu_f32': [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    'mg_vae_attn_workspace_floats': [c_i64, c_int],
    'mg_vae_attn_f32': [c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_vp],
    'mg_vae_attn_rows_f32': [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_vp],
    'mg_vae_latent_in_f32': [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp],
    'mg_vae_video_out_f32': [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_int, c_vp],
    'mg_vae_time_interleave_f32': [c_vp, c_int, c_i64, c_int, c_vp, c_vp],
    'mg_video_to_u8': [c_vp, c_int, c_int, c_int, c_f32, c_f32, c_vp, c_vp],
    'mg_comm_unique_id': [c_vp],
    'mg_comm_create': [c_vp, c_int, c_int, c_vp],
    'mg_comm_destroy': [c_vp],
    'mg_sp_all_to_all': [c_vp, c_vp, c_vp, c_i64, c_vp],
    'mg_sp_all_to_all_4d_bf16': [c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_vp],
    'mg_sp_all_gather': [c_vp, c_vp, c_vp, c_i64, c_vp],
    'mg_shard_all_gather': [c_vp, c_vp, c_vp, c_i64, c_vp],
    'mg_gate_residual_f32': [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int, c_vp],
    'mg_image_to_u8': [c_vp, c_int, c_int, c_f32, c_f32, c_vp, c_vp],
    'mg_sp_pack_qkv_bf16': [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp],
    'mg_sp_copy_blocks_bf16': [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_int, c_i64, c_int, c_vp],
    'mg_sp_unpack_o_bf16': [c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp],
}
# the header's MG_AB_BUILD section: exported by libmoviigen_hip_ab.so only
SIGNATURES_AB = {
    'mg_attn_set_variant': [c_int],
    'mg_gemm_set_variant': [c_int],
    'mg_attn_w64_profile': [c_vp],
    'mg_attn_w64_debug': [c_int],
    'mg_attn_w64_flag_counter': [c_vp],
    'mg_gemm_debug_profile': [c_vp],
    'mg_gemm5_debug_profile': [c_vp],
}
_RESTYPE = {'mg_version': ctypes.c_char_p, 'mg_vae_attn_workspace_floats': ctypes.c_int64, 'mg_attn_workspace_bytes': ctypes.c_int64, 'mg_attn_w64_profile': None,
            'mg_attn_w64_debug': None, 'mg_attn_w64_flag_counter': None, 'mg_gemm_debug_profile': None, 'mg_gemm5_debug_profile': None}
DEFAULT_GEMM_VARIANT = 0   # mg_gemm_set_variant(0) = the product's rule by shape (A/B library)

ERRORS = {-1: 'MG_ERR_ARG (null pointer / bad enum)', -2: 'MG_ERR_SHAPE (unsupported shape or alignment)',
          -3: 'MG_ERR_LAUNCH (kernel launch failed)', -4: 'MG_ERR_UNAVAILABLE (librccl could not be bound)',
          -5: 'MG_ERR_COMM (an RCCL call failed)'}

_lib = None
_lib_ab = None
_use_ab = False
DEFAULT_ATTN_VARIANT = 0   # mg_attn_set_variant(0) = the product's kernel (A/B library)


class MoviigenHipError(RuntimeError):
    pass


def _open(path, sigs):
    if not os.path.exists(path):
        raise MoviigenHipError(
            f'{path} not found: build it with `python __graft_entry__.py build` '
            '(hipcc --offload-arch=gfx950); there is no CPU/torch fallback for the hot path')
    lib = ctypes.CDLL(path)
    for name, args in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    return lib


def load():
    """dlopen the in-tree PRODUCT library; raises (never falls back) when it is absent.  Inside an `ab_library()` scope — tests and
    measurement tools only — the A/B library instead."""
    global _lib
    if _use_ab:
        return load_ab()
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def load_ab():
    """the A/B library (measurement partners, kernel-selection switches, s_memtime hooks): tests/, tools/, bench.py --gemm-variant"""
    global _lib_ab
    if _lib_ab is None:
        _lib_ab = _open(LIB_AB_PATH, {**SIGNATURES, **SIGNATURES_AB})
    return _lib_ab


class ab_library:
    """`with lib.ab_library():` — every kernel call of the scope goes to the A/B library (so that `mg_*_set_variant` and the hooks act on
    what `wan.backend.ops` runs); the switches are reset when the scope ends.  NOT for product code: process-global, not thread-safe."""

    def __enter__(self):
        global _use_ab
        self._prev, _use_ab = _use_ab, True
        return load_ab()

    def __exit__(self, *exc):
        global _use_ab
        if self._prev:          # an inner scope: the outermost one resets
            return False
        h = load_ab()
        h.mg_attn_set_variant(DEFAULT_ATTN_VARIANT)
        h.mg_gemm_set_variant(DEFAULT_GEMM_VARIANT)
        h.mg_attn_w64_debug(0)
        h.mg_attn_w64_profile(None)
        h.mg_attn_w64_flag_counter(None)
        h.mg_gemm_debug_profile(None)
        h.mg_gemm5_debug_profile(None)
        _use_ab = self._prev
        return False


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MoviigenHipError(f'{name} failed: {ERRORS.get(rc, rc)}')
