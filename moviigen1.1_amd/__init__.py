"""moviigen1.1_amd — MI355X-native engine for the MoviiGen1.1 denoising hot path.

This directory is a PATH ENTRY, not an importable name (it contains a dot): put it on sys.path
and `import wan` — the drop-in mirror of the reference's `wan` package:

    sys.path.insert(0, "<repo>/moviigen1.1_amd"); import wan; wan.WanT2V(...)

  wan/        host-side mirror of the reference interface (WanT2V, WanModel, WanVAE, schedulers,
              flash_attention, configs, distributed)
  csrc/       HIP kernels for gfx950 + the C-ABI (include/moviigen_hip.h)
  lib/        libmoviigen_hip.so, built in-tree by __graft_entry__.build()
"""
