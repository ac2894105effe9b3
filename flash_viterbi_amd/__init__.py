"""flash_viterbi_amd — MI355X-native FLASH / FLASH-BS Viterbi decoding path.

Only what the hot path needs: csrc/ (HIP kernels + the C-ABI shim declared in
include/flashvit.h), the C host programs under src/ that keep the reference's
CLI surface, the generate_data counterpart, and thin ctypes mirrors for tests
and bench.  Importing this package never touches the GPU.
"""
__all__ = ["build", "hostio"]
