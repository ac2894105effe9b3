"""ctypes binding of libfvhost.so (include/flashvit_host.h): generate_data text
format in and out, CPU only.  Loader counterpart of the reference's
InitElement/create_vit (src/FLASH_Viterbi_multithread.c:56-107)."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.HOST_LIB
        if not os.path.isfile(path):
            path = _build.build_host()
        L = ctypes.CDLL(path)
        cp, sz, ci, u32 = ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        vp = ctypes.c_void_p
        L.fvh_quantize_text16.argtypes = [vp, vp, sz]
        L.fvh_write_matrix_text16.argtypes = [cp, vp, sz, sz, ci]
        L.fvh_write_matrix_text16_ex.argtypes = [cp, vp, sz, sz, ci, ci]
        L.fvh_write_ints_text.argtypes = [cp, vp, sz]
        L.fvh_read_floats_text.argtypes = [cp, vp, sz]
        L.fvh_read_ints_text.argtypes = [cp, vp, sz]
        L.fvh_write_bin.argtypes = [cp, vp, u32, u32, u32]
        L.fvh_read_bin.argtypes = [cp, vp, u32, u32, u32]
        L.fvh_write_bin_src.argtypes = [cp, vp, u32, u32, u32, cp]
        L.fvh_read_bin_src.argtypes = [cp, vp, u32, u32, u32, cp]
        L.fvh_strerror.restype = cp
        L.fvh_strerror.argtypes = [ci]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise IOError(f"{what}: {lib().fvh_strerror(rc).decode()} ({rc})")


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def quantize_text16(a64):
    """float32 array the reference loader would read back from '%.16f' text of a64."""
    a64 = np.ascontiguousarray(a64, dtype=np.float64)
    out = np.empty(a64.shape, dtype=np.float32)
    _check(lib().fvh_quantize_text16(_ptr(a64), _ptr(out), a64.size), "quantize_text16")
    return out


def write_matrix_text16(path, a64):
    a64 = np.ascontiguousarray(a64, dtype=np.float64)
    assert a64.ndim == 2
    _check(lib().fvh_write_matrix_text16(path.encode(), _ptr(a64), a64.shape[0], a64.shape[1], 1), path)


def append_matrix_text16(path, a64, first):
    """Row block of a matrix written by write_matrix_text16, appended (first=True truncates)."""
    a64 = np.ascontiguousarray(a64, dtype=np.float64)
    assert a64.ndim == 2
    _check(lib().fvh_write_matrix_text16_ex(path.encode(), _ptr(a64), a64.shape[0], a64.shape[1], 1, 0 if first else 1), path)


BIN_MAGIC = 0x31425646


def bin_header(dtype_code, rows, cols):
    return np.array([BIN_MAGIC, dtype_code, rows, cols], dtype=np.uint32).tobytes()


def write_vector_text16(path, v64):
    v64 = np.ascontiguousarray(v64, dtype=np.float64).reshape(-1)
    _check(lib().fvh_write_matrix_text16(path.encode(), _ptr(v64), v64.size, 1, 0), path)


def write_ints_text(path, v):
    v = np.ascontiguousarray(v, dtype=np.int32).reshape(-1)
    _check(lib().fvh_write_ints_text(path.encode(), _ptr(v), v.size), path)


def read_floats_text(path, shape):
    out = np.empty(shape, dtype=np.float32)
    _check(lib().fvh_read_floats_text(path.encode(), _ptr(out), out.size), path)
    return out


def read_ints_text(path, n):
    out = np.empty(n, dtype=np.int32)
    _check(lib().fvh_read_ints_text(path.encode(), _ptr(out), n), path)
    return out


def write_bin_f32(path, a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    _check(lib().fvh_write_bin(path.encode(), _ptr(a2), 1, a2.shape[0], a2.shape[1]), path)


def write_bin_i32(path, a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    _check(lib().fvh_write_bin(path.encode(), _ptr(a2), 2, a2.shape[0], a2.shape[1]), path)


def read_bin_f32(path, rows, cols):
    out = np.empty((rows, cols), dtype=np.float32)
    _check(lib().fvh_read_bin(path.encode(), _ptr(out), 1, rows, cols), path)
    return out


def write_bin_f32_src(path, a, src_text_path):
    """Cache bound to the text file it was parsed from (size + mtime in the header)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    _check(lib().fvh_write_bin_src(path.encode(), _ptr(a2), 1, a2.shape[0], a2.shape[1], src_text_path.encode()), path)


def read_bin_f32_src(path, rows, cols, src_text_path):
    """Raises IOError when the text file exists and is not the one the cache was made from."""
    out = np.empty((rows, cols), dtype=np.float32)
    _check(lib().fvh_read_bin_src(path.encode(), _ptr(out), 1, rows, cols, src_text_path.encode()), path)
    return out


def read_bin_i32(path, rows, cols):
    out = np.empty((rows, cols), dtype=np.int32)
    _check(lib().fvh_read_bin(path.encode(), _ptr(out), 2, rows, cols), path)
    return out


def load_model_text(data_dir, K, M, T, prob):
    """(A, B, Pi, ob) as the reference programs see them: file naming of getAddress
    (FLASH_Viterbi_multithread.c:48-54), float32 via strtof, ob via %d."""
    stem = lambda kind: os.path.join(data_dir, f"{kind}_K{K}_T{T}_prob{prob}.txt")
    return (read_floats_text(stem("A"), (K, K)), read_floats_text(stem("B"), (K, M)),
            read_floats_text(stem("Pi"), (K,)), read_ints_text(stem("ob"), T))
