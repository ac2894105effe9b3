"""ctypes mirror of include/flashvit.h — the host-side Python face of libflashvit.so.

There is deliberately no CPU fallback here: if the HIP library is missing or no GPU is
visible, construction fails loudly.  (The CPU restatement lives in oracle/ and is test
infrastructure; this module never imports it.)
"""
import ctypes
import os

import numpy as np

from . import build as _build

MODE_REFERENCE = 0
MODE_SINGLE_PASS = 1
KERNEL_AUTO, KERNEL_F64_STREAM, KERNEL_F32_REFINE, KERNEL_F16_REFINE, KERNEL_Q16_REFINE, KERNEL_SPARSE_Q16 = 0, 1, 2, 3, 4, 5
KERNEL_U16_REFINE = 6
OPT_KERNEL, OPT_MAX_BATCH, OPT_PROFILE, OPT_SEL_MARGIN, OPT_DEBUG = 1, 2, 3, 4, 100
DEBUG_TIMING_ONLY = (1 << 0) | (1 << 4) | (1 << 5) | (1 << 11) | (1 << 12)      # refused by the shipped library
WARN_BEAM_MISS = 1
UNIQUE_ID_BYTES = 128


class Stats(ctypes.Structure):
    _fields_ = [("set_model_ms", ctypes.c_double), ("decode_ms", ctypes.c_double), ("gpu_ms", ctypes.c_double),
                ("top_pass_ms", ctypes.c_double), ("top_steps_ms", ctypes.c_double),
                ("step_kernel_ms", ctypes.c_double),
                ("step_launches", ctypes.c_longlong), ("task_steps", ctypes.c_longlong),
                ("column_steps", ctypes.c_longlong),
                ("cells", ctypes.c_longlong), ("alg_bytes", ctypes.c_longlong),
                ("table_bytes_per_step", ctypes.c_longlong), ("device_bytes", ctypes.c_longlong),
                ("refine_near", ctypes.c_longlong), ("refine_rescan", ctypes.c_longlong),
                ("beam_exact_sets", ctypes.c_longlong), ("beam_ties", ctypes.c_longlong),
                ("beam_dup_cols", ctypes.c_longlong), ("beam_dup_steps", ctypes.c_longlong),
                ("beam_cand_selects", ctypes.c_longlong), ("density", ctypes.c_double),
                ("passes", ctypes.c_int), ("generations", ctypes.c_int), ("kernel", ctypes.c_int),
                ("ranks", ctypes.c_int), ("refine_saturated", ctypes.c_longlong),
                ("beam_spec_steps", ctypes.c_longlong), ("beam_reach_events", ctypes.c_longlong),
                ("beam_list_short", ctypes.c_longlong), ("beam_list_long", ctypes.c_longlong), ("beam_list_entries", ctypes.c_longlong),
                ("beam_chain_cuts", ctypes.c_longlong)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PassInfo(ctypes.Structure):
    _fields_ = [("L", ctypes.c_int), ("R", ctypes.c_int), ("generation", ctypes.c_int), ("owner", ctypes.c_int)]


EXPORTS = ["fv_create", "fv_destroy", "fv_set_model", "fv_set_option", "fv_decode_full", "fv_decode_beam",
           "fv_decode_vanilla", "fv_decode_checkpoint", "fv_checkpoint_memory_bytes",
           "fv_last_stats", "fv_strerror", "fv_last_error_detail", "fv_reference_memory_bytes",
           "fv_comm_unique_id", "fv_comm_init", "fv_plan_passes", "fv_merge_paths", "fv_set_partition",
           "fv_create_multi", "fv_device_count"]

_lib = None


def load_library():
    """dlopen libflashvit.so (no GPU needed for that) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    # FLASHVIT_TIMING_BUILD=1 (tools/ only): the build that keeps the result-changing timing switches of FV_OPT_DEBUG
    path = _build.HIP_TIMING_LIB if os.environ.get("FLASHVIT_TIMING_BUILD") == "1" else _build.HIP_LIB
    if not os.path.isfile(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = ctypes.CDLL(path)
    vp, ci, cll = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
    L.fv_create.argtypes = [ctypes.POINTER(vp), ci]
    L.fv_create_multi.argtypes = [ctypes.POINTER(vp), vp, ci]
    L.fv_device_count.argtypes = []
    L.fv_destroy.argtypes = [vp]
    L.fv_destroy.restype = None
    L.fv_set_model.argtypes = [vp, vp, vp, vp, ci, ci]
    L.fv_set_option.argtypes = [vp, ci, cll]
    L.fv_decode_full.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    L.fv_decode_beam.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp]
    L.fv_decode_vanilla.argtypes = [vp, vp, ci, vp, vp]
    L.fv_decode_checkpoint.argtypes = [vp, vp, ci, ci, vp, vp]
    L.fv_checkpoint_memory_bytes.argtypes = [ci, ci, ci]
    L.fv_checkpoint_memory_bytes.restype = ctypes.c_longlong
    L.fv_last_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.fv_strerror.argtypes = [ci]
    L.fv_strerror.restype = ctypes.c_char_p
    L.fv_last_error_detail.argtypes = [vp]
    L.fv_last_error_detail.restype = ctypes.c_char_p
    L.fv_reference_memory_bytes.argtypes = [ci, ci, ci, ci]
    L.fv_reference_memory_bytes.restype = cll
    L.fv_comm_unique_id.argtypes = [vp]
    L.fv_comm_init.argtypes = [vp, ci, ci, vp]
    L.fv_set_partition.argtypes = [vp, ci, ci]
    L.fv_plan_passes.argtypes = [ci, ci, ci, ci, ctypes.POINTER(PassInfo), ci]
    L.fv_merge_paths.argtypes = [ci, ci, ci, vp, vp]
    _lib = L
    return L


class FlashVitError(RuntimeError):
    def __init__(self, rc, detail=""):
        msg = load_library().fv_strerror(rc).decode()
        super().__init__(f"flashvit: {msg} ({rc})" + (f": {detail}" if detail else ""))
        self.rc = rc


def plan_passes(T, n_split, mode=MODE_REFERENCE, nranks=1):
    """Host-side schedule (no GPU): list of (L, R, generation, owner)."""
    L = load_library()
    n = L.fv_plan_passes(T, n_split, mode, nranks, None, 0)
    if n < 0:
        raise FlashVitError(n)
    buf = (PassInfo * n)()
    L.fv_plan_passes(T, n_split, mode, nranks, buf, n)
    return [(p.L, p.R, p.generation, p.owner) for p in buf]


def merge_paths(T, n_split, nranks, gathered):
    """The product's post-all-gather merge (host C++), callable without a GPU."""
    g = np.ascontiguousarray(gathered, dtype=np.int32).reshape(nranks * T)
    out = np.empty(T, dtype=np.int32)
    rc = load_library().fv_merge_paths(T, n_split, nranks, _p(g), _p(out))
    if rc < 0:
        raise FlashVitError(rc)
    return out


def reference_memory_bytes(K, T, n_split, beam=0):
    return int(load_library().fv_reference_memory_bytes(K, T, n_split, beam))


def checkpoint_memory_bytes(K, T, step=0):
    return int(load_library().fv_checkpoint_memory_bytes(K, T, step))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class FlashViterbi:
    """One decoder context on one GPU.  Mirrors the reference program's life cycle:
    create_vit() -> calc() -> printAns() becomes set_model() -> decode_*() -> returned path."""

    def __init__(self, device=0):
        """device: one device id, or a list of ids for a single-process multi-device context (fv_create_multi;
        an id may repeat: the members then share that GPU and gather by device-to-device copies)."""
        self._L = load_library()
        h = ctypes.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = np.ascontiguousarray(device, dtype=np.int32)
            rc = self._L.fv_create_multi(ctypes.byref(h), _p(devs), devs.size)
        else:
            rc = self._L.fv_create(ctypes.byref(h), device)
        if rc != 0:
            raise FlashVitError(rc, "fv_create (is a GPU visible? there is no CPU fallback)")
        self._h = h
        self.K = self.M = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fv_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc < 0:
            raise FlashVitError(rc, self._L.fv_last_error_detail(self._h).decode())
        return rc

    def set_option(self, key, value):
        self._check(self._L.fv_set_option(self._h, key, int(value)))

    def set_model(self, A, B, Pi):
        A = np.ascontiguousarray(A, dtype=np.float32)
        B = np.ascontiguousarray(B, dtype=np.float32)
        Pi = np.ascontiguousarray(Pi, dtype=np.float32)
        K, M = B.shape
        assert A.shape == (K, K) and Pi.shape == (K,)
        self._check(self._L.fv_set_model(self._h, _p(A), _p(B), _p(Pi), K, M))
        self.K, self.M = K, M

    def decode_full(self, ob, n_split=1, mode=MODE_REFERENCE):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = self._check(self._L.fv_decode_full(self._h, _p(ob), ob.size, n_split, mode, _p(path), ctypes.byref(score)))
        return path, np.float32(score.value), rc

    def decode_beam(self, ob, n_split, beam, mode=MODE_REFERENCE):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = self._check(self._L.fv_decode_beam(self._h, _p(ob), ob.size, n_split, beam, mode, _p(path), ctypes.byref(score)))
        return path, np.float32(score.value), rc

    def decode_vanilla(self, ob):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = self._check(self._L.fv_decode_vanilla(self._h, _p(ob), ob.size, _p(path), ctypes.byref(score)))
        return path, np.float32(score.value), rc

    def decode_checkpoint(self, ob, step=0):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = self._check(self._L.fv_decode_checkpoint(self._h, _p(ob), ob.size, step, _p(path), ctypes.byref(score)))
        return path, np.float32(score.value), rc

    def stats(self):
        s = Stats()
        self._check(self._L.fv_last_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def set_partition(self, rank, nranks):
        self._check(self._L.fv_set_partition(self._h, rank, nranks))

    def comm_init(self, rank, nranks, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._check(self._L.fv_comm_init(self._h, rank, nranks, buf))


def device_count():
    return int(load_library().fv_device_count())


def comm_unique_id():
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    rc = load_library().fv_comm_unique_id(buf)
    if rc != 0:
        raise FlashVitError(rc)
    return buf.raw
