"""ctypes wrapper of oracle/libfvoracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from the product package (flash_viterbi_amd does not import this module,
and tests/test_boundary.py checks that).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfvoracle.so")

ERRORS = {0: "ok", -1: "bad argument", -2: "out of memory",
          -3: "decoded entry has no finite predecessor (reference: undefined)",
          1: "beam miss: Find_T3_State returned -1 (path holds -1, as the reference prints)"}

_lib = None


def build(force=False):
    src = [os.path.join(HERE, f) for f in ("flashvit_oracle.c", "flashvit_oracle.h", "Makefile")]
    if force or not os.path.isfile(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        res = subprocess.run(["make", "-B", "-C", HERE, "libfvoracle.so"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        # never more OpenMP threads than the CPUs this process may run on (a GPU box's cgroup share)
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        L.fvo_set_threads(min(ncpu, 16))
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.fvo_model_create.restype = vp
        L.fvo_model_create.argtypes = [vp, vp, vp, ci, ci]
        L.fvo_model_destroy.argtypes = [vp]
        L.fvo_full_decode.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        L.fvo_beam_decode.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp]
        L.fvo_full_forward.argtypes = [vp, vp, ci, ci, ci, vp, vp]
        L.fvo_vanilla_decode.argtypes = [vp, vp, ci, vp, vp]
        L.fvo_checkpoint_decode.argtypes = [vp, vp, ci, ci, vp, vp]
        L.fvo_checkpoint_memory_bytes.argtypes = [ci, ci, ci]
        L.fvo_checkpoint_memory_bytes.restype = ctypes.c_longlong
        L.fvo_beam_step_probe.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
        L.fvo_set_threads.argtypes = [ci]
        L.fvo_full_memory_bytes.restype = ctypes.c_longlong
        L.fvo_full_memory_bytes.argtypes = [ci, ci, ci]
        L.fvo_beam_memory_bytes.restype = ctypes.c_longlong
        L.fvo_beam_memory_bytes.argtypes = [ci, ci, ci, ci]
        _lib = L
    return _lib


def set_threads(n):
    return int(lib().fvo_set_threads(n))


class OracleError(RuntimeError):
    def __init__(self, rc):
        super().__init__(f"oracle: {ERRORS.get(rc, 'unknown')} ({rc})")
        self.rc = rc


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleModel:
    """Holds the double log tables for one (A, B, Pi) in float32 as the loader read them."""

    def __init__(self, A, B, Pi):
        self.A = np.ascontiguousarray(A, dtype=np.float32)
        self.B = np.ascontiguousarray(B, dtype=np.float32)
        self.Pi = np.ascontiguousarray(Pi, dtype=np.float32)
        self.K, self.M = self.B.shape
        assert self.A.shape == (self.K, self.K) and self.Pi.shape == (self.K,)
        self._h = lib().fvo_model_create(_p(self.A), _p(self.B), _p(self.Pi), self.K, self.M)
        if not self._h:
            raise MemoryError("fvo_model_create failed")

    def close(self):
        if self._h:
            lib().fvo_model_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def full_decode(self, ob, n_split, check=True):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        T = ob.size
        path = np.empty(T, dtype=np.int32)
        score = ctypes.c_float(0)
        cells = ctypes.c_longlong(0)
        rc = lib().fvo_full_decode(self._h, _p(ob), T, n_split, _p(path), ctypes.byref(score), ctypes.byref(cells))
        if rc < 0 and check:
            raise OracleError(rc)
        return path, np.float32(score.value), cells.value, rc

    def beam_decode(self, ob, n_split, beam, check=True):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        T = ob.size
        path = np.empty(T, dtype=np.int32)
        score = ctypes.c_float(0)
        cells = ctypes.c_longlong(0)
        rc = lib().fvo_beam_decode(self._h, _p(ob), T, n_split, beam, _p(path), ctypes.byref(score), ctypes.byref(cells))
        if rc < 0 and check:
            raise OracleError(rc)
        return path, np.float32(score.value), cells.value, rc

    def vanilla_decode(self, ob, check=True):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = lib().fvo_vanilla_decode(self._h, _p(ob), ob.size, _p(path), ctypes.byref(score))
        if rc < 0 and check:
            raise OracleError(rc)
        return path, np.float32(score.value), rc

    def checkpoint_decode(self, ob, step=0, check=True):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        path = np.empty(ob.size, dtype=np.int32)
        score = ctypes.c_float(0)
        rc = lib().fvo_checkpoint_decode(self._h, _p(ob), ob.size, step, _p(path), ctypes.byref(score))
        if rc < 0 and check:
            raise OracleError(rc)
        return path, np.float32(score.value), rc

    def beam_step_probe(self, hval, hstate, o, blocked):
        """Scores and winning slots of one beam step; blocked=False is the cell-at-a-time restatement."""
        hval = np.ascontiguousarray(hval, dtype=np.float32)
        hstate = np.ascontiguousarray(hstate, dtype=np.int32)
        scr = np.empty(self.K, dtype=np.float32)
        arg = np.empty(self.K, dtype=np.int32)
        rc = lib().fvo_beam_step_probe(self._h, _p(hval), _p(hstate), hval.size, int(o), 1 if blocked else 0, _p(scr), _p(arg))
        if rc:
            raise OracleError(rc)
        return scr, arg

    def full_forward(self, ob, L, R, init_state=-1):
        ob = np.ascontiguousarray(ob, dtype=np.int32)
        row = np.empty(self.K, dtype=np.float32)
        args = np.empty((max(R - L, 0), self.K), dtype=np.int32)
        rc = lib().fvo_full_forward(self._h, _p(ob), L, R, init_state, _p(row), _p(args))
        if rc:
            raise OracleError(rc)
        return row, args


def checkpoint_memory_bytes(K, T, step=0):
    return int(lib().fvo_checkpoint_memory_bytes(K, T, step))


def full_memory_bytes(K, T, n_split):
    return int(lib().fvo_full_memory_bytes(K, T, n_split))


def beam_memory_bytes(K, T, n_split, beam):
    return int(lib().fvo_beam_memory_bytes(K, T, n_split, beam))
