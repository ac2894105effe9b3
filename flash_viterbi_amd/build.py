"""In-tree builds: explicit compiler invocations, no JIT cache, no setuptools.

  build_host()      gcc   -> flash_viterbi_amd/libfvhost.so    (text/binary I/O, CPU only)
  build_hip()       hipcc -> flash_viterbi_amd/libflashvit.so  (HIP kernels + C-ABI, gfx950)
  build_hip(timing=True)   -> libflashvit_timing.so            (the same with -DFV_TIMING_BUILD, for tools/ only)
  build_programs()  gcc   -> flash_viterbi_amd/src/*_hip       (the C host programs)

hipcc cross-compiles gfx950 without a GPU present; the resulting .so files are
git-ignored but travel to the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
INCLUDE = os.path.join(ROOT, "include")
CSRC = os.path.join(PKG, "csrc")

HOST_LIB = os.path.join(PKG, "libfvhost.so")
HIP_LIB = os.path.join(PKG, "libflashvit.so")

HIP_SOURCES = ["fv_context.hip", "fv_full.hip", "fv_beam.hip", "fv_comm.hip", "fv_schedule.cpp"]
HIP_DEPS = ["fv_internal.h", "fv_layout.h", "fv_device_common.h", "fv_kernels.hip.inc", "fv_beam_kernels.hip.inc", "fv_schedule.h"]
OBJ_DIR = os.path.join(PKG, "_obj")
# The timing build keeps the kernel switches that change results (FV_OPT_DEBUG bits 0, 4, 11, 12: parts of a kernel
# left out to time the rest); tools/ load it, the shipped library refuses those bits (fv_set_option: FV_ERR_ARG).
HIP_TIMING_LIB = os.path.join(PKG, "libflashvit_timing.so")


def _newer(target, sources):
    if not os.path.isfile(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.isfile(s))


def _run(cmd, cwd=None):
    res = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return res.stdout + res.stderr


def build_host(force=False):
    src = [os.path.join(CSRC, "fv_textio.c")]
    deps = src + [os.path.join(INCLUDE, "flashvit_host.h")]
    if not force and _newer(HOST_LIB, deps):
        return HOST_LIB
    _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INCLUDE,
          "-o", HOST_LIB] + src)
    return HOST_LIB


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, timing=False):
    """One object per translation unit (compiled side by side), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    lib = HIP_TIMING_LIB if timing else HIP_LIB
    src = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = src + [os.path.join(CSRC, d) for d in HIP_DEPS] + [os.path.join(INCLUDE, "flashvit.h")]
    if not force and _newer(lib, deps):
        return lib
    objdir = OBJ_DIR + ("_timing" if timing else "")
    os.makedirs(objdir, exist_ok=True)
    # -ffp-contract=off: the trellis cell is "float add, double add, round" exactly as the
    # reference writes it (FLASH_Viterbi_multithread.c:170); an fma would change results.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall",
             "-I", INCLUDE, "-I", CSRC] + (["-DFV_TIMING_BUILD"] + EXTRA_TIMING_FLAGS if timing else [])
    objs = [os.path.join(objdir, os.path.splitext(os.path.basename(s))[0] + ".o") for s in src]

    def cc(pair):
        s, o = pair
        if force or not _newer(o, [s] + deps[len(src):]):
            _run([hipcc_path()] + flags + ["-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=min(len(src), os.cpu_count() or 2)) as ex:
        list(ex.map(cc, zip(src, objs)))
    _run([hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs +
         ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return lib


EXTRA_TIMING_FLAGS = []      # e.g. ["-DFV_REPLAY_PROF"] (tools/replay_prof.py)


PROGRAMS = ["FLASH_Viterbi_hip", "FLASH_BS_Viterbi_hip"]


def build_programs(force=False):
    """Default-config builds of the host programs (run_hip.py re-compiles patched copies)."""
    out = []
    srcdir = os.path.join(PKG, "src")
    for name in PROGRAMS:
        c = os.path.join(srcdir, name + ".c")
        exe = os.path.join(srcdir, name)
        if not os.path.isfile(c):
            continue
        if force or not _newer(exe, [c, os.path.join(INCLUDE, "flashvit.h"),
                                     os.path.join(INCLUDE, "flashvit_host.h")]):
            _run(program_cc(c, exe))
        out.append(exe)
    return out


def program_cc(c_file, exe):
    """gcc command for one host program; mirrors reference src/run.py:54 plus the two libs."""
    return ["gcc", "-g", "-O2", "-pthread", c_file, "-o", exe, "-I", INCLUDE,
            "-L" + PKG, "-lflashvit", "-lfvhost", "-lm",
            "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_programs(force)
