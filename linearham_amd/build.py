"""Build the native pieces in-tree (no JIT cache): hipcc for the HIP library, g++ for the C++ host,
gcc for the oracle's C restatement.  `python -m linearham_amd.build` or __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
HIP_SOURCES = ["lh_model.hip", "lh_prune.hip", "lh_forward.hip", "lh_asr.hip", "lh_sample.hip", "lh_capi.hip"]


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(verbose=False, force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "liblinearham_hip.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "lh_device.h"), os.path.join(ROOT, "include", "linearham_amd.h")] + \
        sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc"))  # the generated assembly walk
    if not force and not _stale(out, deps):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC] + srcs + ["-o", out], verbose)
    return out


def build_host(verbose=False, force=False):
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir):
        return None
    srcs = sorted(os.path.join(hostdir, f) for f in os.listdir(hostdir) if f.endswith(".cpp"))
    hdrs = sorted(os.path.join(hostdir, f) for f in os.listdir(hostdir) if f.endswith(".hpp"))
    lib_srcs = [s for s in srcs if not s.endswith("_main.cpp")]
    out = os.path.join(LIB, "liblinearham_host.so")
    deps = srcs + hdrs + [os.path.join(ROOT, "include", "linearham_amd.h")]
    if force or _stale(out, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-I", os.path.join(ROOT, "include"),
              "-I", hostdir] + lib_srcs + ["-o", out, "-L", LIB, "-llinearham_hip", "-Wl,-rpath,$ORIGIN"], verbose)
    exe = os.path.join(LIB, "linearham")
    main = os.path.join(hostdir, "linearham_main.cpp")
    if os.path.exists(main) and (force or _stale(exe, deps)):
        _run(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", hostdir, main,
              "-o", exe, "-L", LIB, "-llinearham_host", "-llinearham_hip", "-Wl,-rpath,$ORIGIN"], verbose)
    return out


def build_oracle(verbose=False, force=False):
    src = os.path.join(ROOT, "oracle", "oracle_kernels.c")
    if not os.path.exists(src):
        return None
    out = os.path.join(ROOT, "oracle", "liboracle_kernels.so")
    if force or _stale(out, [src]):
        _run(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-shared", "-fopenmp", src, "-o", out, "-lm"], verbose)
    return out


def build_all(verbose=False, force=False):
    build_hip(verbose, force)
    build_host(verbose, force)
    build_oracle(verbose, force)


if __name__ == "__main__":
    build_all(verbose=True, force="--force" in sys.argv)
