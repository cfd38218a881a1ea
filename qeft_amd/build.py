"""Build the gfx950 HIP library in-tree: qeft_amd/lib/libqeft_hip.so.

    python -m qeft_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libqeft_hip.so")
SOURCES = ["capi.hip", "gemv_w4.hip", "gemv_v3.hip", "gemv_v3_plain.hip", "gemv_w3.hip", "gemm_w4.hip", "gemm_ws.hip", "aux_w4.hip", "decode_aux.hip", "oneshot.hip"]
HEADERS = ["qeft_common.h", "gemv_w4_kernel.h", "gemv_w4_mfma.h", "gemv_v3.h", "gemv_v3_dispatch.h", "decode_attn.h", os.path.join("..", "..", "include", "qeft_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        lab = ["-DQEFT_LAB"] if os.environ.get("QEFT_BUILD_LAB") == "1" else []      # tools/gemm_clock_lab.py, QEFT_GEMM_ABL
        # kernarg preload: the leading kernel parameters arrive in SGPRs at wave launch (gemv_v3.h orders its parameters for it)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16",
               "-c", os.path.join(CSRC, s), "-o", obj, *lab, *extra_flags]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(obj)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
