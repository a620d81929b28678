"""Recipes that build the two CPU checkers.  TEST INFRASTRUCTURE ONLY (see oracle/README.md).

  build_port()  oracle/_build/liboracle.so   gcc build of oracle/paged_ops_oracle.c
                                             (our CPU restatement; travels to the GPU box)
  build_ref()   oracle/_ref/_ref_C.so        the REFERENCE's own CPU backend, compiled
                                             unmodified from /root/reference/csrc/cpu/*.cpp
                                             where the sources lie (nothing is copied).
                                             Needs /root/reference, so it is built in the
                                             dev container only; the prebuilt .so travels
                                             with the gpurun snapshot (git-ignored).

The reference extension is given TORCH_EXTENSION_NAME=_ref_C, so its operators register as
torch.ops._ref_C.* / torch.ops._ref_C_cache_ops.* and never collide with the product's
torch.ops._C.* in the same process.  Flags are those of the reference's
cmake/cpu_extension.cmake:15-17,52-66 (AVX512 build, g++ 11 => no -mavx512bf16).
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
PORT_SO = os.path.join(HERE, "_build", "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "_ref_C.so")

REF_SOURCES = ["activation.cpp", "attention.cpp", "cache.cpp", "layernorm.cpp",
               "pos_encoding.cpp", "utils.cpp", "torch_bindings.cpp"]  # cmake/cpu_extension.cmake:96-103


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout[-4000:]))
    return r.stdout


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_port():
    src = os.path.join(HERE, "paged_ops_oracle.c")
    os.makedirs(os.path.dirname(PORT_SO), exist_ok=True)
    if _newer(PORT_SO, [src]):
        _run(["gcc", "-O3", "-std=c11", "-fopenmp", "-march=x86-64-v3", "-ffp-contract=off",
              "-fPIC", "-shared", "-Wall", src, "-o", PORT_SO, "-lm"])
    return PORT_SO


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE, "csrc", "cpu"))


def build_ref():
    """Compile the reference's csrc/cpu backend.  Returns the .so path, or None when
    /root/reference is not present (GPU box: the prebuilt file is used as is)."""
    if not reference_available():
        return REF_SO if os.path.exists(REF_SO) else None
    import torch
    from torch.utils import cpp_extension as ce

    csrc = os.path.join(REFERENCE, "csrc")
    srcs = [os.path.join(csrc, "cpu", s) for s in REF_SOURCES]
    out_dir = os.path.dirname(REF_SO)
    obj_dir = os.path.join(out_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    if not _newer(REF_SO, srcs):
        return REF_SO
    incs = [p for p in ce.include_paths() if "rocm" not in p] + [sysconfig.get_paths()["include"], csrc]
    flags = ["-O3", "-std=c++17", "-fPIC", "-fopenmp", "-DVLLM_CPU_EXTENSION",
             "-mavx512f", "-mavx512vl", "-mavx512bw", "-mavx512dq",
             "-DTORCH_EXTENSION_NAME=_ref_C",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-w"]
    for i in incs:
        flags += ["-I", i]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s).replace(".cpp", ".o"))
        objs.append(o)
        if _newer(o, [s]):
            jobs.append(["g++"] + flags + ["-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_run, jobs))
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    _run(["g++", "-shared", "-fopenmp", "-o", REF_SO] + objs +
         ["-L", torch_lib, "-ltorch", "-ltorch_cpu", "-lc10", "-lnuma"])
    return REF_SO


if __name__ == "__main__":
    print(build_port())
    print(build_ref())
