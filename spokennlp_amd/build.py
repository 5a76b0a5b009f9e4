"""Build libamdseg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  `python -m spokennlp_amd.build`"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm_dp.hip", "attention.hip", "attention_split.hip", "elementwise.hip", "optim.hip", "gemm_f32.hip", "longformer.hip", "ponet.hip", "ponet_global.hip", "prof.hip", "parity.hip", "heads.hip", "lf_global.hip", "api.hip"]
OUT = os.path.join(HERE, "libamdseg.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + \
        [os.path.join(HERE, "..", "include", "amdseg.h")]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(deps):
        return OUT
    if not os.path.exists(hipcc):
        if os.path.exists(OUT):
            return OUT          # GPU box without a toolchain: use the prebuilt library that travelled with the repo
        raise RuntimeError("hipcc not found and no prebuilt libamdseg.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-Wno-unused-result"] + srcs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
