"""Build libamdseg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  `python -m spokennlp_amd.build [--force]`

Each `csrc/*.hip` is compiled to its own object under `csrc/build/` (ignored by git; objects are rebuilt when the source, any header
of csrc/ or include/amdseg.h is newer), several at a time, then linked.  A GPU box without hipcc uses the library that travelled."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["gemm.hip", "gemm_dp.hip", "attention.hip", "attention_split.hip", "elementwise.hip", "optim.hip", "gemm_f32.hip", "longformer.hip", "ponet.hip", "ponet_global.hip", "prof.hip", "parity.hip", "heads.hip", "lf_global.hip", "comm.hip", "api.hip"]
OUT = os.path.join(HERE, "libamdseg.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]


def sources_sha():
    """sha256 (16 hex digits) over the kernel sources a measurement depends on: every csrc/*.hip, csrc/*.h and include/amdseg.h, by name.
    profiles/pmc_traffic.json records it (tools/pmc_to_json.py) and bench.py attaches the file's figures only when it still matches."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for f in names:
        h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "amdseg.h"), "rb").read())
    return h.hexdigest()[:16]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True, extra_flags=(), out=OUT):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(HERE, "..", "include", "amdseg.h")]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(srcs + headers):
        return out
    if not os.path.exists(hipcc):
        if os.path.exists(out):
            return out          # GPU box without a toolchain: use the prebuilt library that travelled with the repo
        raise RuntimeError("hipcc not found and no prebuilt libamdseg.so")
    tag = "" if not extra_flags else "_" + "".join(c if c.isalnum() else "_" for c in "".join(extra_flags))
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = _newest(headers)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + tag + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(hdr_t, os.path.getmtime(s)):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + tag + ".o") for s in srcs]
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
