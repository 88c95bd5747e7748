"""Build the C-ABI HIP library (librn_hip.so) in-tree with hipcc for gfx950.

No JIT cache, no torch cpp_extension: explicit hipcc invocations so the .so sits next to the
sources and travels with the tree to the GPU box.  Each source is compiled to its own object under
build/ (git-ignored; only stale objects are rebuilt, in parallel), then linked."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librn_hip.so")
SOURCES = ["rn_pair.hip", "rn_gemm.hip", "rn_chain_rr.hip", "rn_wgrad.hip", "rn_wgrad_blocked.hip", "rn_small.hip", "rn_fphi.hip", "rn_extract.hip", "rn_convnorm.hip", "rn_lstm.hip", "rn_conv.hip"]
HEADERS = [os.path.join(CSRC, "rn_common.h"), os.path.join(HERE, "..", "include", "rn_hip.h"), os.path.join(HERE, "..", "include", "rn_hip_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# RN_DIAG=1: a diagnostics build (timing ablations with wrong results, rn_diag_* entry points) for tools/ -- never the product build
if os.environ.get("RN_DIAG", "0") == "1":
    FLAGS.append("-DRN_DIAG")


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _stale(target, deps):
    t = _mtime(target)
    return t == 0.0 or any(_mtime(d) > t for d in deps)


def _flags_changed() -> bool:
    """The objects on disk were compiled with other flags (a diagnostics build left behind, or the other way round)."""
    try:
        return open(os.path.join(OBJ, "flags.txt")).read() != " ".join(FLAGS)
    except OSError:
        return True


def needs_build() -> bool:
    return _flags_changed() or _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    force = force or _flags_changed()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc] + FLAGS + ["-shared", "-o", LIB + ".tmp"] + [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES])
    os.replace(LIB + ".tmp", LIB)
    with open(os.path.join(OBJ, "flags.txt"), "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
