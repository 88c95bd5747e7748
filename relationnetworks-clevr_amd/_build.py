"""Build the C-ABI HIP library (librn_hip.so) in-tree with hipcc for gfx950.

No JIT cache, no torch cpp_extension: one explicit hipcc invocation so the .so
sits next to the sources and travels with the tree to the GPU box."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librn_hip.so")
SOURCES = ["rn_pair.hip", "rn_gemm.hip", "rn_chain.hip", "rn_chain_rr.hip", "rn_wgrad.hip", "rn_small.hip", "rn_convnorm.hip", "rn_lstm.hip", "rn_conv.hip"]
HEADERS = [os.path.join(CSRC, "rn_common.h"), os.path.join(HERE, "..", "include", "rn_hip.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB + ".tmp"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
