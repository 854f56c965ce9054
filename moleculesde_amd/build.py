"""Build libmsde_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build() and usable as
`python -m moleculesde_amd.build`.  The .so is git-ignored but travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmsde_hip.so")
SOURCES = ["graph.hip", "gin.hip", "schnet.hip", "cfconv_fused.hip", "cfconv_fused_bwd.hip", "sde2d3d.hip", "linear.hip", "gemm_ex.hip", "dense_head.hip", "plan.hip", "dd.hip", "norm.hip", "contrastive.hip", "optim.hip", "pointwise.hip", "gat_tail.hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "msde_common.h"),
                                                        os.path.join(HERE, "..", "include", "msde_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value"] + os.environ.get("MSDE_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
