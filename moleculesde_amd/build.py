"""Build libmsde_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build() and usable as
`python -m moleculesde_amd.build`.  The .so is git-ignored but travels to the GPU box.

Every source is its own translation unit: objects are compiled in parallel (one hipcc per file) and only when the
file or a shared header is newer than its object, then linked into one shared library."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libmsde_hip.so")
SOURCES = ["graph.hip", "gin.hip", "schnet.hip", "cfconv_fused.hip", "cfconv_fused_bwd.hip", "cfconv_pair.hip",
           "sde2d3d.hip", "linear.hip", "gemm_ex.hip", "gemm_rs.hip", "gemm_t2.hip", "gemm_t2_a1.hip", "gemm_t2_a2.hip", "dense_head.hip", "plan.hip", "dd.hip", "norm.hip",
           "contrastive.hip", "optim.hip", "pointwise.hip", "gat_tail.hip", "escore_mol.hip", "escore_mol_bwd.hip"]
HEADERS = [os.path.join(CSRC, "msde_common.h"), os.path.join(CSRC, "gemm_rs.h"), os.path.join(CSRC, "gemm_rs_epi.h"), os.path.join(CSRC, "gemm_t2.h"), os.path.join(CSRC, "escore_mol.h"),
           os.path.join(HERE, "..", "include", "msde_hip.h")]


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"] + \
        os.environ.get("MSDE_HIPCC_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def needs_build():
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return _stale(LIB, srcs + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    flags = _flags()
    stamp = os.path.join(OBJ, "flags.txt")
    old = open(stamp).read() if os.path.exists(stamp) else None
    if old != " ".join(flags):
        force = True
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    workers = int(os.environ.get("MSDE_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        list(ex.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
