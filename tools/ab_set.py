"""Alternating A/B of module constants and / or library builds on the bench line, on ONE box, without editing any source.

    python tools/ab_set.py ROUNDS [--full] [--steps N] name1=SET1 name2=SET2 ...
        SET = comma list of  package.module:ATTR=python-literal,  lib:tools/_build/libmsde_x.so,  arg:<bench.py argument>,  --dp
              (empty: defaults)
    e.g. python tools/ab_set.py 3 units= r5order=moleculesde_amd.slabs:WGRAD_UNIT_ORDER=False r5=lib:tools/_build/libmsde_r5split.so

Each run is a child process (`--child`): it applies the setting after importing the modules, runs bench.main() with the
secondary measurements off and prints ms_per_step (headline; with --full the full step).  Replaces the sed-based ab_flag*.sh."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(setting, bench_args):
    sys.path.insert(0, ROOT)
    for item in filter(None, setting.split(",")):
        if item == "--dp":                 # the multi-GPU step structure on one GPU (1-rank RCCL group)
            bench_args = bench_args + ["--debug_dp_path"]
            continue
        if item.startswith("arg:"):        # a bench.py argument, e.g. arg:--score_kernel,arg:mol
            bench_args = bench_args + [item[4:]]
            continue
        if item.startswith("lib:"):
            from moleculesde_amd import _lib
            _lib.LIB_PATH = os.path.join(ROOT, item[4:])
            continue
        mod, rest = item.split(":", 1)
        attr, val = rest.split("=", 1)
        setattr(importlib.import_module(mod), attr, eval(val))
    import bench
    sys.argv = ["bench.py"] + bench_args
    bench.main()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    rounds = int(sys.argv[1])
    rest = sys.argv[2:]
    full = "--full" in rest
    steps = "300"
    if "--steps" in rest:
        steps = rest[rest.index("--steps") + 1]
    sets = [a.split("=", 1) for a in rest if "=" in a and not a.startswith("--")]
    bench_args = ["--no_cpu_baseline", "--no_configs45", "--no_pipeline", "--no_config2", "--steps", steps] + (["--full"] if full else [])
    res = {n: [] for n, _ in sets}
    for r in range(rounds):
        for name, setting in sets:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", setting] + bench_args, cwd=ROOT,
                                 capture_output=True, text=True)
            try:
                j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
                res[name].append(j["ms_per_step"])
                print("[%s] %.3f ms  tail %s  fwd %s" % (name, j["ms_per_step"], j.get("tail_us"),
                                                        (j.get("roofline_forward_schnet_sde2d3d") or {}).get("ms")), flush=True)
            except Exception:
                print("[%s] FAILED rc=%d\n%s" % (name, out.returncode, out.stderr[-1500:]), flush=True)
    for n, v in res.items():
        if v:
            print("%-12s min %.3f  median %.3f  max %.3f  (%d runs)" % (n, min(v), sorted(v)[len(v) // 2], max(v), len(v)))
