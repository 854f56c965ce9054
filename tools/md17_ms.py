"""config 5 only: ms per MD17 force fine-tuning step (hipGraph replay), as bench.py measures it."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no_cpu_baseline", "--steps", "5", "--warmup", "2", "--stream", "0"],
                     capture_output=True, text=True).stdout.strip().splitlines()[-1]
j = json.loads(out)
print("md17 ms/step", j["config5_md17"]["ms_per_step"], "sampler s", j["config4_sampler"]["seconds_per_trajectory_batch"])
