"""Data-parallel Trainer step with 2 ranks on ONE MI355X (gloo on device tensors): SURVEY §8e semantics --
N ranks with identical data reproduce the 1-rank run; ranks with different shards equal a 1-rank run on the mean
gradient -- for the eager step and the hipGraph step (graph -> bucketed all-reduce -> per-bucket Adam).  The RCCL
transport itself is exercised by `bench.py --debug_dp_path` (log under profiles/)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["configs1", "full", "overlap"])
def test_trainer_step_two_ranks_one_gpu(mode):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", {"full": "29541", "overlap": "29542"}.get(mode, "29540"),
           os.path.join(ROOT, "tests", "dp_gpu_worker.py"), ROOT] + ([mode] if mode != "configs1" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert r.stdout.count("RANK0DONE") == 1 and r.stdout.count("RANK1DONE") == 1, r.stdout[-2000:]
