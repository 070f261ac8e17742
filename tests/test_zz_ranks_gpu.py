"""Row (e) on the one card a test box has: `python bench.py --gpus 2` starts its two ranks itself (the form the driver uses on an
8-GPU node), both ranks run the headline workload -- here on the SAME card, meeting over gloo because RCCL refuses two ranks on
one device (BENCH_SHARED_GPU=1: a rehearsal of launcher, barriers, max over ranks and the summed value, not a measurement).
(Last in the alphabet on purpose: three more processes use the card while it runs.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card(torch_gpu):
    env = dict(os.environ); env["BENCH_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]   # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 1048576 and out["config"]["parallelism"] == "batch-shard x2"
    # whole-job value = both ranks' products over the slowest rank's time
    flops = 2.0 * 32 * 32 * 32 * 1048576 * 2
    assert abs(out["value"] - flops / (out["ms_per_step"] * 1e-3) / 1e9) <= 1e-3 * out["value"]
    assert "cpu_baseline" not in out   # N = 1 only
