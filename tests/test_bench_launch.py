"""bench.py --gpus 2 without an outer launcher: the parent spawns the ranks itself (before importing torch), the ranks
rendezvous on 127.0.0.1 and rank 0 prints exactly one JSON line.  Runs on CPU over gloo with bench.py's stand-in
workload (--rehearse-cpu): the spawn / rendezvous / bucketed all-reduce / max-over-ranks / JSON path is the same one
the GPU run takes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--rehearse-cpu"] + extra,
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_bench_self_launches_two_ranks():
    p, lines = _run(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 2 * rec["config"]["per_gpu_batch"]
    # the wrapper broadcast rank 0's weights and every rank applied the same summed gradients
    assert rec["config"]["max_weight_spread_across_ranks"] == 0.0
    assert rec["rehearsal"] is True


def test_bench_single_rank_line_has_the_same_keys():
    p, lines = _run(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec
    assert rec["n_gpus"] == 1


def test_bench_child_failure_propagates():
    # a rank count that disagrees with --gpus must end non-zero, not hang
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
