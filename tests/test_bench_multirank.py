"""bench.py's own launcher: `python bench.py --gpus N` with no torchrun environment must start N rank
processes itself, shard the tree samples with linearham_amd/sharding.py and print ONE JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_bench_two_ranks_without_torchrun():
    """Two ranks sharing the one GPU of the test box (gloo gather through host memory): n_gpus == 2, every rank's
    samples checked against the CPU oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--preset",
                        "small", "--steps", "2", "--warmup", "1"], env=_clean_env(), capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["backend"] == "gloo"
    assert out["config"]["tree_samples_per_step"] == 2 * out["config"]["tree_samples_per_gpu_per_step"]
    assert out["delta_logl_samples_checked"] >= 4            # samples of both ranks' shards
    assert out["delta_logl_vs_cpu_max_rel"] < 1e-9
    assert out["roofline"]["bound"] == "fp64_valu" and 0 < out["roofline"]["frac"] <= 1.0


@pytest.mark.gpu
def test_bench_two_ranks_on_the_real_family_reuse_the_table():
    """The weak-scaling path of the driver's multi-GPU runs, with the configs[2] family: two ranks (gloo, sharing the
    one GPU), 6144 tree samples each per step out of a table of 6144 rows -- the table is reused (sharding.table_rows'
    rotation: every rank still works through 6144 distinct rows), every rank parses only the rows it evaluates, and
    samples of both shards are checked against the CPU oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--preset",
                        "config2", "--batch", "6144", "--steps", "2", "--warmup", "1", "--no-forward-rate"],
                       env=_clean_env(), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["collective_world_size"] == 2 and c["backend"] == "gloo" and out["scaling"] == "weak"
    assert c["tree_samples_per_step"] == 12288 and c["tree_samples_per_gpu_per_step"] == 6144
    assert c["distinct_tree_samples"] == 6144 and c["distinct_tree_samples_per_gpu"] == 6144
    assert out["delta_logl_samples_checked"] >= 4 and out["delta_logl_vs_cpu_max_rel"] < 1e-9


@pytest.mark.gpu
def test_bench_single_rank_small():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "small", "--steps", "2", "--warmup",
                        "1"], env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["delta_logl_vs_cpu_max_rel"] < 1e-9
    assert out["evals_per_s_with_forward"] > 0 and out["cpu_baseline"]["cores"] >= 1
    # the extra rates measured outside the `value` region: device-side pipeline rows and the ancestral-sequence step
    assert out["pipeline_rows_per_s"] > 0 and out["asr_tree_samples_per_s"] > 0 and out["asr_samples_checked"] == 2


@pytest.mark.gpu
def test_bench_config3_preset():
    """BASELINE.json configs[3] on one rank: 10 000 DISTINCT tree samples of the configs[2] family (NNI-perturbed
    topologies, LogNormal(0, 0.3) branch-length factors), every one evaluated once per step; with N ranks the same
    10 000 are divided (strong scaling).  The first samples are checked against the CPU oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--preset", "config3", "--steps", "2", "--warmup",
                        "1", "--no-cpu-baseline"], env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["scaling"] == "strong" and out["config"]["tree_samples_per_step"] == 10000
    assert out["config"]["distinct_tree_samples"] == 10000 and out["config"]["distinct_tree_samples_per_gpu"] == 10000
    assert out["delta_logl_vs_cpu_max_rel"] < 1e-9 and out["value"] > 0


def test_bench_launcher_fails_loudly_without_gpu():
    """On a box without a GPU the ranks refuse to run (no CPU fallback) and the launcher reports failure."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked tests")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preset", "small", "--steps",
                        "1", "--warmup", "0"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "no CPU fallback" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
