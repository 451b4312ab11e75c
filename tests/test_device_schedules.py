"""Device-resident schedules are not trusted (lh_eval_batch_device): a malformed op must come back as a status code
and NaN for that sample, never as an out-of-bounds access on the GPU.  The reference checks nothing here
(src/PhyloHMM.cpp:421 uses the parsed tree unchecked); this is the C ABI's own contract (include/linearham_amd.h).
Both forms of K1 are exercised: the register-stack form runs behind schedule_ranks_kernel, the cherry-table form
(LH_K1_TABLES=1) behind K0c (schedule_check_kernel)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "device_schedule_worker.py")


def _run(mode, env_extra, n_leaves=12, repeat=1):
    r = subprocess.run([sys.executable, WORKER, mode, str(n_leaves), str(repeat)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stack", "tables"])
@pytest.mark.parametrize("mode", ["tip", "kind", "node", "rank", "slot"])
def test_corrupted_device_schedule_gets_a_status_code(mode, form):
    env = {"LH_K1_TABLES": "1"} if form == "tables" else {}
    good = _run("none", env)
    assert good["status"] == "" and good["host_error"] == "" and all(x is not None for x in good["ll"])
    bad = _run(mode, env)
    # lh_eval_batch (host pointers) refuses the batch whatever the kernel form -- rank bits zeroed (ops built against the
    # round-1 ABI) included: the rank of every op must be the running matrix count
    assert "malformed schedule op" in bad["host_error"], bad
    if mode == "rank" and form == "tables":
        return                        # (on the device the cherry-table form does not read the rank field at all)
    assert "malformed schedule" in bad["status"], bad
    assert bad["second"] == ""                                   # reported once, then cleared
    assert bad["ll"][2] is None                                  # the corrupted sample: NaN
    for i, (a, b) in enumerate(zip(bad["ll"], good["ll"])):
        if i != 2:
            assert a == b, (i, a, b)                             # its neighbours: the bits of the clean run


@pytest.mark.gpu
def test_corrupted_schedule_in_a_larger_tree():
    """A 60-leaf family (more than one two-site wave per rate: the assembly walk of the cherry-table form runs)."""
    for env in ({}, {"LH_K1_TABLES": "1"}):
        bad = _run("tip", env, n_leaves=60)
        assert "malformed schedule" in bad["status"] and bad["ll"][2] is None
        assert all(x is not None for i, x in enumerate(bad["ll"]) if i != 2)


@pytest.mark.gpu
def test_large_tree_forms_agree():
    """A 200-leaf family (tip table 25 KB per rate: the launcher's 'large tree' case) through the segmented
    register-stack kernels (default) and through the cherry-table form with its tip table in the scratch region
    (LH_K1_TIPS_SCRATCH=1, assembly walk lh_prune_walk_asm_s2g.inc): the same log-likelihoods to rounding, and a corrupted
    schedule is caught in both."""
    a = _run("none", {}, n_leaves=200)
    b = _run("none", {"LH_K1_TIPS_SCRATCH": "1"}, n_leaves=200)
    assert a["status"] == "" and b["status"] == ""
    for x, y in zip(a["ll"], b["ll"]):
        assert x is not None and y is not None and abs(x - y) <= 1e-10 * abs(x), (x, y)
    bad = _run("node", {"LH_K1_TIPS_SCRATCH": "1"}, n_leaves=200)
    assert "malformed schedule" in bad["status"] and bad["ll"][2] is None


@pytest.mark.gpu
def test_large_batch_takes_the_thread_per_sample_check():
    """From 12 288 samples on K0c runs a thread per sample instead of a wave per sample (launch_prune): the same batch of
    six samples 2100 times over, cherry-table form, one corrupted schedule -- only that sample is NaN, every copy of the
    others has the bits of the small batch."""
    small = _run("none", {"LH_K1_TABLES": "1"})
    big = _run("tip", {"LH_K1_TABLES": "1"}, repeat=2100)
    assert big["n"] == 12600 and "malformed schedule" in big["status"]
    assert big["nan_at"] == [2] and big["copies_agree"]
    for i, (a, b) in enumerate(zip(big["ll"], small["ll"])):
        if i != 2:
            assert a == b, (i, a, b)
