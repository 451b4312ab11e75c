"""Device-resident schedules are not trusted (lh_eval_batch_device): a malformed op must come back as a status code
and NaN for that sample, never as an out-of-bounds access on the GPU.  The reference checks nothing here
(src/PhyloHMM.cpp:421 uses the parsed tree unchecked); this is the C ABI's own contract (include/linearham_amd.h).
Both forms of K1 are exercised: the register-stack form runs behind schedule_stack_check_kernel (fields, stack
discipline, ranks), the cherry-table form (LH_K1_TABLES=1) behind K0c (schedule_check_kernel).  Every clean run is
compared with the numpy oracle too, so that "the neighbours keep the bits of the clean run" means the right bits."""
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
    res = json.loads(r.stdout.strip().splitlines()[-1])
    if mode == "none":       # clean schedules: the kernel form's numbers against the oracle's (north star 1e-6; asserted 1e-12)
        assert len(res["oracle"]) == len(res["ll"]) // repeat
        for got, want in zip(res["ll"], res["oracle"]):
            assert got is not None and abs(got - want) <= 1e-12 * abs(want), (res["form"], got, want)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stack", "tables"])
@pytest.mark.parametrize("mode", ["tip", "kind", "node", "rank", "slot", "popslot", "pushslot", "unbalanced"])
def test_corrupted_device_schedule_gets_a_status_code(mode, form):
    # (the 12-leaf family has one wave per rate and takes the cherry-table form by itself: LH_K1_STACK puts the
    # register-stack kernels, which run behind the wave-per-sample stack check, in its place)
    env = {"LH_K1_TABLES": "1"} if form == "tables" else {"LH_K1_STACK": "1"}
    good = _run("none", env)
    assert good["status"] == "" and good["host_error"] == "" and all(x is not None for x in good["ll"])
    assert good["form"].startswith("ct" if form == "tables" else "w"), good["form"]
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
def test_large_tree_form_checks_fields_and_discipline():
    """A 200-leaf family (tip table 25 KB per rate: the launcher's 'large tree' case, the segmented register-stack
    kernels behind the same wave-per-sample check without ranks): clean schedules match the oracle, a node out of range and
    a pop from the wrong slot are both caught."""
    a = _run("none", {}, n_leaves=200)
    assert a["status"] == "" and a["form"].startswith("seg4"), a["form"]
    for mode in ("node", "popslot"):
        bad = _run(mode, {}, n_leaves=200)
        assert "malformed schedule" in bad["status"] and bad["ll"][2] is None, (mode, bad["status"])
        for i, (x, y) in enumerate(zip(bad["ll"], a["ll"])):
            if i != 2:
                assert x == y, (mode, i, x, y)


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
