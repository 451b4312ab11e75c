"""Every form of the pruning kernel (K1) against the numpy oracle, by design rather than by accident of shape.

K1 picks a kernel form from the family's shape (lh_prune.hip launch_prune: all rates in one workgroup or not; tip table
whole or a schedule segment at a time; register stack or cherry tables; N-aware tip gather or not; assembly or C++ walk;
stack depth 3 / 4 / 16).  Each test below names the form it means to reach, runs a family that reaches it -- by its own
shape, or pushed there by one of the launcher's test hooks (environment variables read once per process, hence the
subprocess) -- compares log-likelihood (1e-12), rates, xMSA emissions (1e-10), forward arrays (1e-9) and ScaleMatrix counts
(exact) with oracle/linearham_oracle.py, and asserts through lh_family_prune_form that the intended form is the one that ran.

Reference behaviour covered: Partition::TraversalUpdate / LogLikelihood on any topology (src/PhyloHMM.cpp:224-226), N
inside alignment columns encoded 1111 (src/HMM.cpp:69-83, src/utils.cpp:155-164, src/PhyloHMM.cpp:229-235,368-370)."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "forms_worker.py")
HOOKS = ("LH_K1_TABLES", "LH_K1_STACK", "LH_K1_NO_TABLES", "LH_K1_CXX_WALK", "LH_K1_NO_FUSE", "LH_K1_SEGMENTS", "LH_K1_SEG_WAVES",
         "LH_K1_TILE_CAP")


def _run(env_extra, families, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in HOOKS}
    env.update(env_extra)
    r = subprocess.run([sys.executable, WORKER] + list(families), capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=ROOT)
    assert r.returncode == 0, "%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def _expect(report, family, pattern, **conds):
    info = report[family]
    assert re.fullmatch(pattern, info["form"]), (family, info["form"], pattern)
    for k, (op, v) in conds.items():
        assert {"ge": info[k] >= v, "le": info[k] <= v, "gt": info[k] > v}[op], (family, k, info[k], op, v)


# kernel<depth, N-aware>                     register-stack form: w6 / w5 / w4 (all rates in one workgroup), seg4 / seg5
# kernel<depth, N-aware, fused, assembly>    cherry-table form: ct6 / ct5 / ct4
def test_default_forms_without_n():
    rep = _run({}, ["small_igh", "mid60x400", "balanced64", "wide100x600", "wide100x600_r8", "wide100x600_r3"])
    # at most 128 patterns (one wave per rate): the assembly walk over cherry tables; up to 64 patterns in its one-site variant
    _expect(rep, "small_igh", r"ct[456]<4,false,true,true>", n_patterns=("le", 128))
    _expect(rep, "mid60x400", r"ct[456]<4,false,true,true>", n_patterns=("gt", 128))    # two waves per rate, fused (configs[2]'s form)
    # a perfectly balanced 64-leaf tree: five pending siblings, i.e. slots beyond the two register slots live in scratch memory
    _expect(rep, "balanced64", r"ct[456]<16,false,true,true>", max_depth=("ge", 5))
    # more than 256 patterns: three two-site waves per rate, all four rates in one workgroup of twelve waves
    _expect(rep, "wide100x600", r"ct[456]<4,false,true,true>", n_patterns=("gt", 256))
    # the same family with eight rate categories: 24 waves do not fit one workgroup -- a workgroup per (sample, rate), the
    # assembly walk over cherry tables, K2a mixing the rates
    _expect(rep, "wide100x600_r8", r"ct[456]<4,false,false,true>", n_patterns=("gt", 256))
    # ... and with three: nine waves (a workgroup that is not a power of two in size), fused
    _expect(rep, "wide100x600_r3", r"ct[456]<4,false,true,true>", n_patterns=("gt", 256))


def test_default_forms_with_n_inside_columns():
    """Ragged reads and scattered ambiguous bases (tools/synth_family.py ragged / ambiguous): the kN = true instantiations
    -- N tips reading the vector of ones, 25-entry cherry tables, the N-aware assembly walk (a third state plane), the
    N-aware register-stack and segmented kernels."""
    rep = _run({}, ["mixed_small", "mixed_igk", "mixed_120", "mixed_60x400", "mixed_balanced64", "mixed_500"])
    for fam in rep:
        assert rep[fam]["mixed_columns"] > 0, fam
    _expect(rep, "mixed_small", r"ct[456]<4,true,true,true>", n_patterns=("le", 128))    # small family: N-aware assembly walk
    _expect(rep, "mixed_igk", r"ct[456]<4,true,true,true>", n_patterns=("le", 128))
    _expect(rep, "mixed_120", r"seg4<4,true>")                                            # 121 tips: segmented tip table
    _expect(rep, "mixed_60x400", r"ct[456]<4,true,true,true>", n_patterns=("gt", 256))   # twelve waves per workgroup
    _expect(rep, "mixed_balanced64", r"ct[456]<16,true,true,true>", max_depth=("ge", 5))
    _expect(rep, "mixed_500", r"seg4<4,true>")                                            # 11 segments


@pytest.mark.parametrize("hook,expect", [
    # the register-stack form on fused shapes (which by themselves take the cherry-table form): one one-site wave per rate,
    # two two-site waves, twelve waves per workgroup, with and without N
    ({"LH_K1_STACK": "1"}, {"small_igh": r"w[456]<[34],false>", "mixed_small": r"w[456]<[34],true>",
                            "mid60x400": r"w[456]<[34],false>", "wide100x600": r"w[456]<[34],false>",
                            "mixed_60x400": r"w[456]<[34],true>"}),
    # the cherry-table form where the register-stack form would run: large trees (whole tip table in LDS, a workgroup per
    # (sample, rate)); fused shapes take it by themselves
    ({"LH_K1_TABLES": "1"}, {"mixed_120": r"ct[456]<4,true,false,true>", "mixed_500": r"ct[456]<4,true,false,true>",
                             "mid60x400": r"ct[456]<4,false,true,true>"}),
    # its C++ walk instead of the assembly one
    ({"LH_K1_TABLES": "1", "LH_K1_CXX_WALK": "1"}, {"small_igh": r"ct[456]<4,false,true,false>",
                                                   "mixed_small": r"ct[456]<4,true,true,false>",
                                                   "mixed_60x400": r"ct[456]<4,true,true,false>",
                                                   "mid60x400": r"ct[456]<4,false,true,false>",
                                                   "wide100x600_r8": r"ct[456]<4,false,false,false>"}),
    # the same kernels walking the schedule without tables (every cherry its own op)
    ({"LH_K1_NO_TABLES": "1"}, {"small_igh": r"ct[456]<4,false,true,true>", "mid60x400": r"ct[456]<4,false,true,true>",
                                "mixed_small": r"ct[456]<4,true,true,true>"}),
    # one workgroup per (sample, rate), K2a mixing the rates
    ({"LH_K1_NO_FUSE": "1"}, {"small_igh": r"ct[456]<4,false,false,true>", "mid60x400": r"ct[456]<4,false,false,true>",
                              "mixed_small": r"ct[456]<4,true,false,true>", "mixed_60x400": r"ct[456]<4,true,false,true>"}),
    # the large-tree form (tip table a schedule segment at a time) on small trees, both register budgets
    ({"LH_K1_SEGMENTS": "1"}, {"small_igh": r"seg4<4,false>", "mid60x400": r"seg4<4,false>", "mixed_small": r"seg4<4,true>"}),
    ({"LH_K1_SEGMENTS": "1", "LH_K1_SEG_WAVES": "5"}, {"mid60x400": r"seg5<4,false>", "mixed_60x400": r"seg5<4,true>"}),
    # several site tiles per (sample, rate)
    ({"LH_K1_TILE_CAP": "64"}, {"mid60x400": r"\w+<.*>", "mixed_60x400": r"\w+<.*,true.*>"}),
], ids=["stack", "tables", "tables_cxx", "no_tables", "no_fuse", "segments", "segments5", "tiles"])
def test_hooked_forms(hook, expect):
    rep = _run(hook, list(expect))
    for fam, pattern in expect.items():
        _expect(rep, fam, pattern)
