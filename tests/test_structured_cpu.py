"""CPU checks of the C-ABI descriptor semantics: the structured junction tables + rooted-at-naive
schedule (what the HIP kernels consume) must reproduce the dense reference algorithm (oracle)."""
import os

import numpy as np
import pytest

from oracle import linearham_oracle as orc
from tests import desc_builder as db

CASES = [("phylo_hmm_input", "hmm_params", 4), ("phylo_hmm_input_extra", "hmm_params", 4),
         ("phylo_likelihood_hmm_input", "phylo_likelihood_hmm_params", 1)]
ER, PI, ALPHA = [1.0] * 6, [0.17, 0.19, 0.25, 0.39], 1.0


def _load_lib():
    import linearham_amd
    return linearham_amd.load_library()


@pytest.mark.parametrize("case,params,R", CASES)
def test_structured_matches_dense(data_dir, case, params, R):
    ll, h = orc.phylo_loglik(os.path.join(data_dir, case + ".yaml"), os.path.join(data_dir, params),
                             os.path.join(data_dir, "newton.tree"), ER, PI, ALPHA, R)
    desc = db.build_family_desc(h)
    children, root, brlen = db.tree_arrays(h.tree, h.xmsa_labels)
    ops, depth = _load_lib().schedule_tree(h.tree.n_tips, children, root)
    em = db.emulate_prune(desc, h.tree.n_tips, ops, brlen, ER, PI, h.sr)
    np.testing.assert_allclose(em, h.xmsa_emission, rtol=1e-12)
    ll2 = db.emulate_forward(desc, h.xmsa_emission)
    assert abs(ll2 - ll) < 1e-11 * abs(ll)
    ll3 = db.emulate_forward(desc, em)
    assert abs(ll3 - ll) < 1e-11 * abs(ll)


def test_schedule_tree_properties():
    """Random trees: the schedule is a valid post-order, uses every inner node once and its stack
    depth stays within log2(T)."""
    lib = _load_lib()
    rng = np.random.default_rng(5)
    for T in [3, 4, 5, 8, 33, 101, 257]:
        for _ in range(5):
            children, root = random_rooted_tree(T, rng)
            ops, depth = lib.schedule_tree(T, children, root)
            assert depth <= max(0, int(np.ceil(np.log2(T))))
            check_schedule(T, children, root, ops, depth)


def random_rooted_tree(T, rng):
    """Random binary tree over tips 1..T-1 rooted at the inner node adjacent to naive (tip 0)."""
    nodes = list(range(1, T))
    nxt = T
    children = np.zeros((T - 2, 2), dtype=np.int32)
    while len(nodes) > 1:
        i, j = rng.choice(len(nodes), size=2, replace=False)
        a, b = nodes[i], nodes[j]
        children[nxt - T] = (a, b)
        nodes = [x for k, x in enumerate(nodes) if k not in (i, j)] + [nxt]
        nxt += 1
    return children.ravel(), nodes[0]


def check_schedule(T, children, root, ops, depth):
    done = set()
    stack = {}
    acc = None
    for op in ops:
        kind, push = op[0] & 15, bool(op[0] & 16)
        if push:
            assert acc is not None and op[3] not in stack and 0 <= op[3] < depth
            stack[op[3]] = acc
        if kind == 0:
            a, b = op[1], op[2]
            assert 1 <= a < T and 1 <= b < T
            if acc is not None and not push:
                raise AssertionError("cherry would clobber a live accumulator")
            kids = {a, b}
        elif kind == 1:
            assert 1 <= op[1] < T and acc == op[2]
            kids = {op[1], op[2]}
        else:
            assert stack.pop(op[3]) == op[1] and acc == op[2]
            kids = {op[1], op[2]}
        parent = [v for v in range(T, 2 * T - 2)
                  if set(children[2 * (v - T):2 * (v - T) + 2].tolist()) == kids]
        assert len(parent) == 1 and parent[0] not in done
        done.add(parent[0])
        acc = parent[0]
        # a finished subtree whose parent is computed later must stay reachable (acc or stack)
    assert acc == root and not stack and len(done) == T - 2


def test_schedule_rejects_malformed():
    lib = _load_lib()
    with pytest.raises(RuntimeError):
        lib.schedule_tree(4, [1, 2, 4, 4], 5)      # node twice / tip missing
    with pytest.raises(RuntimeError):
        lib.schedule_tree(4, [1, 2, 4, 3], 3)      # root is a tip
    with pytest.raises(RuntimeError):
        lib.schedule_tree(2, [], 2)


def test_library_exports_every_declared_symbol():
    """-m 'not gpu' contract: the C-ABI library loads and exports every symbol of include/*.h."""
    import re
    import linearham_amd
    from linearham_amd import capi
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                            "linearham_amd.h")).read()
    declared = set(re.findall(r"\b(lh_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = linearham_amd.load_library().lib
    for name in declared:
        assert hasattr(lib, name), name


def test_generated_assembly_walk_is_reproducible():
    """linearham_amd/csrc/lh_prune_walk_asm_s{2,1}[n].inc and their clobber lists are GENERATED text (tools/gen_walk_asm.py): the
    committed files must be what the committed generator writes, byte for byte -- a walk edited by hand, or a generator
    edited without regenerating, would leave the product library out of step with its source (build.py lists the .inc
    files among the library's dependencies)."""
    import os
    from tools import gen_walk_asm as g
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "linearham_amd", "csrc")
    names = []
    for S in (2, 1):
        for n_aware in (False, True):
            name = "lh_prune_walk_asm_s%d%s.inc" % (S, "n" if n_aware else "")
            body, clob, _ = g.render(S, n_aware)
            assert open(os.path.join(csrc, name)).read() == body, name
            assert open(os.path.join(csrc, "lh_prune_walk_clobbers_s%d.inc" % S)).read() == clob
            names.append(name)
        names.append("lh_prune_walk_clobbers_s%d.inc" % S)
    assert sorted(f for f in os.listdir(csrc) if f.endswith(".inc")) == sorted(names)
    from linearham_amd import build
    import inspect
    assert ".inc" in inspect.getsource(build.build_hip)
