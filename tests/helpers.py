"""Shared helpers for comparing HMM-like objects/dicts against the reference goldens."""
import numpy as np

REGIONS = ["vpadding", "vgerm", "vd_junction", "dgerm", "dj_junction", "jgerm", "jpadding"]


def oracle_accessors(h):
    """Flatten an oracle HMM/SimpleHMM/PhyloHMM into the accessor names used by test/test.cpp."""
    out = {"locus": h.locus, "flexbounds": {k: list(v) for k, v in h.flexbounds.items()},
           "relpos": dict(h.relpos), "alphabet": h.alphabet, "msa": h.msa.tolist()}
    for rn in REGIONS:
        R = getattr(h, rn)
        out[rn + "_ggene_ranges"] = {k: list(v) for k, v in R.ggene_ranges.items()}
        out[rn + "_naive_bases"] = list(R.naive_bases)
        out[rn + "_site_inds"] = list(R.site_inds)
        if rn.endswith("germ"):
            out[rn + "_state_strs"] = list(R.state_strs)
            out[rn + "_left_del"] = list(R.left_del)
            out[rn + "_right_del"] = list(R.right_del)
            out[rn + "_germ_inds"] = list(R.germ_inds)
        if rn.endswith("junction"):
            out[rn + "_state_strs"] = list(R.state_strs)
            out[rn + "_del"] = list(R.dels)
            out[rn + "_ggene_types"] = list(R.ggene_types)
            out[rn + "_germ_inds"] = list(R.germ_inds)
    for nm in ["vpadding_transition", "vgerm_vd_junction_transition", "vd_junction_transition",
               "vd_junction_dgerm_transition", "dgerm_dj_junction_transition", "dj_junction_transition",
               "dj_junction_jgerm_transition", "jpadding_transition"]:
        if hasattr(h, nm):
            out[nm] = np.asarray(getattr(h, nm)).tolist()
    return out


def assert_close_struct(got, want, name, rtol=0.0, atol=0.0):
    """Exact for ints/strings; |a-b| <= atol + rtol*|b| for floats (recursively)."""
    if isinstance(want, dict):
        assert isinstance(got, dict) and sorted(got) == sorted(want), (name, got, want)
        for k in want:
            assert_close_struct(got[k], want[k], name + "." + k, rtol, atol)
    elif isinstance(want, (list, tuple)):
        got = list(got)
        assert len(got) == len(want), (name, len(got), len(want))
        for i, (g, w) in enumerate(zip(got, want)):
            assert_close_struct(g, w, "%s[%d]" % (name, i), rtol, atol)
    elif isinstance(want, float) or isinstance(got, float):
        assert abs(float(got) - float(want)) <= atol + rtol * abs(float(want)), (name, got, want)
    else:
        assert got == want, (name, got, want)


def catch_approx(a, b):
    """Catch v1.5.4 Approx (test/catch.hpp:2624-2648): |a-b| < eps*(scale+max(|a|,|b|)), eps=1.19e-5."""
    return abs(a - b) < 1.1920929e-7 * 100 * (1.0 + max(abs(a), abs(b)))


def eigen_is_approx(a, b, prec):
    """Eigen isApprox: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)."""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    return np.sum((a - b) ** 2) <= prec * prec * min(np.sum(a * a), np.sum(b * b))
