#!/usr/bin/env python3
"""Transcribe the literal known-answer values of the reference's Catch test file into JSON.

The reference's only test binary (test/test.cpp) holds ~600 hand-written literals (state-space
vectors, dense transition matrices, xMSA index arrays, emission vectors, log-likelihoods, sampled
paths).  They are *data*: this script reads that file as text, evaluates the C++ initialiser
lists / Eigen comma-initialisers as plain arithmetic, and writes `reference_goldens.json`.
It also copies the reference's tiny test input files (data/*.yaml, data/*/IG*.yaml, newton.tree)
into tests/golden/data/ -- those are the fixtures the reference's tests run on.

Run only in the build container (the reference tree is not present on the GPU box):
    python tests/golden/make_goldens.py [/root/reference]
"""
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(src):
    return re.sub(r"//[^\n]*", "", src)


def test_cases(src):
    """Yield (name, body) for each TEST_CASE."""
    for m in re.finditer(r'TEST_CASE\("([^"]+)"[^)]*\)\s*\{', src):
        start = m.end()
        depth = 1
        i = start
        while depth:
            c = src[i]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
            i += 1
        yield m.group(1), src[start:i - 1]


def to_py(expr):
    expr = expr.strip()
    expr = re.sub(r"GermlineType::([VDJ])", r'"\1"', expr)
    expr = expr.replace("{", "[").replace("}", "]")
    expr = re.sub(r"\btrue\b", "True", expr)
    expr = re.sub(r"\bfalse\b", "False", expr)
    val = eval(expr, {"__builtins__": {}}, {})
    return mapify(val)


def mapify(v):
    # {{"k", x}, ...} initialises a std::map: turn [[str, x], ...] into a dict.
    if isinstance(v, list) and v and all(
            isinstance(e, list) and len(e) == 2 and isinstance(e[0], str) for e in v):
        return {e[0]: mapify(e[1]) for e in v}
    return v


def parse_body(body):
    """Return a list of sections; each is dict(meta=..., vars=...)."""
    stmts = [s.strip() for s in body.split(";")]
    sections = []
    cur = None
    dims = {}
    init_args = None
    for s in stmts:
        s = " ".join(s.split())
        if not s:
            continue
        m = re.match(r"REQUIRE\(\w+->LogLikelihood\(\) == Approx\((-?[0-9.]+)\)\)", s)
        if m and cur is not None:
            cur["vars"]["loglikelihood"] = float(m.group(1))
            continue
        if s.startswith("REQUIRE") or s.startswith("YAML::Node"):
            continue
        m = re.match(r"\w+->InitializePhyloParameters\( ?newick_path, (.*)\)$", s)
        if m and cur is not None:
            er, pi, alpha, nrates = to_py("[" + m.group(1) + "]")
            cur["meta"].update(er=er, pi=pi, alpha=alpha, num_rates=nrates)
            continue
        # Eigen declaration with dims: "Eigen::MatrixXd name(r, c)" / "(n)" / no dims
        m = re.match(r"Eigen::(\w+) (\w+)(?:\(([^)]*)\))?$", s)
        if m:
            name = m.group(2)
            d = [int(x) for x in m.group(3).split(",")] if m.group(3) else [0]
            dims[name] = d
            if cur is not None and d == [0]:
                cur["vars"][name] = []
            continue
        m = re.match(r"(\w+)\.resize\(([^)]*)\)$", s)
        if m:
            dims[m.group(1)] = [int(x) for x in m.group(2).split(",")]
            if cur is not None and dims[m.group(1)] == [0]:
                cur["vars"][m.group(1)] = []
            continue
        m = re.match(r"(\w+) << (.*)$", s)
        if m:
            name, vals = m.group(1), to_py("[" + m.group(2) + "]")
            d = dims[name]
            if len(d) == 2:
                assert len(vals) == d[0] * d[1], (name, len(vals), d)
                vals = [vals[r * d[1]:(r + 1) * d[1]] for r in range(d[0])]
            else:
                assert len(vals) == d[0], (name, len(vals), d)
            tgt = cur["vars"] if cur is not None else None
            if tgt is None:
                sections.append({"meta": {}, "vars": {}})
                cur = sections[-1]
                tgt = cur["vars"]
            tgt[name] = vals
            continue
        # plain declaration/assignment: "[type] name = value"
        m = re.match(r"(?:[\w:<>, ]+ )?(\w+) = (.*)$", s)
        if m:
            name, val = m.group(1), m.group(2)
            if val.startswith("std::make_shared") or val.startswith("CreateGermlineGeneMap") \
                    or "PtrCast" in val:
                continue
            pyval = to_py(val)
            if name == "yaml_path":
                prev = cur
                cur = {"meta": {"yaml_path": pyval}, "vars": dict(prev["vars"]) if prev else {}}
                if prev:
                    cur["meta"]["hmm_param_dir"] = prev["meta"].get("hmm_param_dir")
                sections.append(cur)
                continue
            if name in ("hmm_param_dir", "newick_path"):
                if cur is None:
                    continue
                cur["meta"][name] = pyval
                continue
            if cur is None:
                sections.append({"meta": {}, "vars": {}})
                cur = sections[-1]
            cur["vars"][name] = pyval
            continue
        if re.match(r"\w+\.reset\(", s) or re.match(r"\w+->InitializePhyloEmission", s) \
                or re.match(r"(Germline|NTInsertion|NPadding) \w+\(", s):
            continue
        raise ValueError("unparsed statement: %r" % s)
    return sections


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = strip_comments(open(os.path.join(ref, "test", "test.cpp")).read())
    out = {"_source": "matsengrp/linearham test/test.cpp (literals only; see make_goldens.py)"}
    for name, body in test_cases(src):
        secs = parse_body(body)
        if name in ("SimpleHMM", "PhyloHMM"):
            for sec in secs:
                key = name + ":" + os.path.basename(sec["meta"]["yaml_path"]).replace(".yaml", "")
                if name == "PhyloHMM":
                    sec["meta"].setdefault("newick_path", "data/newton.tree")
                out[key] = sec
        elif secs:
            merged = {}
            for sec in secs:
                merged.update(sec["vars"])
            out[name] = {"meta": {}, "vars": merged}
    # The phylomd section of PhyloHMM only re-pins the log-likelihood; drop the stale carried vars.
    k = "PhyloHMM:phylo_likelihood_hmm_input"
    out[k]["vars"] = {"loglikelihood": out[k]["vars"]["loglikelihood"]}
    with open(os.path.join(HERE, "reference_goldens.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # fixtures: the reference tests' own input files
    dst = os.path.join(HERE, "data")
    for sub in ("hmm_params", "phylo_likelihood_hmm_params"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        for fn in sorted(os.listdir(os.path.join(ref, "data", sub))):
            shutil.copyfile(os.path.join(ref, "data", sub, fn), os.path.join(dst, sub, fn))
    for fn in ("newton.tree", "phylo_hmm_input.yaml", "phylo_hmm_input_extra.yaml",
               "phylo_likelihood_hmm_input.yaml", "simple_hmm_input.yaml",
               "simple_hmm_input_extra.yaml"):
        shutil.copyfile(os.path.join(ref, "data", fn), os.path.join(dst, fn))
    os.system("chmod -R u+w " + dst)
    print("sections:", [k for k in out if not k.startswith("_")])


if __name__ == "__main__":
    main()
