#!/usr/bin/env python3
"""End-to-end run of the `linearham` binary on the configs[2] family: --pipeline on an N-row RevBayes table, then
--asr on the pipeline's output.  Prints wall times (family set-up, I/O, sampling and output included).
usage: python tools/e2e_cli.py [n_rows=16384]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth_family as sf  # noqa: E402

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
fam = os.path.join(tempfile.gettempdir(), "lh_e2e_fam")
if not os.path.exists(os.path.join(fam, "meta.json")):
    sf.generate(sf.Spec(n_samples=256), fam)
lines = open(os.path.join(fam, "trees.tsv")).read().splitlines()
big = os.path.join(fam, "trees_big.tsv")
with open(big, "w") as f:
    f.write(lines[0] + "\n")
    body = lines[1:]
    for i in range(n_rows):
        f.write(body[i % len(body)] + "\n")
exe = os.path.join(ROOT, "linearham_amd", "lib", "linearham")
common = ["--yaml-path", os.path.join(fam, "cluster.yaml"), "--cluster-ind", "0", "--hmm-param-dir",
          os.path.join(fam, "hmm_params")]
out, asr = os.path.join(fam, "lh.tsv"), os.path.join(fam, "asr.trees")
t = time.time()
print("[e2e] starting the pipeline process at wall %.3f" % t, flush=True)
subprocess.check_call([exe, "--pipeline"] + common + ["--input-path", big, "--output-path", out, "--num-rates", "4",
                                                      "--seed", "1"])
t1 = time.time() - t
t = time.time()
subprocess.check_call([exe, "--asr"] + common + ["--input-path", out, "--output-path", asr, "--seed", "1"])
t2 = time.time() - t
n_out = sum(1 for _ in open(out)) - 1
n_asr = sum(1 for _ in open(asr))
print("pipeline: %d rows in %.2f s (%.0f rows/s); asr: %d annotated trees in %.2f s (%.0f trees/s, %.1f MB)" %
      (n_out, t1, n_out / t1, n_asr, t2, n_asr / t2, os.path.getsize(asr) / 1e6))
assert n_out == n_rows and n_asr == n_rows
