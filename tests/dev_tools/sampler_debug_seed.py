"""Where the device sampler (K4) and the host sampler (HMM::SampleRow) part on a family of random_sweep_pipeline.py --many: the
first row's states under the std::mt19937 stream RunPipeline gives it, device beside host, and the first index at which they
differ.  usage (GPU box, repo root): python tests/dev_tools/sampler_debug_seed.py seed [seed ...]"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linearham_amd import host  # noqa: E402
from tools import synth_family as sf  # noqa: E402


def mt19937_words(seed, n):
    """std::mt19937(seed) outputs (init_genrand seeding)."""
    mt = [0] * 624
    mt[0] = seed & 0xffffffff
    for i in range(1, 624):
        mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xffffffff
    out, idx = [], 624
    while len(out) < n:
        if idx >= 624:
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7fffffff)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
            idx = 0
        y = mt[idx]
        idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9d2c5680
        y ^= (y << 15) & 0xefc60000
        y ^= y >> 18
        out.append(y & 0xffffffff)
    return np.array(out, dtype=np.uint32)


for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(seed)
    locus = ["igh", "igh", "igk", "igl"][int(rng.integers(4))]
    kw = dict(locus=locus, seed=seed, n_samples=5, n_leaves=int(rng.integers(3, 40)), n_v=int(rng.integers(1, 9)),
              n_j=int(rng.integers(1, 6)), ragged=int(rng.choice([0, 0, 4, 10])), ambiguous=float(rng.choice([0.0, 0.0, 0.01, 0.05])),
              divergence=float(rng.choice([0.0, 0.05, 0.3])))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(1, 6))
    kw.update(n_samples=3, n_leaves=int(rng.integers(3, 12)), n_v=int(rng.integers(60, 320)), n_j=int(rng.integers(4, 40)))
    if locus == "igh":
        kw["n_d"] = int(rng.integers(10, 80))
    R = int(rng.choice([1, 3, 4]))
    rng_seed = int(rng.integers(0, 1000))
    out = tempfile.mkdtemp(prefix="lh_smpdbg_")
    try:
        sf.generate(sf.Spec.small(**kw), out)
        yaml_path, pdir, tsv = (os.path.join(out, x) for x in ("cluster.yaml", "hmm_params", "trees.tsv"))
        h = host.PhyloHMM(yaml_path, 0, pdir, rng_seed)
        r = sf.read_trees_tsv(tsv)[0]
        h.initialize_phylo_parameters(r["tree"], r["er"], r["pi"], r["alpha"], R, is_path=False)
        h.initialize_phylo_emission()
        ll = h.log_likelihood()
        fw = h.dump(2)
        jg = np.asarray(fw.get("jgerm_forward", []), dtype=float)
        print("  log-likelihood", ll, " jgerm_forward: n", jg.size, "nonzero", int((jg != 0).sum()), "nan", int(np.isnan(jg).sum()),
              "inf", int(np.isinf(jg).sum()), "min/max", (float(np.nanmin(jg)), float(np.nanmax(jg))) if jg.size else None, flush=True)
        print("  jgerm_forward", jg.tolist(), flush=True)
        words = mt19937_words(rng_seed, 600)
        print("  first words", words[:4].tolist(), flush=True)
        dev, ref = h.sample_states_with_words(words)
        diff = np.nonzero(dev != ref)[0]
        print("seed", seed, locus, "R", R, "engine seed", rng_seed, "states", len(dev), "first difference at", diff[:5].tolist(), flush=True)
        if diff.size:
            i = int(diff[0])
            print("  device", dev[max(0, i - 3):i + 4].tolist(), " host", ref[max(0, i - 3):i + 4].tolist(), flush=True)
    finally:
        shutil.rmtree(out, ignore_errors=True)
