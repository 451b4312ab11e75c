"""ctypes binding of the C++ host library (liblinearham_host.so): linearham's HMM / SimpleHMM /
PhyloHMM class surface.  Used by tests/ and bench.py; no numerics live here."""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def host_library_path():
    # LH_LIB_DIR: timing experiments load a variant build from its own directory (tools/build_variant.sh) instead of
    # overwriting the product library in place
    return os.path.join(os.environ.get("LH_LIB_DIR") or os.path.join(_HERE, "lib"), "liblinearham_host.so")


def load_host():
    global _LIB
    if _LIB is None:
        p = host_library_path()
        if not os.path.exists(p):
            raise RuntimeError("host library %s is missing: run __graft_entry__.build()" % p)
        lib = C.CDLL(p)
        lib.lhh_last_error.restype = C.c_char_p
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError(load_host().lhh_last_error().decode())


def germline_json(path, gtype):
    out = C.c_char_p()
    _check(load_host().lhh_germline_json(path.encode(), C.c_char(gtype.encode()), C.byref(out)))
    return json.loads(out.value.decode())


def newick_roundtrip(newick, labels):
    """Parse a Newick string as the host does and return the output table's tree column
    (PhyloHMM::WriteOutputLine, src/PhyloHMM.cpp:299-300).  No device needed."""
    out = C.c_char_p()
    _check(load_host().lhh_newick_roundtrip(newick.encode(), "\n".join(labels).encode(), C.byref(out), None, None,
                                            None))
    return out.value.decode()


def newick_arrays(newick, labels):
    """The host's rooted-at-naive arrays of a Newick string: (children [(T-2)*2], root, brlen [2T-2])."""
    import numpy as np
    T = len(labels)
    children = np.zeros(2 * (T - 2), dtype=np.int32)
    brlen = np.zeros(2 * T - 2)
    root = C.c_int32()
    out = C.c_char_p()
    _check(load_host().lhh_newick_roundtrip(newick.encode(), "\n".join(labels).encode(), C.byref(out),
                                            children.ctypes.data_as(C.POINTER(C.c_int32)),
                                            brlen.ctypes.data_as(C.POINTER(C.c_double)), C.byref(root)))
    return children, root.value, brlen


class _HMM:
    def __init__(self, handle):
        self.h = handle
        self.lib = load_host()

    def close(self):
        if self.h:
            self.lib.lhh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dump(self, what):
        """what: bit 0 state space + transitions, bit 1 forward arrays, bit 2 sample, bit 3 xMSA."""
        out = C.c_char_p()
        _check(self.lib.lhh_dump_json(self.h, what, C.byref(out)))
        return json.loads(out.value.decode())

    def log_likelihood(self):
        v = C.c_double()
        _check(self.lib.lhh_loglikelihood(self.h, C.byref(v)))
        return v.value

    def sample_naive_sequence(self):
        out = C.c_char_p()
        _check(self.lib.lhh_sample(self.h, C.byref(out)))
        return out.value.decode()


class SimpleHMM(_HMM):
    def __init__(self, yaml_path, cluster_ind, hmm_param_dir, seed):
        h = C.c_void_p()
        _check(load_host().lhh_simple_create(yaml_path.encode(), cluster_ind, hmm_param_dir.encode(), seed,
                                             C.byref(h)))
        super().__init__(h)


class PhyloHMM(_HMM):
    def __init__(self, yaml_path, cluster_ind, hmm_param_dir, seed):
        h = C.c_void_p()
        _check(load_host().lhh_phylo_create(yaml_path.encode(), cluster_ind, hmm_param_dir.encode(), seed,
                                            C.byref(h)))
        super().__init__(h)

    def initialize_phylo_parameters(self, newick, er, pi, alpha, num_rates, is_path=True):
        er = (C.c_double * 6)(*er)
        pi = (C.c_double * 4)(*pi)
        f = self.lib.lhh_phylo_init_parameters if is_path else self.lib.lhh_phylo_init_parameters_str
        _check(f(self.h, newick.encode(), er, pi, C.c_double(alpha), num_rates))

    def initialize_phylo_emission(self):
        _check(self.lib.lhh_phylo_init_emission(self.h))

    def set_extended_range(self, on=True):
        _check(self.lib.lhh_phylo_set_extended_range(self.h, int(on)))

    def sample_states_with_words(self, words):
        """(device states, host states) of one SampleNaiveSequence whose engine outputs are `words` (test entry)."""
        w = np.ascontiguousarray(words, dtype=np.uint32)
        d = np.zeros(4096, dtype=np.int32)
        s = np.zeros(4096, dtype=np.int32)
        n = C.c_int()
        self.lib.lhh_phylo_sample_words.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.POINTER(C.c_int)]
        _check(self.lib.lhh_phylo_sample_words(self.h, w.ctypes.data, len(w), d.ctypes.data, s.ctypes.data, 4096,
                                               C.byref(n)))
        return d[:n.value].copy(), s[:n.value].copy()

    def run_pipeline(self, input_path, output_path, num_rates):
        _check(self.lib.lhh_run_pipeline(self.h, input_path.encode(), output_path.encode(), num_rates))

    def run_asr(self, input_path, output_path, seed):
        lib = load_host()
        lib.lhh_run_asr.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64]
        _check(lib.lhh_run_asr(self.h, input_path.encode(), output_path.encode(), seed))

    def sizes(self):
        v = [C.c_int() for _ in range(8)]
        _check(self.lib.lhh_phylo_sizes(self.h, *[C.byref(x) for x in v]))
        keys = ["n_tips", "n_sites", "n_xmsa", "s_vd", "s_dj", "w_vd", "w_dj", "g_total"]
        return {k: x.value for k, x in zip(keys, v)}

    def set_devices(self, devices):
        """The HIP devices run_pipeline deals the table's rows to (before the first evaluation)."""
        arr = (C.c_int * len(devices))(*devices)
        _check(self.lib.lhh_phylo_set_devices(self.h, arr, len(devices)))

    def flatten_tsv(self, tsv_path, n, need_family=True, rows=None):
        """Device-ready inputs for lh_eval_batch_device: n samples taken cyclically from the table, or -- rows given --
        the table rows rows[0..n) in that order (only those are parsed: a rank flattens what it evaluates).
        Returns dict(ops, brlen, er, pi, alpha, n_tips, max_depth, n_rows, family)."""
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int64)
            n = len(rows)
        n_tips, depth, n_rows = C.c_int(), C.c_int(), C.c_int()
        fam = C.c_void_p()
        flag = C.c_int(1 if need_family else 0)
        T = self.sizes()["n_tips"]
        ops = np.zeros((n, T - 2, 4), dtype=np.int32)
        brlen = np.zeros((n, 2 * T - 2))
        er, pi, alpha = np.zeros((n, 6)), np.zeros((n, 4)), np.zeros(n)

        def p(a, t):
            return a.ctypes.data_as(C.POINTER(t))
        if rows is not None:
            _check(self.lib.lhh_phylo_flatten_tsv_rows(self.h, tsv_path.encode(), n, p(rows, C.c_int64), p(ops, C.c_int32),
                                                       p(brlen, C.c_double), p(er, C.c_double), p(pi, C.c_double),
                                                       p(alpha, C.c_double), C.byref(n_tips), C.byref(depth),
                                                       C.byref(n_rows), flag, C.byref(fam)))
        else:
            _check(self.lib.lhh_phylo_flatten_tsv(self.h, tsv_path.encode(), n, p(ops, C.c_int32),
                                                  p(brlen, C.c_double), p(er, C.c_double), p(pi, C.c_double),
                                                  p(alpha, C.c_double), C.byref(n_tips), C.byref(depth),
                                                  C.byref(n_rows), flag, C.byref(fam)))
        assert n_tips.value == T
        return dict(ops=ops, brlen=brlen, er=er, pi=pi, alpha=alpha, n_tips=T, max_depth=depth.value,
                    n_rows=n_rows.value, family=fam.value)
