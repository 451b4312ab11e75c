"""ctypes binding of include/linearham_amd.h (liblinearham_hip.so).  No numerics live here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_f64p = C.POINTER(C.c_double)


class _Segments(C.Structure):
    _fields_ = [("n_genes", C.c_int32), ("offsets", c_i32p), ("xmsa_inds", c_i32p)]


class _Junction(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_left", C.c_int32), ("n_right", C.c_int32),
                ("enter_trans", c_f64p), ("enter_lo", c_f64p), ("left_trans", c_f64p),
                ("left_lo", c_f64p), ("left_xmsa", c_i32p), ("right_gp_nli", c_f64p),
                ("right_ntt", c_f64p), ("right_nlo", c_f64p), ("right_trans", c_f64p),
                ("right_gp_li", c_f64p), ("right_xmsa", c_i32p), ("nti_xmsa", c_i32p),
                ("exit_nlo", c_f64p), ("exit_trans", c_f64p), ("exit_gp_li", c_f64p)]


class _FamilyDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("has_d", C.c_int32), ("n_seqs", C.c_int32),
                ("n_sites", C.c_int32), ("msa", c_u8p), ("n_xmsa", C.c_int32),
                ("xmsa_site", c_i32p), ("xmsa_naive_base", c_u8p),
                ("vpadding", _Segments), ("vgerm", _Segments), ("dgerm", _Segments),
                ("jgerm", _Segments), ("jpadding", _Segments),
                ("vgerm_gene_prob", c_f64p), ("vpadding_transition", c_f64p),
                ("vgerm_trans_prod", c_f64p), ("jpadding_transition", c_f64p),
                ("vd", _Junction), ("dj", _Junction)]


class _EvalOutputs(C.Structure):
    _fields_ = [("rates", c_f64p), ("xmsa_emission", c_f64p), ("forward", c_f64p),
                ("scaler_counts", c_i32p)]


EXPORTS = ["lh_last_error", "lh_device_count", "lh_family_create", "lh_family_destroy",
           "lh_forward_size", "lh_scaler_size", "lh_family_info", "lh_family_consensus_sets", "lh_schedule_tree", "lh_eval_batch",
           "lh_eval_batch_device", "lh_forward_batch", "lh_asr_batch", "lh_asr_batch_device",
           "lh_profile_enable", "lh_profile_read", "lh_asr_profile_read", "lh_family_set_extended_range", "lh_warmup", "lh_host_alloc", "lh_host_free", "lh_family_set_sampler",
           "lh_sample_words", "lh_sample_states", "lh_eval_sample_batch", "lh_set_device", "lh_family_status",
           "lh_eval_sample_batch_device", "lh_family_prune_form"]


def library_path():
    # LH_LIB_DIR: timing experiments load a variant build from its own directory (tools/build_variant.sh) instead of
    # overwriting the product library in place
    return os.path.join(os.environ.get("LH_LIB_DIR") or os.path.join(_HERE, "lib"), "liblinearham_hip.so")


class HipLibrary:
    def __init__(self, path=None):
        path = path or library_path()
        if not os.path.exists(path):
            raise RuntimeError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (there is no CPU fallback)" % path)
        self.lib = lib = C.CDLL(path)
        lib.lh_last_error.restype = C.c_char_p
        lib.lh_device_count.restype = C.c_int
        lib.lh_family_create.argtypes = [C.POINTER(_FamilyDesc), C.POINTER(C.c_void_p)]
        lib.lh_family_destroy.argtypes = [C.c_void_p]
        lib.lh_family_destroy.restype = None
        lib.lh_forward_size.argtypes = [C.c_void_p]
        lib.lh_forward_size.restype = C.c_int64
        lib.lh_scaler_size.argtypes = [C.c_void_p]
        lib.lh_scaler_size.restype = C.c_int64
        lib.lh_family_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.lh_family_info.restype = C.c_int
        lib.lh_family_consensus_sets.argtypes = [C.c_void_p]
        lib.lh_family_consensus_sets.restype = C.c_int
        lib.lh_family_prune_form.argtypes = [C.c_void_p]
        lib.lh_family_prune_form.restype = C.c_char_p
        lib.lh_schedule_tree.argtypes = [C.c_int32, c_i32p, C.c_int32, c_i32p, c_i32p]
        lib.lh_eval_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_f64p, c_f64p,
                                      c_f64p, c_f64p, C.c_int32, c_f64p, C.POINTER(_EvalOutputs)]
        lib.lh_eval_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.POINTER(_EvalOutputs), C.c_void_p]
        lib.lh_forward_batch.argtypes = [C.c_void_p, C.c_int32, c_f64p, c_f64p, C.POINTER(_EvalOutputs)]
        lib.lh_asr_batch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_f64p, c_f64p, c_f64p,
                                     c_f64p, C.c_int32, c_u8p, C.c_uint64, C.c_uint64, c_u8p, c_u8p]
        lib.lh_asr_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64,
                                            C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.lh_asr_profile_read.argtypes = [C.c_void_p, c_f64p, C.POINTER(C.c_int64)]
        lib.lh_profile_enable.argtypes = [C.c_void_p, C.c_int]
        lib.lh_family_set_extended_range.argtypes = [C.c_void_p, C.c_int]
        lib.lh_profile_read.argtypes = [C.c_void_p, c_f64p, c_f64p, c_f64p, C.POINTER(C.c_int64)]
        if hasattr(lib, "lh_eval_sample_batch_device"):
            lib.lh_eval_sample_batch_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + \
                [C.c_int32] + [C.c_void_p] * 5
            lib.lh_sample_words.argtypes = [C.c_void_p]
            lib.lh_sample_states.argtypes = [C.c_void_p]
        if hasattr(lib, "lh_set_device"):      # (absent from round-2 builds loaded through LH_LIB_DIR for comparisons)
            lib.lh_set_device.argtypes = [C.c_int32]
            lib.lh_family_status.argtypes = [C.c_void_p]

    def error(self):
        return self.lib.lh_last_error().decode()

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("linearham_hip: " + self.error())

    def device_count(self):
        return self.lib.lh_device_count()

    def schedule_tree(self, n_tips, children, root):
        """children: int32 [(T-2)*2]; returns (ops [T-2,4] int32, max_depth)."""
        children = np.ascontiguousarray(children, dtype=np.int32).ravel()
        ops = np.zeros((max(n_tips - 2, 0), 4), dtype=np.int32)
        depth = C.c_int32(0)
        self.check(self.lib.lh_schedule_tree(n_tips, children.ctypes.data_as(c_i32p), root,
                                             ops.ctypes.data_as(c_i32p), C.byref(depth)))
        return ops, depth.value


_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        _LIB = HipLibrary()
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Segments:
    def __init__(self, offsets, xmsa_inds):
        self.offsets = _i32(offsets)
        self.xmsa_inds = _i32(xmsa_inds)

    def c(self):
        return _Segments(len(self.offsets) - 1, self.offsets.ctypes.data_as(c_i32p),
                         self.xmsa_inds.ctypes.data_as(c_i32p))


class JunctionTables:
    F64 = ["enter_trans", "enter_lo", "left_trans", "left_lo", "right_gp_nli", "right_ntt", "right_nlo",
           "right_trans", "right_gp_li", "exit_nlo", "exit_trans", "exit_gp_li"]
    I32 = ["left_xmsa", "right_xmsa", "nti_xmsa"]

    def __init__(self, n_rows, n_left, n_right, **arrays):
        self.n_rows, self.n_left, self.n_right = n_rows, n_left, n_right
        for k in self.F64:
            setattr(self, k, _f64(arrays[k]))
        for k in self.I32:
            setattr(self, k, _i32(arrays[k]))

    def c(self):
        j = _Junction()
        j.n_rows, j.n_left, j.n_right = self.n_rows, self.n_left, self.n_right
        for k in self.F64:
            setattr(j, k, getattr(self, k).ctypes.data_as(c_f64p))
        for k in self.I32:
            setattr(j, k, getattr(self, k).ctypes.data_as(c_i32p))
        return j


class FamilyDesc:
    """Host-side arrays of lh_family_desc (keeps the numpy buffers alive)."""

    def __init__(self, has_d, msa, xmsa_site, xmsa_naive_base, vpadding, vgerm, dgerm, jgerm, jpadding,
                 vgerm_gene_prob, vpadding_transition, vgerm_trans_prod, jpadding_transition, vd, dj,
                 n_xmsa=None):
        self.has_d = int(has_d)
        self.msa = np.ascontiguousarray(msa, dtype=np.uint8)
        self.xmsa_site = _i32(xmsa_site)
        self.xmsa_naive_base = np.ascontiguousarray(xmsa_naive_base, dtype=np.uint8)
        self.n_xmsa = int(n_xmsa if n_xmsa is not None else len(self.xmsa_site))
        self.vpadding, self.vgerm, self.dgerm, self.jgerm, self.jpadding = vpadding, vgerm, dgerm, jgerm, jpadding
        self.vgerm_gene_prob = _f64(vgerm_gene_prob)
        self.vpadding_transition = _f64(vpadding_transition)
        self.vgerm_trans_prod = _f64(vgerm_trans_prod)
        self.jpadding_transition = _f64(jpadding_transition)
        self.vd, self.dj = vd, dj

    def c(self):
        d = _FamilyDesc()
        d.abi_version = 1
        d.has_d = self.has_d
        d.n_seqs = self.msa.shape[0] if self.msa.ndim == 2 else 0
        d.n_sites = self.msa.shape[1] if self.msa.ndim == 2 else 0
        d.msa = self.msa.ctypes.data_as(c_u8p)
        d.n_xmsa = self.n_xmsa
        d.xmsa_site = self.xmsa_site.ctypes.data_as(c_i32p)
        d.xmsa_naive_base = self.xmsa_naive_base.ctypes.data_as(c_u8p)
        d.vpadding, d.vgerm, d.jgerm, d.jpadding = (self.vpadding.c(), self.vgerm.c(), self.jgerm.c(),
                                                    self.jpadding.c())
        if self.dgerm is not None:
            d.dgerm = self.dgerm.c()
        d.vgerm_gene_prob = self.vgerm_gene_prob.ctypes.data_as(c_f64p)
        d.vpadding_transition = self.vpadding_transition.ctypes.data_as(c_f64p)
        d.vgerm_trans_prod = self.vgerm_trans_prod.ctypes.data_as(c_f64p)
        d.jpadding_transition = self.jpadding_transition.ctypes.data_as(c_f64p)
        d.vd = self.vd.c()
        if self.dj is not None:
            d.dj = self.dj.c()
        return d


class Family:
    """Owning wrapper of an lh_family handle."""

    def __init__(self, desc, lib=None):
        self.hip = lib or load_library()
        self.desc = desc
        h = C.c_void_p()
        cdesc = desc.c()
        self.hip.check(self.hip.lib.lh_family_create(C.byref(cdesc), C.byref(h)))
        self.handle = h
        self.forward_size = self.hip.lib.lh_forward_size(h)
        self.scaler_size = self.hip.lib.lh_scaler_size(h)
        self.n_xmsa = desc.n_xmsa
        self.consensus_sets = self.hip.lib.lh_family_consensus_sets(h)

    def close(self):
        if self.handle:
            self.hip.lib.lh_family_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _outs(self, n, R, want):
        outs = _EvalOutputs()
        res = {}
        if "rates" in want:
            res["rates"] = np.zeros((n, R))
            outs.rates = res["rates"].ctypes.data_as(c_f64p)
        if "xmsa_emission" in want:
            res["xmsa_emission"] = np.zeros((n, self.n_xmsa))
            outs.xmsa_emission = res["xmsa_emission"].ctypes.data_as(c_f64p)
        if "forward" in want:
            res["forward"] = np.zeros((n, self.forward_size))
            outs.forward = res["forward"].ctypes.data_as(c_f64p)
        if "scaler_counts" in want:
            res["scaler_counts"] = np.zeros((n, self.scaler_size), dtype=np.int32)
            outs.scaler_counts = res["scaler_counts"].ctypes.data_as(c_i32p)
        return outs, res

    def eval_batch(self, n_tips, max_depth, ops, brlen, er, pi, alpha, num_rates, want=()):
        ops, brlen, er, pi, alpha = _i32(ops), _f64(brlen), _f64(er), _f64(pi), _f64(alpha)
        n = alpha.shape[0]
        assert ops.shape == (n, n_tips - 2, 4) and brlen.shape == (n, 2 * n_tips - 2)
        assert er.shape == (n, 6) and pi.shape == (n, 4)
        ll = np.zeros(n)
        outs, res = self._outs(n, num_rates, want)
        self.hip.check(self.hip.lib.lh_eval_batch(
            self.handle, n, n_tips, max_depth, ops.ctypes.data_as(c_i32p), brlen.ctypes.data_as(c_f64p),
            er.ctypes.data_as(c_f64p), pi.ctypes.data_as(c_f64p), alpha.ctypes.data_as(c_f64p), num_rates,
            ll.ctypes.data_as(c_f64p), C.byref(outs)))
        return ll, res

    def eval_batch_device(self, n, n_tips, max_depth, ops_ptr, brlen_ptr, er_ptr, pi_ptr, alpha_ptr,
                          num_rates, loglik_ptr, stream=0):
        self.hip.check(self.hip.lib.lh_eval_batch_device(
            self.handle, n, n_tips, max_depth, ops_ptr, brlen_ptr, er_ptr, pi_ptr, alpha_ptr, num_rates,
            loglik_ptr, None, stream))

    def info(self):
        """(distinct alignment columns, distinct (naive base, column) pairs in use): lh_family_info."""
        a, b = C.c_int32(), C.c_int32()
        self.hip.check(self.hip.lib.lh_family_info(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def k1_form(self):
        """The pruning-kernel form of the handle's last evaluation (diagnostic, lh_family_prune_form)."""
        return self.hip.lib.lh_family_prune_form(self.handle).decode()

    def status(self):
        """Synchronise the device and raise if a launch since the last call met a malformed (device-resident) schedule."""
        self.hip.check(self.hip.lib.lh_family_status(self.handle))

    def forward_batch(self, em, want=()):
        em = _f64(em)
        n = em.shape[0]
        assert em.shape == (n, self.n_xmsa)
        ll = np.zeros(n)
        outs, res = self._outs(n, 1, [w for w in want if w in ("forward", "scaler_counts")])
        self.hip.check(self.hip.lib.lh_forward_batch(self.handle, n, em.ctypes.data_as(c_f64p),
                                                     ll.ctypes.data_as(c_f64p), C.byref(outs)))
        return ll, res

    def asr_batch(self, n_tips, max_depth, ops, brlen, er, pi, rates, naive, seed, first_sample=0):
        """lh_asr_batch: returns (anc [n][T-2][L] uint8, rate_choice [n][L] uint8)."""
        ops, brlen, er, pi, rates = _i32(ops), _f64(brlen), _f64(er), _f64(pi), _f64(rates)
        naive = np.ascontiguousarray(naive, dtype=np.uint8)
        n, L = naive.shape
        assert ops.shape == (n, n_tips - 2, 4) and brlen.shape == (n, 2 * n_tips - 2)
        assert er.shape == (n, 6) and pi.shape == (n, 4) and rates.shape[0] == n
        anc = np.zeros((n, n_tips - 2, L), dtype=np.uint8)
        choice = np.zeros((n, L), dtype=np.uint8)
        self.hip.check(self.hip.lib.lh_asr_batch(
            self.handle, n, n_tips, max_depth, ops.ctypes.data_as(c_i32p), brlen.ctypes.data_as(c_f64p),
            er.ctypes.data_as(c_f64p), pi.ctypes.data_as(c_f64p), rates.ctypes.data_as(c_f64p), rates.shape[1],
            naive.ctypes.data_as(c_u8p), seed, first_sample, anc.ctypes.data_as(c_u8p), choice.ctypes.data_as(c_u8p)))
        return anc, choice

    def set_extended_range(self, on=True):
        self.hip.check(self.hip.lib.lh_family_set_extended_range(self.handle, int(on)))

    def profile_enable(self, on=True):
        self.hip.check(self.hip.lib.lh_profile_enable(self.handle, int(on)))

    def profile_read(self):
        a, b, c, k = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        self.hip.check(self.hip.lib.lh_profile_read(self.handle, C.byref(a), C.byref(b), C.byref(c),
                                                    C.byref(k)))
        return {"model_ms": a.value, "prune_ms": b.value, "forward_ms": c.value, "launch_groups": k.value}
