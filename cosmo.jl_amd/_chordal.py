"""ctypes binding of libcosmo_chordal.so (include/cosmo_chordal.h): the chordal decomposition front-end, host C++.

Mirrors `chordal_decomposition!(ws)` / `reverse_decomposition!(ws, settings)` of the reference
(src/chordal_decomposition/chordal_decomposition.jl:10-38, 126-150)."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

NO_MERGE, PARENT_CHILD_MERGE, CLIQUE_GRAPH_MERGE = 0, 1, 2

_PD = C.POINTER(C.c_double)
_PI64 = C.POINTER(C.c_int64)
_PI32 = C.POINTER(C.c_int32)


class Options(C.Structure):
    _fields_ = [("merge_strategy", C.c_int32), ("t_fill", C.c_int32), ("t_size", C.c_int32), ("orderings", _PI64), ("compact_transformation", C.c_int32)]


SIGNATURES = {
    "cosmo_chordal_default_options": (None, [C.POINTER(Options)]),
    "cosmo_chordal_decompose": (C.c_int32, [C.c_int64, C.c_int64, _PI64, _PI64, _PD, _PD, C.c_int64, _PI32, _PI64, C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "cosmo_chordal_free": (None, [C.c_void_p]),
    "cosmo_chordal_last_error": (C.c_char_p, []),
    "cosmo_chordal_sizes": (C.c_int32, [C.c_void_p, _PI64]),
    "cosmo_chordal_get_problem": (C.c_int32, [C.c_void_p, _PI64, _PI64, _PD, _PD, _PI32, _PI64, _PI64, _PI64]),
    "cosmo_chordal_num_cliques": (C.c_int32, [C.c_void_p, C.c_int64, _PI64, _PI64]),
    "cosmo_chordal_get_cliques": (C.c_int32, [C.c_void_p, C.c_int64, _PI64, _PI64]),
    "cosmo_chordal_merge_log": (C.c_int32, [C.c_void_p, C.c_int64, _PI64, _PI64, _PI64, _PI32, C.c_int64]),
    "cosmo_chordal_reverse": (C.c_int32, [C.c_void_p, _PD, _PD, _PD, _PD, C.c_int32]),
    "cosmo_chordal_test_merge_tree": (C.c_int32, [C.c_int64, _PI64, _PI64, _PI64, _PI64, _PI64, _PI64, C.c_int64, C.c_int32, _PI64, _PI64, _PI64, _PI32, _PI64, C.c_int64]),
    "cosmo_chordal_test_reduced_clique_graph": (C.c_int32, [C.c_int64, _PI64, _PI64, C.c_int64, _PI64, _PI64, _PI64, _PI64, _PI64, _PD, _PI32, C.c_int64]),
}

_lib = None


class ChordalError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcosmo_chordal.so")
    if not os.path.exists(path):
        raise ImportError("libcosmo_chordal.so is missing: run `make -C cosmo.jl_amd/chordal` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _chk(lib, rc):
    if rc != 0:
        raise ChordalError(lib.cosmo_chordal_last_error().decode())


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t)


def _sets_csr(sets):
    ptr = np.zeros(len(sets) + 1, dtype=np.int64)
    idx = []
    for k, s in enumerate(sets):
        idx += sorted(int(v) for v in s)
        ptr[k + 1] = len(idx)
    return ptr, _i64(idx if idx else [0])


class Decomposition:
    """One decomposed problem: `ws.ci` + the augmented `ws.p` of the reference."""

    def __init__(self, A, b, kinds, dims, merge_strategy=CLIQUE_GRAPH_MERGE, t_fill=8, t_size=8, orderings=None, compact=True):
        lib = load_library()
        self.lib = lib
        A = sp.csc_matrix(A, dtype=np.float64)
        A.sort_indices()
        self.m, self.n = A.shape
        b = np.ascontiguousarray(b, dtype=np.float64)
        cp = _i64(A.indptr) + 1
        ri = _i64(A.indices) + 1
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        dims = _i64(dims)
        opt = Options()
        lib.cosmo_chordal_default_options(C.byref(opt))
        opt.merge_strategy, opt.t_fill, opt.t_size = int(merge_strategy), int(t_fill), int(t_size)
        opt.compact_transformation = 1 if compact else 0
        self.compact = bool(compact)
        self._ord = None
        if orderings is not None:
            self._ord = _i64(np.concatenate([np.asarray(o, dtype=np.int64) for o in orderings]))
            opt.orderings = _p(self._ord, _PI64)
        h = C.c_void_p()
        _chk(lib, lib.cosmo_chordal_decompose(self.n, self.m, _p(cp, _PI64), _p(ri, _PI64), _p(np.ascontiguousarray(A.data), _PD), _p(b, _PD),
                                              kinds.size, _p(kinds, _PI32), _p(dims, _PI64), C.byref(opt), C.byref(h)))
        self._h = h
        sz = np.zeros(6, dtype=np.int64)
        _chk(lib, lib.cosmo_chordal_sizes(h, _p(sz, _PI64)))
        self.n_new, self.m_new, nnz, nc, self.num_decomposed, self.num_overlaps = (int(v) for v in sz)
        colptr = np.zeros(self.n_new + 1, dtype=np.int64); rowval = np.zeros(max(nnz, 1), dtype=np.int64); nzval = np.zeros(max(nnz, 1))
        self.b = np.zeros(self.m_new); self.kinds = np.zeros(nc, dtype=np.int32); self.dims = np.zeros(nc, dtype=np.int64)
        self.cone_map = np.zeros(nc, dtype=np.int64); self.clique_of = np.zeros(nc, dtype=np.int64)
        _chk(lib, lib.cosmo_chordal_get_problem(h, _p(colptr, _PI64), _p(rowval, _PI64), _p(nzval, _PD), _p(self.b, _PD), _p(self.kinds, _PI32),
                                                _p(self.dims, _PI64), _p(self.cone_map, _PI64), _p(self.clique_of, _PI64)))
        self.A = sp.csc_matrix((nzval[:nnz], rowval[:nnz] - 1, colptr - 1), shape=(self.m_new, self.n_new))

    def cliques(self, cone):
        """Cliques (1-based vertex lists, post order) of the original cone with 1-based index `cone`."""
        num = C.c_int64(0); tot = C.c_int64(0)
        _chk(self.lib, self.lib.cosmo_chordal_num_cliques(self._h, int(cone), C.byref(num), C.byref(tot)))
        if num.value == 0:
            return []
        ptr = np.zeros(num.value + 1, dtype=np.int64); v = np.zeros(max(tot.value, 1), dtype=np.int64)
        _chk(self.lib, self.lib.cosmo_chordal_get_cliques(self._h, int(cone), _p(ptr, _PI64), _p(v, _PI64)))
        return [v[ptr[i]:ptr[i + 1]].tolist() for i in range(num.value)]

    def merge_log(self, cone, capacity=100000):
        nd = C.c_int64(0); nm = C.c_int64(0)
        pairs = np.zeros(2 * capacity, dtype=np.int64); dec = np.zeros(capacity, dtype=np.int32)
        _chk(self.lib, self.lib.cosmo_chordal_merge_log(self._h, int(cone), C.byref(nd), C.byref(nm), _p(pairs, _PI64), _p(dec, _PI32), capacity))
        return pairs[:2 * nd.value].reshape(-1, 2), dec[:nd.value].astype(bool), int(nm.value)

    def reverse(self, s_dec, mu_dec, complete_dual=False):
        s = np.zeros(self.m); mu = np.zeros(self.m)
        sd = np.ascontiguousarray(s_dec, dtype=np.float64); md = np.ascontiguousarray(mu_dec, dtype=np.float64)
        if sd.size != self.m_new or md.size != self.m_new:
            raise ValueError("reverse: vectors of the decomposed problem expected")
        _chk(self.lib, self.lib.cosmo_chordal_reverse(self._h, _p(sd, _PD), _p(md, _PD), _p(s, _PD), _p(mu, _PD), 1 if complete_dual else 0))
        return s, mu

    def close(self):
        if self._h:
            self.lib.cosmo_chordal_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def test_merge_tree(snd, sep, par, snd_post, nvertices, strategy, capacity=1000):
    """Golden-test hook: merge_cliques! on an explicit (1-based) clique tree.  Returns (pairs, decisions, num_merges, parents)."""
    lib = load_library()
    sp_, si = _sets_csr(snd); pp, pi = _sets_csr(sep)
    par = _i64(par); post = _i64(snd_post)
    nd = C.c_int64(0); nm = C.c_int64(0)
    pairs = np.zeros(2 * capacity, dtype=np.int64); dec = np.zeros(capacity, dtype=np.int32); par_out = np.zeros(len(snd), dtype=np.int64)
    _chk(lib, lib.cosmo_chordal_test_merge_tree(len(snd), _p(sp_, _PI64), _p(si, _PI64), _p(pp, _PI64), _p(pi, _PI64), _p(par, _PI64), _p(post, _PI64),
                                                int(nvertices), int(strategy), C.byref(nd), C.byref(nm), _p(pairs, _PI64), _p(dec, _PI32), _p(par_out, _PI64), capacity))
    return pairs[:2 * nd.value].reshape(-1, 2), dec[:nd.value].astype(bool), int(nm.value), par_out


def test_reduced_clique_graph(snd, sep, capacity=10000):
    """Golden-test hook: compute_reduced_clique_graph! + ComplexityWeight + ispermissible.  Returns (rows, cols, weights, permissible)."""
    lib = load_library()
    sp_, si = _sets_csr(snd); pp, pi = _sets_csr(sep)
    ne = C.c_int64(0)
    rows = np.zeros(capacity, dtype=np.int64); cols = np.zeros(capacity, dtype=np.int64); w = np.zeros(capacity); perm = np.zeros(capacity, dtype=np.int32)
    _chk(lib, lib.cosmo_chordal_test_reduced_clique_graph(len(snd), _p(sp_, _PI64), _p(si, _PI64), len(sep), _p(pp, _PI64), _p(pi, _PI64), C.byref(ne),
                                                          _p(rows, _PI64), _p(cols, _PI64), _p(w, _PD), _p(perm, _PI32), capacity))
    k = ne.value
    return rows[:k], cols[:k], w[:k], perm[:k].astype(bool)
