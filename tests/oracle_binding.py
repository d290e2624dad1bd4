"""ctypes binding of oracle/liboracle.so (the CPU restatement).  Test infrastructure only."""
import ctypes as C
import os
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-f", "oracle/Makefile", "oracle/liboracle.so"], cwd=_ROOT)
        L = C.CDLL(path)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int] * 6
        L.oracle_add_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_finish.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_finish_count.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_destroy.argtypes = [C.c_void_p]
        for f in ("oracle_node_count", "oracle_kmer_count"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("oracle_set_size", "oracle_set_count", "oracle_set_last_put"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        L.oracle_K.restype = C.c_int
        L.oracle_K.argtypes = [C.c_void_p]
        L.oracle_dump_nodes.restype = C.c_uint64
        L.oracle_dump_nodes.argtypes = [C.c_void_p] * 7
        L.oracle_pregraph.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_int] * 6 + [C.c_char_p]
        _LIB = L
    return _LIB


class Oracle:
    """Incremental oracle context: add reads, then finish() (all files) or finish_count() (.kmerFreq only)."""

    def __init__(self, K, P=8, D=0, a_gb=0, mer127=False, max_read_len=100):
        self.L = lib()
        self.h = self.L.oracle_create(K, P, D, a_gb, int(mer127), max_read_len)
        self.NW = 4 if mer127 else 2
        self.P = P
        self.K = self.L.oracle_K(self.h)

    def add_reads(self, codes, lens=None):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n, stride = codes.shape
        for r in range(n):
            self.L.oracle_add_read(self.h, codes[r].ctypes.data, int(lens[r]) if lens is not None else stride)

    def finish(self, prefix):
        self.L.oracle_finish(self.h, prefix.encode())

    def finish_count(self, prefix):
        self.L.oracle_finish_count(self.h, prefix.encode())

    def nodes(self):
        n = self.L.oracle_node_count(self.h)
        keys = np.zeros((n, self.NW), dtype=np.uint64)
        A = np.zeros(n, dtype=np.uint32)
        B = np.zeros(n, dtype=np.uint32)
        ord_ = np.zeros(n, dtype=np.uint64)
        setid = np.zeros(n, dtype=np.int32)
        slot = np.zeros(n, dtype=np.uint64)
        got = self.L.oracle_dump_nodes(self.h, keys.ctypes.data, A.ctypes.data, B.ctypes.data, ord_.ctypes.data,
                                       setid.ctypes.data, slot.ctypes.data)
        assert got == n
        return dict(keys=keys, A=A, B=B, ord=ord_, set=setid, slot=slot)

    def set_sizes(self):
        return [self.L.oracle_set_size(self.h, p) for p in range(self.P)]

    def set_last_put(self):
        return [self.L.oracle_set_last_put(self.h, p) for p in range(self.P)]

    def kmer_count(self):
        return self.L.oracle_kmer_count(self.h)

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None


def run_oracle(codes, K, P, prefix, D=0, a_gb=0, mer127=False, lens=None, max_read_len=None):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, stride = codes.shape
    lp = None
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        lp = lens.ctypes.data
    lib().oracle_pregraph(codes.ctypes.data, lp, n, stride, K, P, D, a_gb, int(mer127),
                          max_read_len if max_read_len is not None else stride, prefix.encode())
