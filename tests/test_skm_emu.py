"""CPU check of the partition engine's shared host/device logic (csrc/skm.hpp + csrc/extract.hpp) through the
serial harness tests/emu_skm.cpp: super-k-mer cutting, record format, flank rules, ordinals, 63-bit key re-cut
and the one-k-mer-one-partition invariant, against the oracle's per-k-mer records."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, case_codes, oracle_records
from soapdenovo2_amd import api


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emu") / "libemu_skm.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                           os.path.join(ROOT, "tests", "emu_skm.cpp"), "-o", so])
    L = C.CDLL(so)
    L.emu_skm_count.restype = C.c_int64
    L.emu_skm_count.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    return L


@pytest.mark.parametrize("name,m127,log2_parts", [("t6k_k31", False, 8), ("t8k_k63", False, 10), ("t8k_k63", True, 6),
                                                   ("t6k_k127", True, 8), ("t5k_k24", False, 4)])
def test_partition_logic_matches_oracle(golden, emu, tmp_path, name, m127, log2_parts):
    c = golden["cases"][name]
    codes = case_codes(c)
    want, _, K = oracle_records(codes, c["K"], 1, mer127=m127, prefix=str(tmp_path / "o"))
    nw = 4 if m127 else 2
    packed = api.pack_reads_uniform(codes)
    out = np.zeros((want.shape[0] + 16, nw + 2), dtype=np.uint64)
    nrec, maxd = C.c_int64(0), C.c_int64(0)
    n = emu.emu_skm_count(packed.ctypes.data, codes.shape[0], codes.shape[1], K, int(m127), log2_parts, out.ctypes.data,
                          out.shape[0], C.byref(nrec), C.byref(maxd))
    assert n == want.shape[0], n
    got = out[:n]
    key = lambda r: r[np.lexsort([r[:, i] for i in range(nw - 1, -1, -1)])]
    w = key(want).copy()
    w[:, nw + 1] &= np.uint64((1 << 56) - 1)                 # oracle records carry the set id in the top byte
    w[:, nw] &= np.uint64(~(1 << (32 + 24)) & 0xFFFFFFFFFFFFFFFF)   # ... and the linear flag set by finish_count
    assert (key(got) == w).all()
    reads, kpr = codes.shape[0], codes.shape[1] - K + 1
    assert reads <= nrec.value <= reads * kpr
    print(name, "records/read", nrec.value / reads, "max distinct in a partition", maxd.value)


@pytest.mark.parametrize("name,m127", [("t6k_k31", False), ("t8k_k63", False), ("t8k_k63", True), ("t6k_k127", True), ("t5k_k24", False)])
def test_tiled_cutter_matches_serial_walk(golden, emu, name, m127):
    """K1's tile formulation (csrc/skm_tile.hpp: m-mer chunks, segment minima, start bits, next start, record build) run
    serially on the CPU gives exactly the runs and records of skm_split_read + skm_make_record, for every segment length."""
    c = golden["cases"][name]
    codes = case_codes(c)[:1500]
    K = c["K"] | 1
    packed = api.pack_reads_uniform(codes)
    emu.emu_tile_check.restype = C.c_int64
    emu.emu_tile_check.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    m = max(7, min(16, K - 6))
    w = K - m + 1
    for S in (7, 9, 11, 13, 15):
        if S > w and S != 7:
            continue
        for R in (1, 5, 32):
            n = emu.emu_tile_check(packed.ctypes.data, codes.shape[0], codes.shape[1], K, int(m127), 9, S, R)
            assert n >= codes.shape[0], (S, R, n)
    assert emu.emu_tile_pick_segment(88, 48) == 11


@pytest.mark.parametrize("name,m127", [("t6k_k31", False), ("t8k_k63", False), ("t6k_k127", True), ("t5k_k24", False)])
def test_tiled_cutter_ragged_matches_serial_walk(golden, emu, name, m127):
    """The RAGGED form of K1's tiles (rows sized for the batch's longest read, a read's own length masks its segments; the
    reference chops reads of any length >= K + 1 the same way, prlHashReads.c:163-259,642-648): same runs, same records as the
    serial walk, for reads trimmed to every length from K + 1 to the full one -- including tiles of one length and a bound
    larger than any read."""
    c = golden["cases"][name]
    codes = case_codes(c)[:1200]
    K = c["K"] | 1
    full = codes.shape[1]
    rng = np.random.default_rng(11)
    lens = rng.integers(K + 1, full + 1, size=codes.shape[0]).astype(np.int32)
    lens[:7] = [K + 1, full, K + 2, full, K + 1, K + 1, full]
    lens[100:140] = full - 3                                   # a stretch of one length inside the mix
    reads = [codes[i, :lens[i]] for i in range(codes.shape[0])]
    words, word_off, kmer_base = api.pack_reads_ragged(reads, K)
    emu.emu_tile_check_ragged.restype = C.c_int64
    emu.emu_tile_check_ragged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 6
    m = max(7, min(16, K - 6))
    w = K - m + 1
    for S in (7, 9, 11, 13, 15):
        if S > w and S != 7:
            continue
        for R, bound in ((1, full), (5, full), (32, full), (24, full + 9)):
            n = emu.emu_tile_check_ragged(words.ctypes.data, word_off.ctypes.data, lens.ctypes.data, len(reads), bound, K, int(m127), 9, S, R)
            assert n >= len(reads), (S, R, bound, n)
    # a read longer than the bound is refused, not cut wrongly
    assert emu.emu_tile_check_ragged(words.ctypes.data, word_off.ctypes.data, lens.ctypes.data, len(reads), full - 1, K, int(m127), 9, 7, 8) == -107


def test_sliced_crc_equals_bytewise(emu):
    assert emu.emu_crc_check(20000) == 0
