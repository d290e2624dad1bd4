"""ctypes binding of libsoapdenovo2_amd.so -- the C ABI declared in include/soapdenovo2_amd.h.

Python is plumbing here (tests, bench.py, the multi-GPU launcher): the product is the shared library and
the SOAPdenovo-63mer / SOAPdenovo-127mer executables next to it.  There is no Python or CPU fallback for the
device operators: if the library is missing, or no HIP device is usable, the calls raise.

Mirrors the reference's entry point `int call_pregraph(int argc, char **argv)`
(standardPregraph/pregraph.c:62) as :func:`call_pregraph`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libsoapdenovo2_amd.so")
BIN_DIR = os.path.join(_HERE, "bin")

PG_ORD_BITS = 56
PG_ORD_MASK = (1 << PG_ORD_BITS) - 1

_lib = None


class PgError(RuntimeError):
    pass


def build(verbose: bool = False) -> None:
    """Compile the HIP extension and executables for gfx950 (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", ROOT, "-j8"], capture_output=not verbose, text=True)
    if out.returncode != 0:
        raise PgError("build failed:\n" + (out.stdout or "") + (out.stderr or ""))


# int fetch(void *user, uint64_t first_record, uint64_t n_records, uint64_t *dst)  (pg_graph_begin_streamed)
FETCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p)


def lib() -> C.CDLL:
    """Load the shared library (import torch first when torch is used in the same process, so that both
    share torch's HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SOAPDENOVO2_AMD_LIB", LIB_PATH)        # A/B builds of the library (development aid)
    if not os.path.exists(path):
        raise PgError(f"{path} is missing: run `make` (or __graft_entry__.build()) first; "
                      "there is no fallback implementation")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    u64p = C.c_void_p
    L.pg_last_error.restype = C.c_char_p
    L.pg_version.restype = C.c_char_p
    L.call_pregraph.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.call_pregraph_127mer.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.pg_packed_words.restype = C.c_size_t
    L.pg_packed_words.argtypes = [C.c_uint32]
    L.pg_pack_read.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.pg_host_build_graph.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pg_host_write_kmerfreq.argtypes = [u64p, C.c_char_p]
    L.pg_host_graph_begin.restype = C.c_void_p
    L.pg_host_graph_begin.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.pg_graph_begin.restype = C.c_void_p
    L.pg_graph_begin.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.pg_graph_begin_streamed.restype = C.c_void_p
    L.pg_graph_begin_streamed.argtypes = [FETCH_FN, C.c_void_p, C.c_uint64, u64p, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_char_p, C.c_int]
    L.pg_graph_begin_device.restype = C.c_void_p
    L.pg_graph_begin_device.argtypes = [u64p, C.c_int, C.c_uint64, u64p, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.pg_host_graph_add_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    L.pg_host_graph_finish.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    L.pg_host_graph_resolve_repeats.argtypes = [C.c_void_p, C.c_int]
    L.pg_graph_use_device.argtypes = [C.c_void_p, C.c_int]
    L.pg_expect_kmers.argtypes = [C.c_void_p, C.c_uint64]
    L.pg_set_read_len_bound.argtypes = [C.c_void_p, C.c_uint32]
    L.pg_expect.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.pg_host_plan_memory.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, u64p]
    L.pg_sort_records.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    L.pg_host_graph_add_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.pg_host_read_all.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.pg_host_replay_layout.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, u64p, u64p]
    L.pg_create.restype = C.c_void_p
    L.pg_create.argtypes = [C.c_int] * 5
    L.pg_create_engine.restype = C.c_void_p
    L.pg_create_engine.argtypes = [C.c_int] * 6
    L.pg_create_planned.restype = C.c_void_p
    L.pg_create_planned.argtypes = [C.c_int] * 6 + [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.pg_create_sized.restype = C.c_void_p
    L.pg_create_sized.argtypes = [C.c_int] * 6 + [C.c_uint64]
    L.pg_destroy.argtypes = [C.c_void_p]
    L.pg_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.pg_set_autogrow.argtypes = [C.c_void_p, C.c_int]
    L.pg_count_reads.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
    L.pg_route_count.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, u64p, C.c_void_p]
    L.pg_route_scatter.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                   u64p, u64p, u64p, C.c_void_p]
    L.pg_count_records.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_void_p]
    if not hasattr(L, "pg_skm_route"):          # an older A/B build of the library
        _lib = L
        return L
    L.pg_skm_route.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, u64p, u64p, C.c_uint64, u64p, C.c_void_p]
    L.pg_skm_ingest.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, C.c_void_p]
    L.pg_distinct.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    L.pg_table_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.pg_stats.argtypes = [C.c_void_p, u64p]
    L.pg_finalize.argtypes = [C.c_void_p, C.c_int, u64p, u64p, C.c_void_p]
    L.pg_export.argtypes = [C.c_void_p, u64p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    L.pg_export_take.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pg_export_take_ws.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pg_sort_records_ws.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    L.pg_device_free.argtypes = [C.c_void_p]
    L.pg_device_arena_pin.argtypes = [C.c_int]
    L.pg_device_arena_unpin.argtypes = [C.c_int]
    L.pg_device_arena_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    L.pg_host_emu_arena_blocks.restype = C.c_longlong
    L.pg_host_emu_arena_blocks.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.pg_export_peek.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.pg_records_checksum.argtypes = [C.c_void_p, C.c_uint64, C.c_int, u64p, C.c_void_p]
    L.pg_set_counts.argtypes = [C.c_void_p, u64p, C.c_void_p]
    L.pg_last_put.argtypes = [C.c_void_p, u64p, C.c_void_p]
    L.pg_host_last_put_matters.argtypes = [u64p, C.c_int, C.c_int, C.c_int]
    # multi-GPU exchange (include/soapdenovo2_amd.h, section 4)
    L.pg_comm_unique_id.argtypes = [C.c_void_p]
    L.pg_comm_create.restype = C.c_void_p
    L.pg_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.pg_comm_create_local.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.pg_comm_destroy.argtypes = [C.c_void_p]
    L.pg_comm_rank.argtypes = [C.c_void_p]
    L.pg_comm_size.argtypes = [C.c_void_p]
    L.pg_comm_transport.argtypes = [C.c_void_p]
    L.pg_comm_stats.argtypes = [C.c_void_p, u64p]
    L.pg_comm_pipeline_stats.argtypes = [C.c_void_p, u64p]
    L.pg_comm_create_host.restype = C.c_void_p
    L.pg_comm_create_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.pg_comm_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.pg_exchange_counts.argtypes = [C.c_void_p, u64p, u64p, C.c_void_p]
    L.pg_exchange_records.argtypes = [C.c_void_p, u64p, u64p, C.c_uint64, C.c_int, u64p, u64p, u64p, u64p, C.c_void_p]
    L.pg_exchange_allreduce_u64.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_void_p]
    L.pg_exchange_gather_records.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_int, u64p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    L.pg_count_reads_sharded.argtypes = [C.c_void_p, C.c_void_p, u64p, u64p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
    L.pg_host_skm_cut.restype = C.c_int64
    L.pg_host_skm_cut.argtypes = [u64p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, u64p, u64p, C.c_uint64]
    L.pg_host_skm_expand.restype = C.c_int64
    L.pg_host_skm_expand.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int, u64p, C.c_uint64]
    L.pg_host_emu_layout_growable.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, u64p, u64p, u64p, u64p, C.c_uint64]
    L.pg_host_regroup_plan.argtypes = [u64p, C.c_uint64, C.c_int, C.c_int, u64p, u64p]
    L.pg_host_emu_clip_tips.argtypes = [u64p, C.c_uint64, u64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p]
    L.pg_host_emu_layout_static.argtypes = [u64p, u64p, C.c_int, C.c_uint64, C.c_int, C.c_int, u64p]
    L.pg_host_emu_home_slots.argtypes = [u64p, C.c_uint64, C.c_int, C.c_uint64, u64p]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "pg_last_error", "pg_version", "call_pregraph", "call_pregraph_127mer", "pg_packed_words", "pg_pack_read",
    "pg_host_build_graph", "pg_host_graph_begin", "pg_host_graph_add_reads", "pg_host_graph_finish", "pg_process_exits_after_this", "pg_host_graph_resolve_repeats", "pg_host_graph_add_packed", "pg_graph_use_device", "pg_sort_records", "pg_expect_kmers", "pg_create_sized", "pg_graph_begin", "pg_graph_begin_streamed", "pg_host_read_all", "pg_host_replay_layout", "pg_host_write_kmerfreq", "pg_create", "pg_create_engine", "pg_destroy", "pg_reset", "pg_set_autogrow", "pg_count_reads", "pg_route_count",
    "pg_route_scatter", "pg_count_records", "pg_skm_route", "pg_skm_ingest", "pg_distinct", "pg_stats", "pg_table_info", "pg_finalize", "pg_export",
    "pg_export_take", "pg_export_take_ws", "pg_export_peek", "pg_records_checksum", "pg_sort_records_ws", "pg_device_free", "pg_device_arena_pin", "pg_device_arena_unpin", "pg_device_arena_stats", "pg_host_emu_arena_blocks", "pg_set_counts", "pg_last_put", "pg_host_last_put_matters", "pg_comm_unique_id", "pg_comm_create", "pg_comm_create_local", "pg_comm_destroy", "pg_comm_rank", "pg_comm_size",
    "pg_comm_transport", "pg_comm_stats", "pg_exchange_counts", "pg_exchange_records", "pg_exchange_allreduce_u64",
    "pg_exchange_gather_records", "pg_count_reads_sharded", "pg_host_skm_cut", "pg_host_skm_expand",
    "pg_host_emu_layout_static", "pg_graph_begin_device", "pg_host_emu_clip_tips", "pg_exchange_regroup_by_set", "pg_comm_regroup_stats", "pg_graph_begin_sharded", "pg_host_regroup_plan", "pg_host_bam_pair_state", "pg_device_scratch_offer", "pg_device_scratch_withdraw", "pg_host_emu_layout_growable", "pg_exchange_regroup_by_set_ws", "pg_host_edge_file_in_background", "pg_graph_add_packed_device", "pg_host_emu_home_slots", "pg_comm_pipeline_stats", "pg_comm_create_host", "pg_comm_flush",
    "pg_set_read_len_bound", "pg_graph_add_packed_device_ragged", "pg_expect", "pg_host_plan_memory", "pg_create_planned", "pg_graph_add_packed_device_segments",
]


PLAN_FIELDS = ["peak", "peak_stage", "tables", "record_pool", "export_allocated", "export_after_count", "reads_kept", "batch_and_exchange", "kmer_sets",
               "layout_arrays", "sort_work_space_outside_pool", "log2_partition_ids", "log2_partitions_stored", "direct_chunks", "pool_records", "fits",
               "stage1_pass1_count", "stage2_hand_over", "stage3_layout", "stage4_graph_pass2", "est_kmers", "export_records", "set_slots", "log2_slots"]


def plan_memory(reads_total: int, read_len: int, distinct_total: int, K: int, n_sets: int = 8, a_gb: int = 0, n_ranks: int = 1,
                device_bytes: int = 288 * 10**9, fastq_bytes: int = 0) -> dict:
    """pg_host_plan_memory: the device memory one rank of the command takes, stage by stage (no GPU)."""
    out = np.zeros(24, dtype=np.uint64)
    _check(lib().pg_host_plan_memory(reads_total, read_len, fastq_bytes, distinct_total, K, 1 if K > 63 else 0, n_sets, a_gb, n_ranks, device_bytes,
                                     out.ctypes.data_as(C.POINTER(C.c_uint64))), "pg_host_plan_memory")
    d = {k: int(v) for k, v in zip(PLAN_FIELDS, out)}
    d["counts_twice"] = bool(d["log2_slots"] >> 32)
    d["log2_slots"] &= 0xFFFFFFFF
    return d


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise PgError(f"{what} failed ({rc}): {lib().pg_last_error().decode()}")


def binary(mer127: bool = False) -> str:
    return os.path.join(BIN_DIR, "SOAPdenovo-127mer" if mer127 else "SOAPdenovo-63mer")


def call_pregraph(args: Sequence[str], mer127: bool = False, in_process: bool = False) -> int:
    """`pregraph <args>`; args as for the reference, e.g. ["-s", cfg, "-K", "31", "-o", prefix, "-p", "8"].

    By default runs the executable in a child process (fatal input errors `exit()` like the reference's);
    in_process=True calls the C entry point directly."""
    if in_process:
        argv = [b"pregraph"] + [str(a).encode() for a in args]
        arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
        fn = lib().call_pregraph_127mer if mer127 else lib().call_pregraph
        return fn(len(argv), arr)
    return subprocess.run([binary(mer127), "pregraph"] + [str(a) for a in args]).returncode


# ---------------------------------------------------------------------------------------------------------
# host helpers
# ---------------------------------------------------------------------------------------------------------
def packed_words(length: int) -> int:
    return (length + 31) // 32


def pack_reads_uniform(codes: np.ndarray) -> np.ndarray:
    """(n, L) uint8 base codes -> (n * words_per_read + 8,) uint64 in the device read format (vectorised
    equivalent of pg_pack_read: first base in the most significant bits, reads word-aligned; 8 words of
    readable padding at the end)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, L = codes.shape
    wpr = packed_words(L)
    padded = np.zeros((n, wpr * 32), dtype=np.uint64)
    padded[:, :L] = codes & 3
    shifts = (62 - 2 * np.arange(32, dtype=np.uint64)).astype(np.uint64)
    words = (padded.reshape(n, wpr, 32) << shifts[None, None, :]).sum(axis=2, dtype=np.uint64)
    out = np.zeros(n * wpr + 8, dtype=np.uint64)
    out[: n * wpr] = words.reshape(-1)
    return out


def pack_reads_ragged(reads: Sequence[np.ndarray], K: int):
    """List of 1-D uint8 code arrays (each len >= K + 1) -> (words, word_off, kmer_base) numpy uint64 arrays."""
    L = lib()
    n = len(reads)
    word_off = np.zeros(n, dtype=np.uint64)
    kmer_base = np.zeros(n + 1, dtype=np.uint64)
    total = 0
    for i, r in enumerate(reads):
        word_off[i] = total
        total += packed_words(len(r))
        kmer_base[i + 1] = kmer_base[i] + np.uint64(len(r) - K + 1)
    words = np.zeros(total + 8, dtype=np.uint64)
    for i, r in enumerate(reads):
        r = np.ascontiguousarray(r, dtype=np.uint8)
        L.pg_pack_read(r.ctypes.data, len(r), words[int(word_off[i]):].ctypes.data)
    return words, word_off, kmer_base


def host_build_graph(records: np.ndarray, set_last_put, K: int, n_sets: int, prefix: str, mer127: bool = False,
                     cut_single: bool = True, a_gb: int = 0, max_read_len: int = 100, n_threads: int = 0):
    """records: (n, nw + 2) uint64.  Writes <prefix>.vertex/.edge.gz/.preGraphBasic; returns (n_vertex, n_edge)."""
    records = np.ascontiguousarray(records, dtype=np.uint64)
    slp = np.ascontiguousarray(set_last_put, dtype=np.uint64)
    nv, ne = C.c_int(0), C.c_int(0)
    rc = lib().pg_host_build_graph(records.ctypes.data, records.shape[0], slp.ctypes.data, K, int(mer127), n_sets,
                                   int(cut_single), a_gb, max_read_len, n_threads, prefix.encode(), C.byref(nv), C.byref(ne))
    _check(rc, "pg_host_build_graph")
    return nv.value, ne.value


def host_pregraph_files(records: np.ndarray, set_last_put, codes: np.ndarray, lens, K: int, n_sets: int, prefix: str,
                        mer127: bool = False, cut_single: bool = True, a_gb: int = 0, max_read_len: int = 100, n_threads: int = 0,
                        batches: int = 1, resolve_repeats: bool = False, packed: bool = False,
                        device: int = -1, device_edges: bool = False, streamed: bool = False):
    """All host stages incl. pass 2: writes .edge.gz .preArc .vertex .preGraphBasic (and, with resolve_repeats, the
    reference's -R files .path and .markOnEdge); returns (n_vertex, n_edge, n_prearc)."""
    records = np.ascontiguousarray(records, dtype=np.uint64)
    slp = np.ascontiguousarray(set_last_put, dtype=np.uint64)
    if streamed:                         # records handed over through the fetch callback, in replay order
        rw = records.shape[1]
        order = np.argsort(records[:, rw - 1], kind="stable")
        srt = np.ascontiguousarray(records[order])
        per_set = np.bincount((srt[:, rw - 1] >> np.uint64(56)).astype(np.int64), minlength=n_sets).astype(np.uint64)

        def _fetch(user, first, n, dst):
            C.memmove(dst, srt[first:].ctypes.data, n * rw * 8)
            return 0
        cb = FETCH_FN(_fetch)
        h = lib().pg_graph_begin_streamed(cb, None, srt.shape[0], per_set.ctypes.data, slp.ctypes.data, K, int(mer127), n_sets, int(cut_single),
                                          a_gb, max_read_len, n_threads, prefix.encode(), device if device_edges else -1)
    elif device_edges:                   # edges (and then pass 2) on HIP device `device`
        h = lib().pg_graph_begin(records.ctypes.data, records.shape[0], slp.ctypes.data, K, int(mer127), n_sets, int(cut_single),
                                 a_gb, max_read_len, n_threads, prefix.encode(), device)
    else:
        h = lib().pg_host_graph_begin(records.ctypes.data, records.shape[0], slp.ctypes.data, K, int(mer127), n_sets, int(cut_single),
                                      a_gb, max_read_len, n_threads, prefix.encode())
    if not h:
        raise PgError("pg_graph_begin failed: " + lib().pg_last_error().decode())
    if resolve_repeats:
        _check(lib().pg_host_graph_resolve_repeats(h, 1), "pg_host_graph_resolve_repeats")
    if device >= 0:                      # pass 2 on the HIP device instead of the host threads
        _check(lib().pg_graph_use_device(h, device), "pg_graph_use_device")
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, stride = codes.shape
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.int32)
    bounds = np.linspace(0, n, batches + 1).astype(int)
    for b in range(batches):
        lo, hi = int(bounds[b]), int(bounds[b + 1])
        if hi > lo and packed:           # the reads as pass 1 packs them (pg_pack_read), back to back
            ls = lens[lo:hi] if lens is not None else np.full(hi - lo, stride, dtype=np.int32)
            words, _, _ = pack_reads_ragged([codes[i, :ls[i - lo]] for i in range(lo, hi)], K)
            ls = np.ascontiguousarray(ls, dtype=np.int32)
            _check(lib().pg_host_graph_add_packed(h, words.ctypes.data, ls.ctypes.data, hi - lo, n_threads), "pg_host_graph_add_packed")
        elif hi > lo:
            _check(lib().pg_host_graph_add_reads(h, codes[lo:].ctypes.data, lens[lo:].ctypes.data if lens is not None else None,
                                                 hi - lo, stride, n_threads), "pg_host_graph_add_reads")
    nv, ne, na = C.c_int(0), C.c_int(0), C.c_longlong(0)
    _check(lib().pg_host_graph_finish(h, C.byref(nv), C.byref(ne), C.byref(na)), "pg_host_graph_finish")
    return nv.value, ne.value, na.value


def host_read_all(config: str, K: int):
    """All reads the reference would hand to a pass over the inputs, in its order: (codes [n, stride] uint8, lens int32, n_records,
    max_rd_len).  (The BAM reader's pairing state is back at -3 after every file, readseq1by1.c:584-587: each pass sees the same reads.)"""
    nrec, nacc, mrl = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    _check(lib().pg_host_read_all(config.encode(), K, None, None, 0, 0, C.byref(nrec), C.byref(nacc), C.byref(mrl)), "pg_host_read_all")
    n, stride = nacc.value, max(mrl.value, 1)
    codes = np.zeros((max(n, 1), stride), dtype=np.uint8)
    lens = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib().pg_host_read_all(config.encode(), K, codes.ctypes.data, lens.ctypes.data, n, stride, C.byref(nrec), C.byref(nacc),
                                  C.byref(mrl)), "pg_host_read_all")
    return codes[:n], lens[:n], nrec.value, mrl.value


def host_bam_state() -> int:
    return int(lib().pg_host_bam_pair_state(0, 0))


def host_replay_layout(records: np.ndarray, set_last_put, n_sets: int, mer127: bool = False, a_gb: int = 0):
    """Slot of every record in its reference k-mer set, and the per-set table sizes."""
    records = np.ascontiguousarray(records, dtype=np.uint64)
    slp = np.ascontiguousarray(set_last_put, dtype=np.uint64)
    slots = np.zeros(records.shape[0], dtype=np.uint64)
    sizes = np.zeros(n_sets, dtype=np.uint64)
    _check(lib().pg_host_replay_layout(records.ctypes.data, records.shape[0], slp.ctypes.data, int(mer127), n_sets, a_gb,
                                       slots.ctypes.data, sizes.ctypes.data), "pg_host_replay_layout")
    return slots, sizes


def host_write_kmerfreq(hist: np.ndarray, prefix: str) -> None:
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    assert hist.shape == (256,)
    _check(lib().pg_host_write_kmerfreq(hist.ctypes.data, prefix.encode()), "pg_host_write_kmerfreq")


# ---------------------------------------------------------------------------------------------------------
# device operators (torch tensors carry the device memory; the kernels are the library's)
# ---------------------------------------------------------------------------------------------------------
PG_COMM_RCCL, PG_COMM_P2P, PG_COMM_HOST = 0, 1, 2
HOST_A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))


class Comm:
    """One rank of a pass-1 communicator (pg_comm_*): `Comm.rccl(n, rank, device, id)` for one rank per process,
    `Comm.local(devices)` for several ranks (host threads) in this process."""

    def __init__(self, handle, owner=True):
        self.h = handle
        self.owner = owner
        self._keep = None

    @staticmethod
    def host(n_ranks: int, rank: int, device: int, alltoallv) -> "Comm":
        """One rank per process, the variable all-to-all brought by the caller (pg_comm_create_host): alltoallv(send: bytes-like
        view, send_off, send_cnt, recv: writable view, recv_off, recv_cnt) over HOST memory, lists of n_ranks byte offsets / counts.
        For ranks that cannot talk RCCL (e.g. several processes on one GPU under gloo): exchange staged through the host."""
        def thunk(_user, send, soff, scnt, recv, roff, rcnt):
            try:
                so, sc = [int(soff[i]) for i in range(n_ranks)], [int(scnt[i]) for i in range(n_ranks)]
                ro, rc = [int(roff[i]) for i in range(n_ranks)], [int(rcnt[i]) for i in range(n_ranks)]
                s_len = max([o + c for o, c in zip(so, sc)] + [0])
                r_len = max([o + c for o, c in zip(ro, rc)] + [0])
                sv = (C.c_char * max(s_len, 1)).from_address(send)
                rv = (C.c_char * max(r_len, 1)).from_address(recv)
                alltoallv(memoryview(sv).cast("B"), so, sc, memoryview(rv).cast("B"), ro, rc)
                return 0
            except Exception as e:                                   # an exception must not unwind through the C frames
                import sys, traceback
                traceback.print_exc(file=sys.stderr)
                return 1
        fn = HOST_A2A_FN(thunk)
        h = lib().pg_comm_create_host(n_ranks, rank, device, C.cast(fn, C.c_void_p), None)
        if not h:
            raise PgError("pg_comm_create_host failed: " + lib().pg_last_error().decode())
        c = Comm(h)
        c._keep = fn                                                 # the callback lives as long as the communicator
        return c

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(lib().pg_comm_unique_id(C.addressof(buf)), "pg_comm_unique_id")
        return bytes(buf)

    @staticmethod
    def rccl(n_ranks: int, rank: int, device: int, uid: bytes) -> "Comm":
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        h = lib().pg_comm_create(n_ranks, rank, device, C.addressof(buf))
        if not h:
            raise PgError("pg_comm_create failed: " + lib().pg_last_error().decode())
        return Comm(h)

    @staticmethod
    def local(devices: Sequence[int], transport: int = -1):
        n = len(devices)
        dv = (C.c_int * n)(*devices)
        out = (C.c_void_p * n)()
        _check(lib().pg_comm_create_local(n, C.addressof(dv), transport, C.addressof(out)), "pg_comm_create_local")
        return [Comm(out[i]) for i in range(n)]

    @property
    def rank(self) -> int:
        return lib().pg_comm_rank(self.h)

    @property
    def size(self) -> int:
        return lib().pg_comm_size(self.h)

    @property
    def transport(self) -> str:
        return {PG_COMM_RCCL: "rccl", PG_COMM_P2P: "p2p", PG_COMM_HOST: "host"}[lib().pg_comm_transport(self.h)]

    def pipeline_stats(self) -> dict:
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().pg_comm_pipeline_stats(self.h, out.ctypes.data), "pg_comm_pipeline_stats")
        return {"exchange_ms": int(out[0]) / 1000.0, "bytes_sent": int(out[1]), "host_waits": int(out[2]), "repeated_cuts": int(out[3]), "rounds": int(out[4]),
                "owner_region_records": int(out[5])}

    def flush(self, counter, stream=None) -> None:
        """What the last round of counter.count_sharded left in flight is appended to its partition streams (pg_finalize does it too)."""
        _check(lib().pg_comm_flush(counter.h, self.h, stream if stream is not None else counter._stream()), "pg_comm_flush")

    def stats(self) -> dict:
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().pg_comm_stats(self.h, out.ctypes.data), "pg_comm_stats")
        return {"rounds": int(out[0]), "sent_records": int(out[1]), "recv_records": int(out[2]), "cap": int(out[3])}

    def allreduce_u64(self, d_tensor, stream=None) -> None:
        _check(lib().pg_exchange_allreduce_u64(self.h, d_tensor.data_ptr(), d_tensor.numel(), stream), "pg_exchange_allreduce_u64")

    def close(self) -> None:
        if self.h and self.owner:
            lib().pg_comm_destroy(self.h)
        self.h = None


def arena_stats(device: int = 0) -> dict:
    """The library's device arena (csrc/arena.hpp): what it reserved, mapped and handed out on `device`."""
    out = (C.c_uint64 * 8)()
    lib().pg_device_arena_stats(device, out)
    keys = ("active", "reserved", "mapped", "in_use", "peak_in_use", "blocks_cut", "pieces_created", "create_us")
    return dict(zip(keys, [int(x) for x in out]))


class arena_pinned:
    """`with api.arena_pinned(0): ...` keeps the arena's physical memory across contexts created and destroyed inside the block."""
    def __init__(self, device: int = 0):
        self.device = device
    def __enter__(self):
        lib().pg_device_arena_pin(self.device)
        return self
    def __exit__(self, *a):
        lib().pg_device_arena_unpin(self.device)


def edge_file_in_background(on: bool) -> None:
    """<prefix>.edge.gz of the graphs THIS THREAD begins is written beside pass 2 (the library's flag is thread-local: call it on the
    thread that calls graph_begin*, not on another one)."""
    lib().pg_host_edge_file_in_background(1 if on else 0)


def hip_free(ptr) -> None:
    """hipFree of a device pointer the library handed over (pg_export_take)."""
    lib().pg_device_free(ptr)


def host_skm_cut(packed: np.ndarray, n_reads: int, read_len: int, K: int, mer127: bool, log2_parts: int, ord_base: int, n_owners: int):
    """Host twin of pg_skm_route (the same inline code the kernels run, skm.hpp): the batch's super-k-mer records and, per
    record, partition << 8 | owner.  -> (records [n, W] uint64, tags [n] uint64)."""
    nw = 4 if mer127 else 2
    W = 6 if nw == 2 else 8
    cap = n_reads * (read_len - K + 1)
    recs = np.zeros((cap, W), dtype=np.uint64)
    tags = np.zeros(cap, dtype=np.uint64)
    packed = np.ascontiguousarray(packed, dtype=np.uint64)
    n = lib().pg_host_skm_cut(packed.ctypes.data, n_reads, read_len, K, int(mer127), log2_parts, ord_base, n_owners, recs.ctypes.data,
                              tags.ctypes.data, cap)
    if n < 0:
        raise PgError("pg_host_skm_cut failed: " + lib().pg_last_error().decode())
    return recs[:n], tags[:n]


def host_skm_expand(records: np.ndarray, K: int, mer127: bool):
    """Host twin of the record expansion of K2: every k-mer occurrence of the records as rows
    (key words..., left, right, ordinal)."""
    nw = 4 if mer127 else 2
    records = np.ascontiguousarray(records, dtype=np.uint64)
    cap = int(((records[:, 0] >> np.uint64(2)) & np.uint64(0xFFFF)).sum()) if len(records) else 0
    out = np.zeros((max(cap, 1), nw + 3), dtype=np.uint64)
    n = lib().pg_host_skm_expand(records.ctypes.data, records.shape[0], K, int(mer127), out.ctypes.data, cap)
    if n < 0:
        raise PgError("pg_host_skm_expand failed: " + lib().pg_last_error().decode())
    return out[:n]


class KmerCounter:
    """Pass-1 counting context on one GPU (pg_create ... pg_export)."""

    def __init__(self, K: int, n_sets: int = 8, mer127: bool = False, log2_slots: int = 24, device: int = 0, engine: int = 0):
        import torch  # noqa: F401  (must be loaded before the library, see lib())
        self.torch = torch
        self.K, self.P, self.mer127, self.device = K, n_sets, mer127, device
        self.nw = 4 if mer127 else 2
        self.h = (lib().pg_create_engine(device, K, int(mer127), n_sets, log2_slots, engine) if engine
                  else lib().pg_create(device, K, int(mer127), n_sets, log2_slots))
        if not self.h:
            raise PgError("pg_create failed: " + lib().pg_last_error().decode())

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def count_uniform(self, d_packed, n_reads: int, read_len: int, ord_base: int = 0) -> int:
        """d_packed: int64/uint64 CUDA tensor in the device read format.  Returns the k-mers in the batch."""
        n_kmers = n_reads * (read_len - self.K + 1)
        _check(lib().pg_count_reads(self.h, d_packed.data_ptr(), None, None, n_reads, read_len, n_kmers, ord_base,
                                    self._stream()), "pg_count_reads")
        return n_kmers

    def set_read_len_bound(self, max_len: int) -> None:
        """No read of the ragged batches to come is longer (0: unknown -- every ragged batch then asks the device and waits)."""
        _check(lib().pg_set_read_len_bound(self.h, max_len), "pg_set_read_len_bound")

    def count_ragged(self, d_packed, d_word_off, d_kmer_base, n_reads: int, n_kmers: int, ord_base: int = 0) -> int:
        _check(lib().pg_count_reads(self.h, d_packed.data_ptr(), d_word_off.data_ptr(), d_kmer_base.data_ptr(), n_reads, 0,
                                    n_kmers, ord_base, self._stream()), "pg_count_reads")
        return n_kmers

    def route_count(self, d_packed, n_reads: int, read_len: int, n_owners: int):
        t = self.torch
        counts = t.zeros(n_owners, dtype=t.int64, device=f"cuda:{self.device}")
        n_kmers = n_reads * (read_len - self.K + 1)
        _check(lib().pg_route_count(self.h, d_packed.data_ptr(), None, None, n_reads, read_len, n_kmers, n_owners,
                                    counts.data_ptr(), self._stream()), "pg_route_count")
        return counts

    def route_scatter(self, d_packed, n_reads: int, read_len: int, ord_base: int, n_owners: int, owner_off, out):
        t = self.torch
        cursor = t.zeros(n_owners, dtype=t.int64, device=f"cuda:{self.device}")
        n_kmers = n_reads * (read_len - self.K + 1)
        _check(lib().pg_route_scatter(self.h, d_packed.data_ptr(), None, None, n_reads, read_len, n_kmers, ord_base, n_owners,
                                      owner_off.data_ptr(), cursor.data_ptr(), out.data_ptr(), self._stream()), "pg_route_scatter")

    def record_words(self) -> int:
        return 6 if self.nw == 2 else 8

    def skm_route(self, d_packed, n_reads: int, read_len: int, ord_base: int, n_owners: int, cap: int):
        """Engine 2, multi-GPU step 1 -> (records [n_owners, cap, W] int64, parts [n_owners, cap] int32, counts [n_owners] int64)."""
        t = self.torch
        dev = f"cuda:{self.device}"
        recs = t.empty((n_owners, cap, self.record_words()), dtype=t.int64, device=dev)
        parts = t.empty((n_owners, cap), dtype=t.int32, device=dev)
        counts = t.zeros(n_owners, dtype=t.int64, device=dev)
        _check(lib().pg_skm_route(self.h, d_packed.data_ptr(), n_reads, read_len, ord_base, n_owners, recs.data_ptr(), parts.data_ptr(),
                                  cap, counts.data_ptr(), self._stream()), "pg_skm_route")
        return recs, parts, counts

    def skm_ingest(self, d_records, d_parts, n_records: int) -> None:
        _check(lib().pg_skm_ingest(self.h, d_records.data_ptr(), d_parts.data_ptr(), n_records, self._stream()), "pg_skm_ingest")

    def count_records(self, d_records, n_records: int) -> None:
        _check(lib().pg_count_records(self.h, d_records.data_ptr(), n_records, self._stream()), "pg_count_records")

    def count_sharded(self, comm: "Comm", d_packed, n_reads: int, read_len: int, ord_base: int = 0, stream=None) -> None:
        """One round of multi-GPU pass 1 (collective over comm): cut, all-to-all, append.  d_packed may be None with n_reads = 0."""
        st = stream if stream is not None else self._stream()
        _check(lib().pg_count_reads_sharded(self.h, comm.h, d_packed.data_ptr() if d_packed is not None else None, None, None, n_reads,
                                            read_len if n_reads else 0, 0, ord_base, st), "pg_count_reads_sharded")

    def set_autogrow(self, on: bool) -> None:
        _check(lib().pg_set_autogrow(self.h, int(on)), "pg_set_autogrow")

    def reset(self) -> None:
        _check(lib().pg_reset(self.h, self._stream()), "pg_reset")

    def distinct(self) -> int:
        out = C.c_uint64(0)
        _check(lib().pg_distinct(self.h, C.byref(out), self._stream()), "pg_distinct")
        return out.value

    def table_info(self):
        s, b = C.c_uint64(0), C.c_uint32(0)
        _check(lib().pg_table_info(self.h, C.byref(s), C.byref(b)), "pg_table_info")
        return s.value, b.value

    def finalize(self, delow: int = 0, want_last_put: bool = True):
        hist = np.zeros(256, dtype=np.uint64)
        last = np.zeros(self.P, dtype=np.uint64)
        _check(lib().pg_finalize(self.h, delow, hist.ctypes.data, last.ctypes.data if want_last_put else None, self._stream()),
               "pg_finalize")
        return hist, last

    def set_counts(self) -> np.ndarray:
        out = np.zeros(256, dtype=np.uint64)
        _check(lib().pg_set_counts(self.h, out.ctypes.data, self._stream()), "pg_set_counts")
        return out[: self.P]

    def last_put(self) -> np.ndarray:
        out = np.zeros(self.P, dtype=np.uint64)
        _check(lib().pg_last_put(self.h, out.ctypes.data, self._stream()), "pg_last_put")
        return out

    def stats(self) -> dict:
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().pg_stats(self.h, out.ctypes.data), "pg_stats")
        keys = ["engine", "distinct", "records", "unit_bytes", "pool_used", "pool_chunks", "parts_or_slots", "export_capacity"]
        return {k: int(v) for k, v in zip(keys, out)}

    def checksum(self) -> np.ndarray:
        """Order-independent digest of the distinct k-mers after finalize (pg_records_checksum on the export array in place):
        [column sums mod 2^64 (nw + 2 of them, zero-padded to 6), sum of the coverage fields, saturated nodes]."""
        ptr, n = C.c_void_p(0), C.c_uint64(0)
        _check(lib().pg_export_peek(self.h, C.byref(ptr), C.byref(n)), "pg_export_peek")
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().pg_records_checksum(ptr, n.value, self.nw + 2, out.ctypes.data, self._stream()), "pg_records_checksum")
        return out

    def export(self, sort: bool = False) -> np.ndarray:
        """(n, nw + 2) uint64 records on the host (key words, cnt, set << 56 | first ordinal); with sort=True in the layout
        replay's insertion order (pg_sort_records on the device)."""
        t = self.torch
        n = self.distinct()
        rw = self.nw + 2
        d = t.empty(max(n, 1) * rw, dtype=t.int64, device=f"cuda:{self.device}")
        got = C.c_uint64(0)
        _check(lib().pg_export(self.h, d.data_ptr(), n, C.byref(got), self._stream()), "pg_export")
        assert got.value == n
        if sort:
            with t.cuda.device(self.device):
                _check(lib().pg_sort_records(d.data_ptr(), n, int(self.nw == 4), self._stream()), "pg_sort_records")
        return d[: n * rw].cpu().numpy().view(np.uint64).reshape(n, rw)

    def close(self) -> None:
        if self.h:
            lib().pg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
