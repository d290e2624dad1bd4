"""Host stages of the product (layout replay, tips, edges, writers: pg_host_build_graph) on CPU: fed with the
distinct k-mers the oracle counts (shuffled, as the device export order is unspecified) they must reproduce the
reference's .vertex / .edge.gz / .preGraphBasic byte for byte."""
import os

import numpy as np
import pytest

from conftest import case_codes, case_tag, md5_file, md5_gz_text, oracle_records, host_runs
from soapdenovo2_amd import api

CASES = ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63"]


@pytest.mark.parametrize("name", CASES)
def test_host_graph_matches_reference(golden, tmp_path, name):
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in host_runs(c):
        P, D, a, m = run
        t = case_tag(name, run)
        rec, last, K = oracle_records(codes, c["K"], P, D=D, mer127=bool(m), a_gb=a, prefix=str(tmp_path / ("o_" + t)))
        rec = rec[np.random.default_rng(5).permutation(len(rec))]
        pre = str(tmp_path / t)
        api.host_build_graph(rec, last, K, P, pre, mer127=bool(m), cut_single=(D == 0), a_gb=a, max_read_len=c["L"])
        want = golden["md5"][t]
        assert md5_file(pre + ".vertex") == want["vertex"], t
        assert md5_file(pre + ".preGraphBasic") == want["preGraphBasic"], t
        assert md5_gz_text(pre + ".edge.gz") == want["edge"], t


def test_layout_replay_slots(golden):
    """Slot-by-slot: the replayed layout equals the oracle's tables (which equal the reference's, since the
    oracle's .vertex order is pinned on it) for growable and static (-a) sets, both k-mer widths."""
    from oracle_binding import Oracle
    for name, P, a, m in (("t6k_k31", 7, 0, False), ("t6k_k31", 2, 1, False), ("t8k_k63", 3, 0, True), ("m60k_k63", 8, 0, False)):
        c = golden["cases"][name]
        codes = case_codes(c)
        o = Oracle(c["K"], P=P, a_gb=a, mer127=m, max_read_len=c["L"])
        o.add_reads(codes)
        nd = o.nodes()
        sizes_want = o.set_sizes()
        last = np.array(o.set_last_put(), dtype=np.uint64)
        o.close()
        nw = 4 if m else 2
        rec = np.zeros((len(nd["A"]), nw + 2), dtype=np.uint64)
        rec[:, :nw] = nd["keys"]
        rec[:, nw + 1] = (nd["set"].astype(np.uint64) << np.uint64(56)) | nd["ord"]
        perm = np.random.default_rng(3).permutation(len(rec))
        slots, sizes = api.host_replay_layout(rec[perm], last, P, mer127=m, a_gb=a)
        assert list(sizes) == sizes_want
        assert (slots == nd["slot"][perm]).all()


def test_growth_on_trailing_duplicate():
    """newhash.c:477 runs the growth test before the probe, so a duplicate arriving when count == max grows
    the set although nothing is inserted.  1031 * 0.77f -> max 793: exactly 793 distinct keys, then a repeat."""
    from oracle_binding import Oracle
    rng = np.random.default_rng(9)
    K = 31
    o = Oracle(K, P=1, max_read_len=K + 2)
    reads = []
    while o.L.oracle_node_count(o.h) + 2 <= 793:
        r = rng.integers(0, 4, size=(1, K + 1), dtype=np.uint8)
        o.add_reads(r)
        reads.append(r)
    while o.L.oracle_node_count(o.h) < 793:       # one k-mer at a time: a read of K + 1 whose first k-mer is known
        r = np.concatenate([reads[0][0][1:], rng.integers(0, 4, size=1, dtype=np.uint8)])[None, :]
        o.add_reads(r)
        reads.append(r)
    assert o.L.oracle_node_count(o.h) == 793 and o.set_sizes() == [1031]
    o.add_reads(reads[0])                         # pure duplicates: the set must grow now
    grown = o.set_sizes()[0]
    assert grown > 1031
    nd = o.nodes()
    last = np.array(o.set_last_put(), dtype=np.uint64)
    o.close()
    rec = np.zeros((793, 4), dtype=np.uint64)
    rec[:, :2] = nd["keys"]
    rec[:, 3] = nd["ord"]
    slots, sizes = api.host_replay_layout(rec, last, 1)
    assert sizes[0] == grown and (slots == nd["slot"]).all()
    # without the last-put information the replay cannot know about the trailing duplicate
    slots2, sizes2 = api.host_replay_layout(rec, np.zeros(1, dtype=np.uint64), 1)
    assert sizes2[0] == 1031


def test_last_put_matters_only_at_a_growth_threshold():
    """pg_host_last_put_matters: 1031 slots * 0.77f -> max 793, so a trailing duplicate grows a growable set holding
    exactly 793 keys (and nothing else does); -a pools never grow."""
    L = api.lib()
    u = lambda *v: np.array(v, dtype=np.uint64).ctypes.data
    assert L.pg_host_last_put_matters(u(793), 1, 0, 0) == 1
    assert L.pg_host_last_put_matters(u(792), 1, 0, 0) == 0
    assert L.pg_host_last_put_matters(u(794), 1, 0, 0) == 0
    assert L.pg_host_last_put_matters(u(10, 500, 793), 3, 0, 0) == 1
    assert L.pg_host_last_put_matters(u(793), 1, 2, 0) == 0
