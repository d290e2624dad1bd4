"""The device graph stages (csrc/dev_graph.hpp, csrc/dev_tips.hpp) run on the HostBackend -- the same function objects the HIP
kernels run, on host threads -- against a plain model and against the sequential host stages.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from conftest import case_codes, oracle_records, host_runs
from soapdenovo2_amd import api

EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fcfs_model(keys, S, nw):
    """put_kmerset into a table that never grows (newhash.c:487-528): first empty slot at or after key mod size, in arrival order."""
    table = [None] * S
    for i, k in enumerate(keys):
        v = 0
        for w in k:
            v = (v << 64) | int(w)
        h = v % S                                   # exact for the 63-mer modulus; the 127-mer chain of 32-bit chunks is a true modulus while S < 2^32
        while table[h] is not None:
            h = (h + 1) % S
        table[h] = i
    return table


def _emu_layout(rec, per_set, S, nw, threads):
    P = len(per_set)
    out = np.zeros((P * S, nw + 1), dtype=np.uint64)
    cnt = np.array(per_set, dtype=np.uint64)
    rc = api.lib().pg_host_emu_layout_static(rec.ctypes.data, cnt.ctypes.data, P, S, int(nw == 4), threads, out.ctypes.data)
    return rc, out


@pytest.mark.parametrize("nw", [2, 4])
@pytest.mark.parametrize("S,n,threads", [(1031, 600, 1), (1031, 1000, 4), (257, 250, 3), (4099, 3000, 8), (97, 96, 2)])
def test_layout_static_equals_first_come_first_served_probing(nw, S, n, threads):
    """Random keys, several sets, loads up to 99 %: at these loads the last probe cluster wraps around the end of the table in
    most sets, which is the rotated-frame path of layout_static."""
    rng = np.random.default_rng(S * 7 + n + nw)
    P = 3
    per_set = [n, 0, max(1, n // 2)]
    total = sum(per_set)
    rec = np.zeros((total, nw + 2), dtype=np.uint64)
    rec[:, :nw] = rng.integers(0, 1 << 62, size=(total, nw), dtype=np.uint64)
    rec[:, 0] >>= np.uint64(3)                       # K <= 63 / 127 leaves the top bits of word 0 clear
    # homes concentrated near the end of the table in set 0, so that its last cluster certainly wraps
    for i in range(per_set[0] // 4):
        v = 0
        for w in rec[i, :nw]:
            v = (v << 64) | int(w)
        want = S - 1 - (i % 5)
        v += (want - v % S) % S
        for w in range(nw - 1, -1, -1):
            rec[i, w] = np.uint64(v & 0xFFFFFFFFFFFFFFFF); v >>= 64
    rec[:, nw] = np.arange(total, dtype=np.uint64) + np.uint64(1000)            # cnt: anything recognisable
    at = 0
    for s, c in enumerate(per_set):
        rec[at:at + c, nw + 1] = (np.uint64(s) << np.uint64(56)) | np.arange(c, dtype=np.uint64)
        at += c
    rc, out = _emu_layout(rec, per_set, S, nw, threads)
    assert rc == 0
    at = 0
    wrapped = False
    for s, c in enumerate(per_set):
        table = _fcfs_model(rec[at:at + c, :nw], S, nw)
        img = out[s * S:(s + 1) * S]
        for slot in range(S):
            if table[slot] is None:
                assert img[slot, 0] == EMPTY, (s, slot)
            else:
                assert (img[slot] == rec[at + table[slot], :nw + 1]).all(), (s, slot)
        wrapped = wrapped or (c and table[S - 1] is not None and table[0] is not None)
        at += c
    assert wrapped


def test_layout_static_refuses_a_full_pool():
    rec = np.zeros((97, 4), dtype=np.uint64)
    rec[:, 1] = np.arange(97, dtype=np.uint64)
    rc, _ = _emu_layout(rec, [97], 97, 2, 1)
    assert rc == 1                                   # unsuited: the caller replays on the host (which reports the exploded pool)


@pytest.mark.parametrize("name,P,a,m", [("t6k_k31", 2, 1, False), ("t8k_k63", 2, 1, True), ("t6k_k127", 3, 1, True)])
def test_layout_static_equals_the_host_replay_on_golden_cases(golden, tmp_path, name, P, a, m):
    """-a pools of the golden cases: the emulated device layout puts every k-mer into the slot the sequential host replay
    (pinned slot by slot on the oracle, tests/test_host_graph.py) puts it."""
    c = golden["cases"][name]
    codes = case_codes(c)
    rec, last, K = oracle_records(codes, c["K"], P, mer127=m, a_gb=a, prefix=str(tmp_path / "o"))
    nw = 4 if m else 2
    rec = rec[np.argsort(rec[:, nw + 1], kind="stable")]                    # replay order: (set, first ordinal)
    slots, sizes = api.host_replay_layout(rec, last, P, mer127=m, a_gb=a)
    S = int(sizes[0])
    assert all(int(x) == S for x in sizes)
    per_set = [int(((rec[:, nw + 1] >> np.uint64(56)) == np.uint64(s)).sum()) for s in range(P)]
    # the image is P * S slots of 24 / 40 bytes: a few GB at -a 1 -- keep the comparison to the occupied slots
    out = np.zeros((P * S, nw + 1), dtype=np.uint64)
    cnt = np.array(per_set, dtype=np.uint64)
    rc = api.lib().pg_host_emu_layout_static(np.ascontiguousarray(rec).ctypes.data, cnt.ctypes.data, P, S, int(m), 4, out.ctypes.data)
    assert rc == 0
    sets = (rec[:, nw + 1] >> np.uint64(56)).astype(np.int64)
    got = out[sets * S + slots.astype(np.int64)]
    assert (got == rec[:, :nw + 1]).all()
    assert int((out[:, 0] != EMPTY).sum()) == len(rec)


@pytest.mark.parametrize("threads,places", [(1, 1), (5, 1), (5, 3)])
@pytest.mark.parametrize("name", ["t6k_k31", "t8k_k63", "t6k_k127", "t5k_k24", "m60k_k63", "d8k_k127", "r8k_k127", "d8k_k63", "m100k_k31"])
def test_device_tip_decisions_equal_the_sequential_scan(golden, tmp_path, name, threads, places, monkeypatch):
    """removeSingleTips / removeMinorTips as the device decides them (dev_tips.hpp on the HostBackend: the fixed point over start
    decisions) against the sequential slot-order scan (Graph::tip_scan, pinned on the reference's files by tests/test_host_graph.py):
    the same tips, and afterwards the same counter words in every node.  places = 3: the scans over the sets run "where the set
    lives" with a list and a counter per place that the lead gathers, as in a sharded run (backend.hpp)."""
    monkeypatch.setenv("PG_EMU_PLACES", str(places))
    c = golden["cases"][name]
    codes = case_codes(c)
    for run in host_runs(c):
        P, D, a, m = run
        rec, last, K = oracle_records(codes, c["K"], P, D=D, mer127=bool(m), a_gb=a, prefix=str(tmp_path / "o"))
        rec = np.ascontiguousarray(rec)
        out = np.zeros(8, dtype=np.uint64)
        rc = api.lib().pg_host_emu_clip_tips(rec.ctypes.data, len(rec), last.ctypes.data, K, int(bool(m)), P, int(D == 0), a, threads, out.ctypes.data)
        assert rc == 0, api.lib().pg_last_error()
        single_seq, minor_seq, single_dev, minor_dev, diff, rounds, cycles = (int(x) for x in out[:7])
        assert (single_dev, minor_dev) == (single_seq, minor_seq), (name, run)
        assert diff == 0, (name, run, diff)
        assert rounds >= cycles >= 1


def _growable_vs_replay(rec, last, P, m, threads):
    nw = 4 if m else 2
    rec = np.ascontiguousarray(rec[np.argsort(rec[:, nw + 1], kind="stable")])
    want_slots, want_sizes = api.host_replay_layout(rec, last, P, mer127=m, a_gb=0)
    slots = np.zeros(len(rec), dtype=np.uint64)
    sizes = np.zeros(P, dtype=np.uint64)
    rounds = np.zeros(P, dtype=np.uint64)
    cap = int(sum(int(x) for x in want_sizes)) + 7
    nodes = np.zeros((cap, nw + 1), dtype=np.uint64)
    rc = api.lib().pg_host_emu_layout_growable(rec.ctypes.data, len(rec), last.ctypes.data, int(m), P, threads, slots.ctypes.data, sizes.ctypes.data,
                                               rounds.ctypes.data, nodes.ctypes.data, cap)
    assert rc == 0, api.lib().pg_last_error()
    assert [int(x) for x in sizes] == [int(x) for x in want_sizes]
    bad = np.nonzero(slots != want_slots)[0]
    assert len(bad) == 0, (len(bad), bad[:5], slots[bad[:5]], want_slots[bad[:5]])
    # the image: every record's key and payload word in its slot, everything else empty
    base = np.concatenate([[0], np.cumsum(want_sizes.astype(np.uint64))]).astype(np.uint64)
    at = base[(rec[:, nw + 1] >> np.uint64(api.PG_ORD_BITS)).astype(np.int64)] + slots
    assert (nodes[at.astype(np.int64)] == rec[:, : nw + 1]).all()
    filled = np.zeros(cap, dtype=bool)
    filled[at.astype(np.int64)] = True
    assert (nodes[:cap - 7][~filled[:cap - 7], 0] == np.uint64(0xFFFFFFFFFFFFFFFF)).all() and int(filled.sum()) == len(rec)
    return rounds


@pytest.mark.parametrize("name,P,m", [("t6k_k31", 7, False), ("t8k_k63", 3, True), ("m60k_k63", 8, False), ("t6k_k127", 3, True), ("m100k_k31", 2, False)])
def test_layout_growable_equals_the_host_replay(golden, tmp_path, name, P, m):
    """The growable (-a 0) sets' layout as the device computes it -- the in-place rehash as a fixed point over insertion times,
    dev_rehash.hpp on the HostBackend -- puts every k-mer into the slot the sequential host replay puts it (which is pinned slot by
    slot on the oracle and through the golden files on the reference)."""
    c = golden["cases"][name]
    codes = case_codes(c)
    rec, last, K = oracle_records(codes, c["K"], P, mer127=m, prefix=str(tmp_path / "o"))
    rounds = _growable_vs_replay(rec, last, P, m, threads=4)
    assert int(rounds.max()) >= 2


@pytest.mark.parametrize("blind_max,dense_min", [(None, None), (0, None), (2000, None), (0, 1), (2000, 1), (0, 3000), (0, 0)])
def test_layout_growable_random_keys_and_the_trailing_duplicate(blind_max, dense_min, monkeypatch):
    """Random keys (no genome structure), set sizes right at the growth thresholds, with and without a duplicate put behind the
    last new key (newhash.c:477 tests the growth before it probes).  blind_max: up to which size the fixed point's rounds are
    launched eight at a time without a read-back (dev_rehash.hpp; default 2^18 keys: all of these sets; 0: none; 2000: some sizes of a set).
    dense_min: from which size the first round runs over a list of the cluster starts whose length stays on the device (default 2^18
    keys: none of these sets; 1: every size that is not launched blind; 0: never)."""
    if blind_max is not None:
        monkeypatch.setenv("PG_RH_BLIND_MAX", str(blind_max))
    if dense_min is not None:
        monkeypatch.setenv("PG_RH_DENSE_MIN", str(dense_min))
    rng = np.random.default_rng(77)
    for n in (1, 5, 793, 794, 795, 1590, 1591, 5000, 40000):
        for trailing in (False, True):
            rec = np.zeros((n, 4), dtype=np.uint64)
            rec[:, :2] = rng.integers(0, 1 << 62, size=(n, 2), dtype=np.uint64)
            rec[:, 0] >>= np.uint64(3)
            rec[:, 3] = np.arange(n, dtype=np.uint64) * np.uint64(3)
            last = np.array([int(rec[-1, 3]) + (5 if trailing else 1)], dtype=np.uint64)
            _growable_vs_replay(rec, last, 1, False, threads=3)


def test_layout_growable_random_keys_four_words():
    """The same with four-word keys (the 127-mer flavour: other initial size, chained 32-bit modulus for the home slot), several
    sets of different sizes in one call."""
    rng = np.random.default_rng(177)
    P = 3
    counts = [30000, 7, 12345]
    recs = []
    for s, n in enumerate(counts):
        r = np.zeros((n, 6), dtype=np.uint64)
        r[:, :4] = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        r[:, 0] >>= np.uint64(3)
        r[:, 5] = (np.arange(n, dtype=np.uint64) * np.uint64(2)) | (np.uint64(s) << np.uint64(api.PG_ORD_BITS))
        recs.append(r)
    rec = np.concatenate(recs)
    last = np.array([2 * n + 3 for n in counts], dtype=np.uint64)        # every set saw a duplicate put after its last new key
    rounds = _growable_vs_replay(rec, last, P, True, threads=4)
    assert int(rounds[0]) > int(rounds[1])


@pytest.mark.parametrize("mer127", [False, True])
def test_home_slot_by_reciprocal_equals_the_reference_modulus(mer127):
    """key mod size through the precomputed reciprocal (graph_lookup.hpp: rem128) against Python's integers: the exact 128-bit
    modulus of the 63-mer build, and the 127-mer build's 32-bit chunks folded in 64-bit arithmetic -- whose `t << 32` overflows
    once a set is larger than 2^32 slots (newhash.c:36-57); sizes from 1 to 2^63 - 1, keys incl. the extremes."""
    rng = np.random.default_rng(11)
    nw = 4 if mer127 else 2
    M = (1 << 64) - 1
    sizes = [1, 2, 3, 1031, 16777213, (1 << 32) - 1, 1 << 32, (1 << 32) + 15, 4294967311 * 3, (1 << 40) + 9, (1 << 62) + 1, (1 << 63) - 1, (1 << 63) - 25]
    sizes += [int(x) for x in rng.integers(1, 1 << 62, size=20)] + [int(x) for x in rng.integers(1, 1 << 34, size=20)]
    keys = rng.integers(0, 1 << 63, size=(4000, nw), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(4000, nw), dtype=np.uint64)
    keys[0] = 0
    keys[1] = M
    keys[2, :] = [M if i % 2 else 0 for i in range(nw)]
    keys = np.ascontiguousarray(keys)
    out = np.zeros(len(keys), dtype=np.uint64)
    for size in sizes:
        api._check(api.lib().pg_host_emu_home_slots(keys.ctypes.data, len(keys), int(mer127), size, out.ctypes.data), "pg_host_emu_home_slots")
        for i in range(0, len(keys), 7 if size > 5 else 1):
            w = [int(x) for x in keys[i]]
            if not mer127:
                want = ((w[0] << 64) | w[1]) % size
            else:
                t = w[0] % size
                for x in w[1:]:
                    t = (((t << 32) & M) | (x >> 32)) % size
                    t = (((t << 32) & M) | (x & 0xFFFFFFFF)) % size
                want = t
            assert int(out[i]) == want, (size, w)


def test_layout_growable_fuzz_over_thresholds_and_skewed_homes(monkeypatch):
    """Many small random sets against the host replay with the fixed point's three thresholds drawn at random (rounds launched blind, first
    round over a list of the cluster starts, the re-sweep list by appends or by a prefix sum), both key widths, one to four host threads, and every
    third set with small numbers for keys: their homes crowd (runs of consecutive slots), clusters are thousands of keys long -- the heap behind the
    sweep's six registers -- and wrap around the end of the table."""
    for seed in range(120):
        rng = np.random.default_rng(9000 + seed)
        monkeypatch.setenv("PG_RH_BLIND_MAX", str(rng.choice([0, 0, 500, 5000])))
        monkeypatch.setenv("PG_RH_DENSE_MIN", str(rng.choice([0, 1, 1, 700, 4000])))
        monkeypatch.setenv("PG_RH_LIST_SHIFT", str(rng.choice([0, 2, 5, 10])))
        n = int(rng.choice([3, 50, 700, 793, 794, 1591, 3000, 9000, 25000]))
        four = bool(rng.integers(0, 2))
        nw = 4 if four else 2
        rec = np.zeros((n, nw + 2), dtype=np.uint64)
        rec[:, :nw] = rng.integers(0, 1 << 62, size=(n, nw), dtype=np.uint64)
        if rng.integers(0, 3) == 0:                                  # small numbers: home = key while the table is larger than they are, key mod size after
            rec[:, : nw - 1] = 0
            rec[:, nw - 1] = rng.integers(0, 1 << int(rng.integers(8, 20)), size=n, dtype=np.uint64)
        rec[:, 0] >>= np.uint64(3)
        rec = rec[np.unique(rec[:, :nw], axis=0, return_index=True)[1]]
        rec = rec[rng.permutation(len(rec))]
        rec[:, nw + 1] = np.arange(len(rec), dtype=np.uint64) * np.uint64(3)
        last = np.array([int(rec[-1, nw + 1]) + (5 if rng.integers(0, 2) else 1)], dtype=np.uint64)
        _growable_vs_replay(rec, last, 1, four, threads=int(rng.integers(1, 5)))
