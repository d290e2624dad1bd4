"""The N > 1 pass-1 path on CPU (gloo, world sizes 2 and 3) through the library's own routing code.

Each rank cuts its share of the reads into super-k-mer records with pg_host_skm_cut -- the host twin of pg_skm_route: the
same inline functions the kernels run (csrc/skm.hpp), owner = minimizer partition mod world -- the records travel through
a variable-size all-to-all (what pg_exchange_records does over RCCL), and every rank expands what it received with
pg_host_skm_expand.  Checked: every k-mer occurrence of the input arrives exactly once; all occurrences of a canonical
k-mer meet on ONE rank; the per-rank reduction (saturating counters, first ordinal) put together equals the oracle's
single-process records bit for bit.  The oracle is the checker; nothing here restates the routing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, oracle_records

CASES = {"k21": (3000, 400, 60, 0.01, 77, 21, False), "k63": (4000, 300, 150, 0.005, 5, 63, False), "k127": (6000, 120, 250, 0.004, 9, 127, True)}


def _reduce(occ, nw):
    """occurrence rows (key words, left, right, ord) -> records (key words, cnt word, first ordinal) with the reference's
    saturating counters (newhash.c:74-140)."""
    if len(occ) == 0:
        return np.zeros((0, nw + 2), dtype=np.uint64)
    keys = occ[:, :nw]
    order = np.lexsort([keys[:, i] for i in range(nw - 1, -1, -1)])
    occ = occ[order]
    keys = occ[:, :nw]
    new = np.ones(len(occ), dtype=bool)
    new[1:] = (keys[1:] != keys[:-1]).any(axis=1)
    gid = np.cumsum(new) - 1
    n = int(gid[-1]) + 1
    out = np.zeros((n, nw + 2), dtype=np.uint64)
    out[:, :nw] = keys[new]
    puts = np.bincount(gid, minlength=n)
    A = (np.minimum(puts, 255).astype(np.uint64) << np.uint64(24))
    B = np.where(puts == 1, np.uint64(1 << 27), np.uint64(0))
    for c in range(4):
        l = np.minimum(np.bincount(gid, weights=(occ[:, nw] == c), minlength=n).astype(np.uint64), 63)
        r = np.minimum(np.bincount(gid, weights=(occ[:, nw + 1] == c), minlength=n).astype(np.uint64), 63)
        A |= l << np.uint64(6 * c)
        B |= r << np.uint64(6 * c)
    out[:, nw] = A | (B << np.uint64(32))
    first = np.full(n, np.iinfo(np.uint64).max, dtype=np.uint64)
    np.minimum.at(first, gid, occ[:, nw + 2])
    out[:, nw + 1] = first
    return out


def _worker(rank, world, port, tmp, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from soapdenovo2_amd import api, synth
    G, N, L, err, seed, K, m127 = CASES[case]
    codes = synth.reads_codes(G, N, L, err, seed)
    bounds = np.linspace(0, N, world + 1).astype(int)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    kpr = L - K + 1
    packed = api.pack_reads_uniform(codes[lo:hi])
    recs, tags = api.host_skm_cut(packed, hi - lo, L, K, m127, 7, lo * kpr, world)
    owner = (tags & np.uint64(0xFF)).astype(np.int64)
    assert ((tags >> np.uint64(8)).astype(np.int64) % world == owner).all()
    W = recs.shape[1]
    order = np.argsort(owner, kind="stable")                  # grouped by owner, as pg_skm_route writes them
    recs, owner = recs[order], owner[order]
    send_counts = torch.from_numpy(np.bincount(owner, minlength=world).astype(np.int64))
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    out = torch.from_numpy(recs.view(np.int64).reshape(-1).copy())
    inp = torch.empty(int(recv_counts.sum()) * W, dtype=torch.int64)
    dist.all_to_all_single(inp, out, [int(c) * W for c in recv_counts], [int(c) * W for c in send_counts])
    got = inp.numpy().view(np.uint64).reshape(-1, W)
    occ = api.host_skm_expand(got, K, m127)
    np.save(os.path.join(tmp, f"occ{rank}.npy"), occ)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("k21", 2), ("k63", 2), ("k63", 3), ("k127", 2)])
def test_partition_owner_exchange(tmp_path, case, world):
    from soapdenovo2_amd import synth
    G, N, L, err, seed, K, m127 = CASES[case]
    port = 29500 + (os.getpid() * 7 + world * 13 + len(case)) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), case), nprocs=world, join=True)
    codes = synth.reads_codes(G, N, L, err, seed)
    want, _, _ = oracle_records(codes, K, 1, mer127=m127, prefix=str(tmp_path / "o"))
    nw = 4 if m127 else 2
    parts = [np.load(str(tmp_path / f"occ{r}.npy")) for r in range(world)]
    assert sum(len(p) for p in parts) == N * (L - K + 1)                    # every occurrence arrived, once
    per_rank = [_reduce(p, nw) for p in parts]
    got = np.concatenate(per_rank)
    key = lambda r: r[np.lexsort([r[:, i] for i in range(nw - 1, -1, -1)])]
    g, w = key(got), key(want).copy()
    assert g.shape == w.shape                                                # no k-mer on two ranks
    w[:, nw + 1] &= np.uint64((1 << 56) - 1)                                # the oracle's set id (P = 1: zero anyway)
    w[:, nw] &= np.uint64(~(1 << (32 + 24)) & 0xFFFFFFFFFFFFFFFF)           # ... and the linear flag of finish_count
    assert (g == w).all()


# ---- the regroup after pass 1: distinct k-mers to the rank that owns their reference set (set s -> rank s mod world) ----------
def _regroup_worker(rank, world, port, tmp, P):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from soapdenovo2_amd import api
    full = np.load(os.path.join(tmp, "records.npy"))
    mine = np.ascontiguousarray(full[rank::world])                   # any split of the distinct k-mers will do for pass 1's owners
    rw = full.shape[1]
    counts = np.zeros(world, dtype=np.uint64)
    grouped = np.zeros_like(mine)
    assert api.lib().pg_host_regroup_plan(mine.ctypes.data, len(mine), rw, world, counts.ctypes.data, grouped.ctypes.data) == 0
    send_counts = torch.from_numpy(counts.astype(np.int64))
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    inp = torch.empty(int(recv_counts.sum()) * rw, dtype=torch.int64)
    dist.all_to_all_single(inp, torch.from_numpy(grouped.view(np.int64).reshape(-1).copy()), [int(c) * rw for c in recv_counts], [int(c) * rw for c in send_counts])
    got = inp.numpy().view(np.uint64).reshape(-1, rw)
    np.save(os.path.join(tmp, f"regrouped{rank}.npy"), got)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,P", [(2, 8), (3, 8), (3, 5)])
def test_regroup_by_set_owner(tmp_path, world, P):
    """Every rank ends up with exactly the k-mers of the sets it owns, nobody holds more than its share of sets, nothing is
    lost or doubled -- through pg_host_regroup_plan, the host twin of the device grouping step."""
    from soapdenovo2_amd import synth
    codes = synth.reads_codes(20000, 3000, 100, 0.005, 41)
    rec, _, _ = oracle_records(codes, 31, P, prefix=str(tmp_path / "o"))
    rec = rec[np.random.default_rng(1).permutation(len(rec))]
    np.save(str(tmp_path / "records.npy"), rec)
    port = 31500 + (os.getpid() * 5 + world * 17 + P) % 2000
    mp.spawn(_regroup_worker, args=(world, port, str(tmp_path), P), nprocs=world, join=True)
    parts = [np.load(str(tmp_path / f"regrouped{r}.npy")) for r in range(world)]
    nw = rec.shape[1] - 2
    sets_all = (rec[:, nw + 1] >> np.uint64(56)).astype(np.int64)
    for r, p in enumerate(parts):
        s = (p[:, nw + 1] >> np.uint64(56)).astype(np.int64)
        assert (s % world == r).all()
        assert len(p) == int((sets_all % world == r).sum())              # its sets, whole
        assert len(p) <= -(-P // world) * np.bincount(sets_all, minlength=P).max()
    got = np.concatenate(parts)
    key = lambda a: a[np.lexsort([a[:, i] for i in range(a.shape[1] - 1, -1, -1)])]
    assert (key(got) == key(rec)).all()
