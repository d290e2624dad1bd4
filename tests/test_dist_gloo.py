"""The N > 1 exchange logic on CPU (gloo, world size 2): the owner partition + variable-size all-to-all that
bench.py / a multi-GPU pregraph run perform around pg_route_scatter / pg_count_records.  The device kernels are
stood in by the oracle (occurrence records of each rank's reads); what is tested is that after the exchange every
rank holds exactly the occurrences of the reference sets it owns, and that per-rank reduction + union equals the
single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, case_codes, oracle_records


def _occurrences(codes, K, P, ord_base):
    """(key_hi, key_lo, meta, set) per k-mer occurrence, via the oracle one read at a time."""
    from oracle_binding import Oracle
    o = Oracle(K, P=P, max_read_len=codes.shape[1])
    recs = []
    L = codes.shape[1]
    kpr = L - K + 1
    # brute force in numpy (small inputs): canonical k-mer + flanks per position, as SURVEY.md A.1
    def val(seq):
        v = 0
        for c in seq:
            v = (v << 2) | int(c)
        return v
    from oracle_binding import lib
    for r in range(codes.shape[0]):
        s = codes[r]
        for j in range(kpr):
            w = val(s[j:j + K])
            rc = val([(int(c) ^ 2) for c in s[j:j + K][::-1]])
            if w <= rc:
                key, left, right = w, (int(s[j - 1]) if j > 0 else 4), (int(s[j + K]) if j < L - K else 4)
            else:
                key, left, right = rc, ((int(s[j + K]) ^ 2) if j < L - K else 4), ((int(s[j - 1]) ^ 2) if j > 0 else 4)
            recs.append((key >> 64, key & ((1 << 64) - 1), ((ord_base + r * kpr + j) << 6) | (left << 3) | right))
    o.close()
    return recs


def _crc_set(hi, lo, P):
    import zlib
    # hash_kmer = CRC-32 with register init 0 and a final xor (hashFunction.c:123-131); zlib xors its start value, so
    # passing 0xFFFFFFFF starts the register at 0, and zlib applies the final xor itself
    crc = zlib.crc32(int(hi).to_bytes(8, "little") + int(lo).to_bytes(8, "little"), 0xFFFFFFFF) & 0xFFFFFFFF
    v = crc if crc < 0x80000000 else crc + 0xFFFFFFFF00000000
    return v % P


def _worker(rank, world, port, tmp, K, P):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from soapdenovo2_amd import synth
    codes = synth.reads_codes(3000, 60, 60, 0.01, 77)
    per = codes.shape[0] // world
    mine = codes[rank * per:(rank + 1) * per]
    kpr = codes.shape[1] - K + 1
    recs = _occurrences(mine, K, P, rank * per * kpr)
    owner = [_crc_set(h, l, P) % world for h, l, _ in recs]
    send = [[r for r, o in zip(recs, owner) if o == d] for d in range(world)]
    send_counts = torch.tensor([len(x) for x in send], dtype=torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    flat = [w for part in send for r in part for w in r]
    # uint64 payload travels as int64
    out = torch.from_numpy(np.array(flat, dtype=np.uint64).view(np.int64)) if flat else torch.empty(0, dtype=torch.int64)
    inp = torch.empty(int(recv_counts.sum()) * 3, dtype=torch.int64)
    dist.all_to_all_single(inp, out, [int(c) * 3 for c in recv_counts], [int(c) * 3 for c in send_counts])
    got = inp.numpy().view(np.uint64).reshape(-1, 3)
    np.save(os.path.join(tmp, f"recv{rank}.npy"), got)
    dist.barrier()
    dist.destroy_process_group()


def test_owner_exchange_world2(tmp_path):
    K, P, world = 21, 4, 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), K, P), nprocs=world, join=True)
    from soapdenovo2_amd import synth
    codes = synth.reads_codes(3000, 60, 60, 0.01, 77)
    want, last, _ = oracle_records(codes, K, P, prefix=str(tmp_path / "o"))
    parts = [np.load(str(tmp_path / f"recv{r}.npy")) for r in range(world)]
    assert sum(len(p) for p in parts) == codes.shape[0] * (codes.shape[1] - K + 1)
    seen = {}
    for r, p in enumerate(parts):
        for hi, lo, meta in p:
            s = _crc_set(int(hi), int(lo), P)
            assert s % world == r                          # every occurrence landed on the owner of its set
            k = (int(hi), int(lo))
            ordv = int(meta) >> 6
            c, first = seen.get(k, (0, 1 << 62))
            seen[k] = (c + 1, min(first, ordv))
    assert len(seen) == want.shape[0]
    for row in want:
        c, first = seen[(int(row[0]), int(row[1]))]
        assert min(c, 255) == (int(row[2]) & 0xFFFFFFFF) >> 24            # total coverage
        assert first == int(row[3]) & ((1 << 56) - 1)                      # first-occurrence ordinal
        assert int(row[3]) >> 56 == _crc_set(int(row[0]), int(row[1]), P)  # set id
