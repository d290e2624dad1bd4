#!/usr/bin/env python3
"""How many of the records K1 writes could share a reservation?  ("reserve a wave's records of one partition with one
returned atomic".)  The host twin of the cutter (pg_host_skm_cut: the same inline code the kernel runs, skm.hpp) cuts synthetic reads of the bench's
model into super-k-mer records; the script counts, per K1 tile (24 reads: one workgroup's) and per 64 consecutive records (a wave's, as the
kernel deals runs to lanes), the records whose partition id occurs more than once in that group -- the only ones a shared reservation could serve.
No GPU.   python scripts/k1_tile_partition_share.py [--reads 96000] [--out profiles/r06_k1_shared_reservation.json]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapdenovo2_amd import api, synth


def share(part, group_of):
    """fraction of records whose (group, partition) pair occurs more than once, and the atomics saved if each such pair took one"""
    key = group_of.astype(np.uint64) << np.uint64(32) | part.astype(np.uint64)
    _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    shared = cnt[inv] > 1
    return float(shared.mean()), float(1.0 - len(cnt) / len(key))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=96000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "cases": []}
    for K, mer127, read_len, genome, err, log2_parts, tile in ((63, False, 150, 100_000_000, 0.001, 21, 24), (31, False, 100, 4_600_000, 0.005, 17, 24), (127, True, 150, 100_000_000, 0.001, 21, 24)):
        codes = synth.reads_codes(genome, a.reads, read_len, err, 7)
        packed = api.pack_reads(codes) if hasattr(api, "pack_reads") else None
        if packed is None:
            wpr = (read_len + 31) // 32
            packed = np.zeros((a.reads, wpr), dtype=np.uint64)
            for i in range(read_len):
                packed[:, i >> 5] |= codes[:, i].astype(np.uint64) << np.uint64(62 - 2 * (i & 31))
        recs, tags = api.host_skm_cut(packed.reshape(-1), a.reads, read_len, K, mer127, log2_parts, 0, 1)
        part = (tags >> np.uint64(8)).astype(np.uint64)
        kpr = read_len - K + 1
        # the read a record comes from: records come out in read order, ~the same number a read -- an even spread is exact enough for a count of chance meetings
        n = len(part)
        per_read = n / a.reads
        read_of = (np.arange(n) / per_read).astype(np.int64)
        t_share, t_saved = share(part, read_of // tile)
        w_share, w_saved = share(part, np.arange(n) // 64)
        res["cases"].append({"K": K, "read_len": read_len, "reads": a.reads, "partitions": 1 << log2_parts, "records_per_read": round(per_read, 3),
                             "records_per_tile_of_%d_reads" % tile: round(per_read * tile, 1),
                             "tile": {"records_sharing_a_partition_with_another": round(t_share, 5), "atomics_saved_if_shared": round(t_saved, 5)},
                             "wave_of_64_records": {"records_sharing_a_partition_with_another": round(w_share, 5), "atomics_saved_if_shared": round(w_saved, 5)}})
        print(res["cases"][-1])
    res["reading"] = ("A tile's ~110 records (a wave's 64) fall into 2 M partitions (128 k at configs[1]): the pairs that meet are adjacent super-k-mers of one read whose minimizers "
                      "hash to the same partition, or chance.  A reservation shared by the records of one partition inside a wave or a workgroup would save that share of K1's returned atomics and "
                      "no more; the kernel is bound by their rate (910 M a pass at 16 G/s).  Not built.")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
