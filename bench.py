#!/usr/bin/env python3
"""bench.py -- pass 1 of `pregraph` (k-mer extraction + hash-set insert) on synthetic reads resident in HBM.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One step = one full pass of the hot path over the workload: empty the k-mer set, then extract and insert every
k-mer occurrence of every read (for N > 1: extract + route by owner, RCCL all-to-all, insert).  Reads are
generated on the GPU before the timed region and stay resident in HBM; FASTQ parsing and PCIe are not in `value`.
Workload = BASELINE.json configs[2] ("C. elegans-scale 200M x 150 bp synthetic, K=63, 1xMI355X"), the
configuration the metric (K = 63) is quoted on; per-GPU work is fixed as N grows (weak scaling).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def gen_packed_reads(torch, dev, genome_len, n_reads, read_len, err, seed, chunk=2_000_000):
    """Synthetic reads of SURVEY.md 8d drawn on the GPU (uniform genome, uniform starts, strand flip p=0.5,
    substitution errors), packed 2 bit/base MSB-first, word-aligned: int64 tensor [n_reads * wpr + 8]."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=g)
    wpr = (read_len + 31) // 32
    out = torch.zeros(n_reads * wpr + 8, dtype=torch.int64, device=dev)
    ar = torch.arange(read_len, device=dev, dtype=torch.int64)
    shifts = (62 - 2 * torch.arange(32, device=dev, dtype=torch.int64))
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        starts = torch.randint(0, genome_len - read_len, (n,), device=dev, generator=g, dtype=torch.int64)
        reads = genome[starts[:, None] + ar[None, :]]
        flip = torch.rand(n, device=dev, generator=g) < 0.5
        rc = torch.flip(reads, dims=[1]) ^ 2
        reads = torch.where(flip[:, None], rc, reads)
        if err > 0:
            mask = torch.rand(reads.shape, device=dev, generator=g) < err
            shift = torch.randint(1, 4, reads.shape, device=dev, generator=g, dtype=torch.uint8)
            reads = torch.where(mask, (reads + shift) & 3, reads)
        padded = torch.zeros((n, wpr * 32), dtype=torch.int64, device=dev)
        padded[:, :read_len] = reads.to(torch.int64)
        words = (padded.view(n, wpr, 32) << shifts[None, None, :]).sum(dim=2)
        out[lo * wpr:(lo + n) * wpr] = words.view(-1)
        del reads, rc, padded, words, starts, flip
    return out


def cpu_baseline(args, cores):
    """The reference's own pthreaded pregraph (oracle/_ref, built from /root/reference by oracle/Makefile.ref)
    timed on this box's host cores on a bounded sample of the same read distribution; pass-1 time is the
    reference's own 'Time spent on hashing reads' line."""
    from soapdenovo2_amd import synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    n = args.cpu_sample_reads
    with tempfile.TemporaryDirectory() as td:
        g = min(args.genome, 20_000_000)
        cfg = synth.make_case(td, "cpu", g, n, args.read_len, args.err, args.seed + 1)
        if os.path.exists(ref):
            t0 = time.time()
            out = subprocess.run([ref, "pregraph", "-s", cfg, "-K", str(args.kmer), "-o", os.path.join(td, "o"), "-p", str(cores)],
                                 capture_output=True, text=True)
            wall = time.time() - t0
            m = re.search(r"Time spent on hashing reads: (\d+)s", out.stderr)
            m2 = re.search(r"Time spent on pre-graph construction: (\d+)s", out.stderr)
            sec = float(m.group(1)) if m else None
            if not sec:
                sec = float(m2.group(1)) if m2 and float(m2.group(1)) > 0 else wall
            return {"value": n / sec, "unit": "reads/s", "cores": cores, "kind": "reference",
                    "sample": f"{n} reads x {args.read_len} bp, genome {g}, err {args.err}, K={args.kmer}, -p {cores}; "
                              f"pass 1 {sec:.0f} s of {wall:.0f} s whole command"}
        # fall back to the single-threaded C restatement
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_binding import Oracle
        codes = synth.reads_codes(g, n // 8, args.read_len, args.err, args.seed + 1)
        o = Oracle(args.kmer, P=8, max_read_len=args.read_len)
        t0 = time.time()
        o.add_reads(codes)
        sec = time.time() - t0
        o.close()
        return {"value": (n // 8) / sec, "unit": "reads/s", "cores": 1, "kind": "port",
                "sample": f"{n // 8} reads x {args.read_len} bp, K={args.kmer}, oracle/pregraph_oracle.c pass 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=200_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--kmer", type=int, default=63)
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--err", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--sets", type=int, default=8, help="the reference's -p (k-mer sets)")
    ap.add_argument("--mer127", action="store_true", help="four-word k-mers (the SOAPdenovo-127mer flavour); implied by --kmer > 63")
    ap.add_argument("--batch-reads", type=int, default=16_000_000)
    ap.add_argument("--log2-slots", type=int, default=0, help="0 = size from the expected distinct count")
    ap.add_argument("--cpu-sample-reads", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--engine", type=int, default=2, help="2 = super-k-mer partitions counted in LDS (default), 1 = one DRAM-resident set")
    ap.add_argument("--comm", default="nccl", help="nccl (RCCL over xGMI) or gloo (test only: exchange staged through the host)")
    ap.add_argument("--share-gpu", action="store_true", help="test only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from soapdenovo2_amd import api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.comm == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    def a2a(out_t, in_t, out_splits, in_splits):
        """variable-size all-to-all of int64/int32 device tensors (RCCL; gloo goes through host copies)"""
        if args.comm == "nccl":
            dist.all_to_all_single(out_t, in_t, out_splits, in_splits)
        else:
            o = torch.empty(out_t.shape, dtype=out_t.dtype)
            dist.all_to_all_single(o, in_t.cpu(), out_splits, in_splits)
            out_t.copy_(o)

    K, L, P = args.kmer, args.read_len, args.sets
    kpr = L - K + 1
    kpr_w = K - max(7, min(16, K - 6)) + 1          # m-mers per k-mer (skm.hpp): a read makes about 2*kpr/(w+1) + 1 records
    n_reads = args.reads
    n_kmers = n_reads * kpr
    # expected distinct k-mers per GPU: genomic (<= genome) + error k-mers (~ K per error, capped by read geometry)
    exp_err = n_reads * L * args.err * min(K, kpr)
    expected = min(n_kmers, args.genome + exp_err) * (1.0 if world == 1 else 1.15)
    log2_slots = args.log2_slots
    if not log2_slots:
        log2_slots = 20
        while (1 << log2_slots) * 0.6 < expected:
            log2_slots += 1
    packed = gen_packed_reads(torch, dev, args.genome, n_reads, L, args.err, args.seed + 1000 * rank)
    engine = args.engine
    mer127 = args.mer127 or K > 63
    kc = api.KmerCounter(K, n_sets=P, mer127=mer127, log2_slots=log2_slots, device=local, engine=engine)
    kc.set_autogrow(False)
    wpr = (L + 31) // 32
    batches = [(lo, min(args.batch_reads, n_reads - lo)) for lo in range(0, n_reads, args.batch_reads)]
    ord0 = rank * n_kmers
    ev, ev2 = [], []

    def step(timed):
        kc.reset()
        for lo, n in batches:
            view = packed[lo * wpr:]
            if world == 1:
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                kc.count_uniform(view, n, L, ord0 + lo * kpr)
                if timed:
                    e1.record()
                    ev.append((e0, e1, n))
            elif engine == 2:
                # partition engine across GPUs: owner(partition) = partition mod world; super-k-mer records travel
                rw = kc.record_words()
                cap = int(n * (2.0 * kpr / (kpr_w + 1) + 1) * 1.5 / world) + 1024
                recs, parts, counts = kc.skm_route(view, n, L, ord0 + lo * kpr, world, cap)
                recv_counts = torch.empty_like(counts)
                a2a(recv_counts, counts, None, None)
                sc, rc = counts.tolist(), recv_counts.tolist()
                send_r = torch.cat([recs[o, :sc[o]].reshape(-1) for o in range(world)])
                send_p = torch.cat([parts[o, :sc[o]] for o in range(world)])
                in_r = torch.empty(sum(rc) * rw, dtype=torch.int64, device=dev)
                in_p = torch.empty(sum(rc), dtype=torch.int32, device=dev)
                a2a(in_r, send_r, [c * rw for c in rc], [c * rw for c in sc])
                a2a(in_p, send_p, rc, sc)
                kc.skm_ingest(in_r, in_p, sum(rc))
            else:
                counts = kc.route_count(view, n, L, world)
                off = torch.zeros(world + 1, dtype=torch.int64, device=dev)
                off[1:] = torch.cumsum(counts, 0)
                send_counts = counts.clone()
                recv_counts = torch.empty_like(send_counts)
                a2a(recv_counts, send_counts, None, None)
                sc, rc = send_counts.tolist(), recv_counts.tolist()
                rw = kc.nw + 1
                out = torch.empty(sum(sc) * rw, dtype=torch.int64, device=dev)
                kc.route_scatter(view, n, L, ord0 + lo * kpr, world, off, out)
                inp = torch.empty(sum(rc) * rw, dtype=torch.int64, device=dev)
                a2a(inp, out, [c * rw for c in rc], [c * rw for c in sc])
                kc.count_records(inp, sum(rc))
        # -d filter + linear marking + coverage histogram (for the partition engine this is also where counting happens)
        if timed:
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
        kc.finalize(0, want_last_put=False)
        if timed:
            f1.record()
            ev2.append((f0, f1))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    distinct = kc.distinct()                       # also raises if the set overflowed
    if world > 1:
        t = torch.tensor([distinct], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        distinct = int(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        total_reads = n_reads * world
        rec = {
            "metric": f"pregraph_pass1_reads_per_sec_K{K}", "value": total_reads / (dt / args.steps), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "distinct_kmers_per_sec": distinct / (dt / args.steps),
            "kmer_occurrences_per_sec": n_kmers * world / (dt / args.steps),
            "config": {"workload": f"C. elegans-scale synthetic: {n_reads} reads/GPU x {L} bp, genome {args.genome} bp, err {args.err}, "
                                   f"K={K}, -p {P} sets (BASELINE.json configs[2])",
                       "reads_per_gpu": n_reads, "read_len": L, "K": K, "genome": args.genome, "err": args.err,
                       "distinct_kmers": distinct, "table_slots_log2": log2_slots,
                       "engine": engine,
                       "parallelism": ("single GPU, " + ("super-k-mer partitions counted in LDS" if engine == 2 else "fused extract+insert into one DRAM set"))
                       if world == 1 else (f"partition-owner (partition mod {world}), super-k-mer records over RCCL all-to-all" if engine == 2
                                           else f"set-id owner, k-mer records over RCCL all-to-all x{world}")},
        }
        if world == 1 and ev:
            slot_b = 80 if mer127 else 48
            bytes_per_read = kpr * slot_b + (L + 3) // 4          # node read + node write per occurrence + packed read
            dur = [e0.elapsed_time(e1) * 1e-3 for e0, e1, _ in ev]
            dur2 = [f0.elapsed_time(f1) * 1e-3 for f0, f1 in ev2]
            traffic = None
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            nwk = "4" if mer127 else "2"
            if engine == 1:
                kernel = f"count_reads_kernel<{nwk}>"
                alg = [n * bytes_per_read for _, _, n in ev]
                achieved = sum(alg) / sum(dur) / 1e9
                launches, avg_ms, per_launch = len(ev), sum(dur) / len(dur) * 1e3, sum(alg) / len(alg)
                extra = {}
            else:
                # Partition engine: two kernels share the pass.  K1 (skm_scatter_tiled_kernel) reads the packed reads and
                # writes super-k-mer records; K2 (skm_count_kernel) reads the records, counts every partition in LDS and writes
                # the distinct k-mers.  `achieved` follows the contract: SURVEY.md 8d's algorithmic bytes per read (one node
                # read + one node write per k-mer occurrence + the packed read) x the reads one launch of the dominant kernel
                # processes / its duration.  The bytes that formulation really moves (passes x record bytes) are listed too.
                st = kc.stats()
                rec_bytes = st["records"] * st["unit_bytes"]
                k1_bytes = n_reads * wpr * 8 + rec_bytes
                k2_bytes = rec_bytes + distinct * (kc.nw + 2) * 8
                k1_s, k2_s = sum(dur) / args.steps, sum(dur2) / args.steps
                alg_pass = n_reads * bytes_per_read
                if k1_s >= k2_s:
                    kernel, achieved = f"skm_scatter_tiled_kernel<{nwk}>", alg_pass / k1_s / 1e9
                    launches, avg_ms, per_launch = len(ev), sum(dur) / len(dur) * 1e3, alg_pass / len(batches)
                else:
                    kernel, achieved = f"skm_count_kernel<{nwk}>", alg_pass / k2_s / 1e9
                    launches, avg_ms, per_launch = len(ev2), k2_s * 1e3, alg_pass
                extra = {"definition": "SURVEY.md 8d algorithmic bytes/read x reads per launch / launch time of the dominant kernel",
                         "pass1_both_kernels_GBps": alg_pass / (k1_s + k2_s) / 1e9, "pass1_both_kernels_frac": alg_pass / (k1_s + k2_s) / 1e9 / 8000.0,
                         "k1_scatter_ms_per_step": k1_s * 1e3, "k2_count_ms_per_step": k2_s * 1e3,
                         "k1_moved_GBps": k1_bytes / k1_s / 1e9, "k2_moved_GBps": k2_bytes / k2_s / 1e9,
                         "moved_bytes_per_step": k1_bytes + k2_bytes, "moved_over_algorithmic": (k1_bytes + k2_bytes) / alg_pass,
                         "records_per_read": st["records"] / n_reads, "record_bytes": st["unit_bytes"], "partitions": st["parts_or_slots"]}
            if os.path.exists(tf):
                # PMC bytes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes) are kept per read in profiles/pmc_traffic.json
                # (measured on the 20 M-read variant of this workload) and scaled to the reads of one launch
                try:
                    tj = json.load(open(tf))
                    kn = kernel.split("<")[0]
                    if kn + "_bytes_per_read" in tj:
                        reads_per_launch = n_reads if kn == "skm_count_kernel" else n_reads / len(batches)
                        traffic = tj[kn + "_bytes_per_read"] * reads_per_launch
                    else:
                        traffic = tj.get(kn + "_bytes_per_launch")
                except Exception:
                    traffic = None
            rec["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                               "traffic": traffic, "kernel": kernel, "launches": launches, "avg_launch_ms": avg_ms,
                               "algorithmic_bytes_per_launch": per_launch, **extra}
            if not args.no_cpu_baseline:
                # the reference's workers each scan the whole k-mer buffer (prlHashReads.c:79-90), so its pass 1 stops
                # scaling long before a 100+-core host is used up: cap -p at 16 and say so
                rec["cpu_baseline"] = cpu_baseline(args, min(os.cpu_count() or 1, 16))
        print(json.dumps(rec), flush=True)
    kc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
