#!/usr/bin/env python3
"""bench.py -- pass 1 of `pregraph` (k-mer extraction + hash-set insert) on synthetic reads resident in HBM.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One step = one full pass of the hot path over the workload: empty the k-mer set, then extract and insert every
k-mer occurrence of every read -- exactly the device work `call_pregraph` does for pass 1: K1 per batch (for N > 1:
pg_count_reads_sharded = cut + route by owner, RCCL all-to-all, append) and pg_finalize (K2: count every partition,
-d filter, linear marks, coverage histogram).  Reads are generated on the GPU before the timed region and stay
resident in HBM; FASTQ parsing and PCIe are not in `value`.  Next to it, measured once outside the timed steps and
reported in the same JSON line: the export + sort hand-over, a PCIe-inclusive pass (pinned host batches, copies
overlapped with the kernels), and `whole_command` -- the `SOAPdenovo-63mer pregraph` executable on a FASTQ prefix of
the same read distribution written to local disk, its stages, and byte equality of its five files with the
reference binary's (oracle/_ref), whose own timings are the `cpu_baseline`.
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


def workload_label(n_reads, L, genome, err, K, P, world):
    """config.workload, built from the arguments; a BASELINE.json config is named only when the arguments ARE that config."""
    named = {(10_000_000, 100, 4_600_000, 0.005, 31, 1): "BASELINE.json configs[1]", (200_000_000, 150, 100_000_000, 0.001, 63, 1): "BASELINE.json configs[2]",
             (375_000_000, 150, 3_100_000_000, 0.0005, 63, 8): "BASELINE.json configs[3]: 3 G reads over 8 GPUs", (375_000_000, 150, 3_100_000_000, 0.0005, 127, 8): "BASELINE.json configs[4]: 3 G reads over 8 GPUs"}
    name = named.get((n_reads, L, genome, err, K, world))
    return (f"synthetic (SURVEY.md 8d): {n_reads} reads/GPU x {L} bp, genome {genome} bp, err {err}, K={K}, -p {P} sets, {world} GPU(s)"
            + (f" = {name}" if name else " (not one of BASELINE.json's configs)"))


def library_source_sha():
    """What the counters in profiles/pmc_traffic.json were taken on: a hash of the sources of the two pass-1 kernels -- partition_kernels.hip and the
    headers its device code comes from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("partition_kernels.hip", "device_ctx.hpp", "extract.hpp", "kmer.hpp", "occ32.hpp", "skm.hpp", "skm_tile.hpp"):
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, "soapdenovo2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def md5_outputs(prefix):
    import gzip
    import hashlib
    out = {}
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        out[ext] = hashlib.md5(open(f"{prefix}.{ext}", "rb").read()).hexdigest()
    h = hashlib.md5()
    with gzip.open(prefix + ".edge.gz", "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    out["edge"] = h.hexdigest()
    return out


def whole_command(args, cores):
    """The `pregraph` executable end to end on a FASTQ prefix of the benchmarked read distribution (local disk), next to the
    reference's pthreaded pregraph (oracle/_ref, built from /root/reference by oracle/Makefile.ref) on the same file and the
    same host cores.  Returns (whole_command, cpu_baseline)."""
    from soapdenovo2_amd import api, synth
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer" if args.kmer > 63 else "SOAPdenovo-63mer")
    mine = api.binary(args.kmer > 63)
    n = args.whole_reads
    td = tempfile.mkdtemp(prefix="pgbench_", dir=os.environ.get("PG_BENCH_TMP"))
    try:
        t0 = time.time()
        fq, cfg = os.path.join(td, "reads.fq"), os.path.join(td, "lib.cfg")
        gen = os.path.join(ROOT, "soapdenovo2_amd", "bin", "synth_fastq")
        if os.path.exists(gen):
            # the C++ generator (scripts/synth_fastq.cpp): the same read distribution, and no GPU context in THIS process while the
            # executable runs -- a child that allocates its tens of gigabytes beside a parent holding a context on the same GPU waited
            # 2.5 - 4 s for them in round 3's driver run (device_context_s), 0.03 s on its own
            subprocess.check_call([gen, fq, str(args.genome), str(n), str(args.read_len), str(args.err), str(args.seed + 1)])
        else:
            codes = synth.gpu_reads_codes(args.genome, n, args.read_len, args.err, args.seed + 1)
            synth.write_fastq_fast(fq, codes)
            del codes
        synth.write_config(cfg, fq, args.read_len)
        os.sync()                                   # the generator's write-back is not part of the command being timed
        t_gen = time.time() - t0
        wc = {"workload": f"first {n} reads of the benchmarked distribution ({args.read_len} bp, genome {args.genome}, err {args.err}) as FASTQ on local disk, "
                          f"K={args.kmer}, -p {args.sets}", "reads": n, "fastq_bytes": os.path.getsize(fq), "generate_s": round(t_gen, 1)}
        base = [ "pregraph", "-s", cfg, "-K", str(args.kmer), "-p", str(args.sets)]
        t0 = time.time()
        out = subprocess.run([mine] + base + ["-o", os.path.join(td, "amd")], capture_output=True, text=True, env=dict(os.environ, PG_HOST_VERBOSE="1"))
        wall = time.time() - t0
        wc.update({"rc": out.returncode, "wall_s": wall, "reads_per_sec": n / wall if out.returncode == 0 else None,
                   "stages_s": {m.group(1): float(m.group(2)) for m in re.finditer(r"\[cli\] ([^:]+): ([0-9.]+)s", out.stderr)}})
        m = re.search(r"(\d+) node\(s\) allocated", out.stderr)
        if m:
            wc["distinct_kmers"] = int(m.group(1))
        if out.returncode != 0:
            wc["stderr_tail"] = out.stderr[-800:]
        cpu = None
        if os.path.exists(ref):
            t0 = time.time()
            r = subprocess.run([ref, "pregraph", "-s", cfg, "-K", str(args.kmer), "-p", str(cores), "-o", os.path.join(td, "ref")], capture_output=True, text=True)
            rwall = time.time() - t0
            m1 = re.search(r"Time spent on hashing reads: (\d+)s", r.stderr)
            sec = float(m1.group(1)) if m1 and float(m1.group(1)) > 0 else rwall
            cpu = {"value": n / sec, "unit": "reads/s", "cores": cores, "kind": "reference",
                   "sample": f"{n} reads x {args.read_len} bp, genome {args.genome}, err {args.err}, K={args.kmer}, -p {cores}: pass 1 ('Time spent on hashing "
                             f"reads') {sec:.0f} s of {rwall:.0f} s whole command", "whole_command_reads_per_sec": n / rwall}
            # the same reference binary on the headline workload at its FULL size (it cannot run inside a bench: 50 minutes): the committed run of the build
            # container, whose md5s the 200 M-read leg below is held against
            try:
                fr = json.load(open(os.path.join(ROOT, "profiles", "r04_ref_200M_K63_a40.json")))
                hs = next((float(re.search(r"hashing reads: (\d+)s", l).group(1)) for l in fr.get("log", []) if "hashing reads" in l), None)
                cpu["full_size_reference"] = {"reads": fr["workload"]["reads"], "read_len": fr["workload"]["read_len"], "K": fr["workload"]["kmer"], "threads": fr["workload"]["sets"],
                                              "whole_command_s": fr.get("reference_wall_s"), "pass1_s": hs,
                                              "pass1_reads_per_sec": fr["workload"]["reads"] / hs if hs else None, "where": "the build container's 8 cores (profiles/r04_ref_200M_K63_a40.json)"}
            except Exception:
                pass
            wc["reference_wall_s"] = rwall
            wc["reference_threads"] = cores
            wc["speedup_vs_reference"] = rwall / wall if out.returncode == 0 else None
            # (-p fixes the order of .vertex / .edge.gz, so the reference runs at the same -p as the executable: one run serves
            #  as the CPU baseline and as the byte-for-byte check)
            refp = os.path.join(td, "ref") if r.returncode == 0 else None
            if refp and out.returncode == 0:
                a, b = md5_outputs(os.path.join(td, "amd")), md5_outputs(refp)
                wc["files_identical_to_reference"] = a == b
                wc["md5"] = a
        return wc, cpu
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


def big_command(args):
    """The executable on FASTQ files of scripts/synth_fastq.cpp (bytes that depend on the arguments only), compared with the md5s of the
    REFERENCE's own runs on the same files, which took it a quarter of an hour to an hour each in the build container and are committed
    (profiles/r0*_ref_*.json): 60 M reads at K = 63 with -a 16 (static pools) and with the default growable sets -- either way the
    k-mer-set layout is made on the device (SURVEY.md App. C "K6": dev_graph.hpp / dev_rehash.hpp); 20 M reads at K = 127 (the
    SOAPdenovo-127mer flavour, configs[4]'s path); configs[2] at its full 200 M reads with -a 40 and with growable sets (the reference: 2982 s / 5562 s).
    BASELINE.json configs[1] at its full size (10 M x 100 bp over 4.6 Mb, err 0.005, K = 31, -p 8; the reference: profiles/r05_ref_10M_K31.json -- the reads are
    scripts/synth_fastq.cpp's at seed 7, 95 807 825 distinct k-mers, not SURVEY.md's numpy draw of the same model (95 803 852): the reference was run on THIS file).
    Returns {"whole_command_10M_k31": {...}, "whole_command_60M_a16": {...}, "whole_command_60M": {...}, "whole_command_k127_20M": {...}, "whole_command_200M_a40": {...},
    "whole_command_200M_a0": {...}, "whole_command_20M_ragged": {...}}."""
    groups = [[("whole_command_10M_k31", "r05_ref_10M_K31.json")],
              [("whole_command_60M_a16", "r03_ref_60M_K63_a16.json"), ("whole_command_60M", "r03_ref_60M_K63.json")],
              [("whole_command_k127_20M", "r04_ref_20M_K127.json")],
              # (round 6: configs[2] at its full size with GROWABLE sets too -- the layout of 143 M-key sets through every size they live through, on the device)
              [("whole_command_200M_a40", "r04_ref_200M_K63_a40.json"), ("whole_command_200M_a0", "r06_ref_200M_K63_a0.json")],
              # round 6: trimmed reads, lengths uniform in [100, 150] (synth_fastq's min_len) -- every batch ragged, cut by the tiled K1 and
              # threaded by pass 2 where pass 1 left them; the reference chops reads of any length alike (prlHashReads.c:163-259,642-648)
              [("whole_command_20M_ragged", "r06_ref_20M_ragged_K63.json")]]
    gen = os.path.join(ROOT, "soapdenovo2_amd", "bin", "synth_fastq")
    if not os.path.exists(gen):
        return None
    from soapdenovo2_amd import api
    res = {}
    for group in groups:
        exps = [(k, os.path.join(ROOT, "profiles", f)) for k, f in group if os.path.exists(os.path.join(ROOT, "profiles", f))]
        if not exps:
            continue
        w = json.load(open(exps[0][1]))["workload"]
        if args.sets != w["sets"]:
            continue
        td = tempfile.mkdtemp(prefix="pgbig_", dir=os.environ.get("PG_BENCH_TMP"))
        try:
            if shutil_free_gb(td) < w["reads"] * (2 * w["read_len"] + 20) / 1e9 + 8:
                res[exps[0][0]] = {"skipped": "not enough room for the FASTQ in " + td}
                continue
            fq, cfg = os.path.join(td, "reads.fq"), os.path.join(td, "lib.cfg")
            t0 = time.time()
            subprocess.check_call([gen, fq, str(w["genome"]), str(w["reads"]), str(w["read_len"]), str(w["err"]), str(w["seed"])]
                                  + ([str(os.cpu_count() or 8), str(w["min_len"])] if w.get("min_len") else []))
            open(cfg, "w").write(f"max_rd_len={w['read_len']}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nrank=1\nq={fq}\n")
            os.sync()
            gen_s = round(time.time() - t0, 1)
            for key, exp_path in exps:
                exp = json.load(open(exp_path))
                w = exp["workload"]
                mer127 = w["kmer"] > 63
                out = {"workload": f"{w['reads']} reads x {(str(w['min_len']) + ' - ') if w.get('min_len') else ''}{w['read_len']} bp, genome {w['genome']}, err {w['err']} (scripts/synth_fastq.cpp, seed {w['seed']}), "
                                   f"K={w['kmer']}, -p {w['sets']} -a {w['a_gb']}", "reads": w["reads"], "fastq_bytes": os.path.getsize(fq), "generate_s": gen_s}
                pre = os.path.join(td, "amd_" + key)
                t0 = time.time()
                r = subprocess.run([api.binary(mer127), "pregraph", "-s", cfg, "-K", str(w["kmer"]), "-o", pre, "-p", str(w["sets"])] + (["-a", str(w["a_gb"])] if w["a_gb"] else []),
                                   capture_output=True, text=True, env=dict(os.environ, PG_HOST_VERBOSE="1", PG_STARTUP_TRACE="1"))
                wall = time.time() - t0
                out.update({"rc": r.returncode, "wall_s": wall, "reads_per_sec": w["reads"] / wall if r.returncode == 0 else None,
                            "stages_s": {m.group(1): float(m.group(2)) for m in re.finditer(r"\[cli\] ([^:]+): ([0-9.]+)s", r.stderr)}})
                m = re.search(r"Time spent on rebuilding the k-mer set layout: ([0-9.]+)s", r.stderr)
                if m:
                    out["layout_s"] = float(m.group(1))
                # how long the process waited for its device context (HIP start-up + the record pool's allocation): 0.1 s on an idle GPU,
                # seconds when another process has just released its memory -- part of wall_s and of the pass-1 stage either way
                m0, m1 = re.search(r"at ([0-9.]+)s: input files sized", r.stderr), re.search(r"at ([0-9.]+)s: device context created", r.stderr)
                if m0 and m1:
                    out["device_context_s"] = round(float(m1.group(1)) - float(m0.group(1)), 2)
                    # what it was made of (PG_STARTUP_TRACE): the steps that took more than 50 ms -- on a box whose device memory another process has
                    # just released, or has never been handed out cleared, the record pool's hipMalloc waits for the driver to clear it
                    out["device_context_steps_s"] = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"\[(?:ctx|cli)\]   ([^\n]*?):?\s+([0-9.]+) s(?! \(since)", r.stderr)
                                                     if float(m.group(2)) >= 0.05}
                m = re.search(r"reader: ([0-9.]+)s cutting \+ parsing", r.stderr)
                if m:
                    out["reader_s"] = float(m.group(1))
                out["layout_on_device"] = "k-mer set layout on the device" in r.stderr
                # the command's device arena (csrc/arena.hpp): how much physical memory it created, once, and how long that took
                m = re.search(r"arena \(device \d+\): ([0-9.]+) GB of physical memory in (\d+) piece\(s\), created in ([0-9.]+)s in all; peak in use ([0-9.]+) GB; (\d+) block\(s\) cut, (\d+) given back", r.stderr)
                if m:
                    out["arena"] = {"physical_gb": float(m.group(1)), "create_s": float(m.group(3)), "peak_in_use_gb": float(m.group(4)), "blocks_cut": int(m.group(5)), "driver_frees_before_exit": 0}
                if os.environ.get("PG_BENCH_KEEP_STDERR"):                # (GPU calls: the command's own account of itself, for profiles/)
                    open(os.path.join(os.environ["PG_BENCH_KEEP_STDERR"], key + ".stderr.txt"), "w").write(r.stderr)
                m = re.search(r"tips decided on the device: (\d+) scan\(s\), (\d+) fixed-point round\(s\), ([0-9.]+)s", r.stderr)
                if m:
                    out["tips"] = {"scans": int(m.group(1)), "rounds": int(m.group(2)), "seconds": float(m.group(3))}
                if r.returncode == 0:
                    got = md5_outputs(pre)
                    out["md5"] = got
                    out["files_identical_to_reference"] = got == exp["md5"]
                    out["reference_wall_s"] = exp.get("reference_wall_s")
                    out["expectation"] = (f"profiles/{os.path.basename(exp_path)} (oracle/_ref/SOAPdenovo-{'127' if mer127 else '63'}mer pregraph -p {w['sets']}"
                                          + (f" -a {w['a_gb']}" if w["a_gb"] else "") + " on the same file, build container)")
                    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "edge.gz"):
                        try:
                            os.remove(pre + "." + ext)
                        except OSError:
                            pass
                else:
                    out["stderr_tail"] = r.stderr[-800:]
                res[key] = out
        except Exception as e:                                       # a leg that cannot run says so; the bench line still comes out
            res[exps[0][0]] = {"skipped": f"{type(e).__name__}: {e}"}
        finally:
            import shutil
            shutil.rmtree(td, ignore_errors=True)
    return res or None


def shutil_free_gb(path):
    import shutil
    return shutil.disk_usage(path).free / 1e9


def pass_other_k(torch, api, packed, n_reads, L, P, genome, err, batch_reads, device, K=127):
    """Pass 1 at another K over resident reads, timed like the headline pass (1 warm-up, 2 timed passes, inputs resident), with the same
    conservation check on the timed result.  K = 127: configs[4]'s path, the four-word k-mers of the SOAPdenovo-127mer flavour
    (prlHashReads.c:374-378, kmer.c:532), 80 B per k-mer occurrence (40-byte node read + write) + the packed read; K <= 63: 48 B an occurrence."""
    mer127 = K > 63
    kpr = L - K + 1
    if kpr < 1:
        return None
    n_kmers = n_reads * kpr
    expected = min(n_kmers, genome + n_reads * L * err * min(K, kpr))
    log2_slots = 20
    while (1 << log2_slots) * 0.6 < expected:
        log2_slots += 1
    kc = api.KmerCounter(K, n_sets=P, mer127=mer127, log2_slots=log2_slots, device=device, engine=2)
    try:
        api._check(api.lib().pg_expect_kmers(kc.h, n_kmers), "pg_expect_kmers")
        kc.set_autogrow(False)
        wpr = (L + 31) // 32
        batches = [(lo, min(batch_reads, n_reads - lo)) for lo in range(0, n_reads, batch_reads)]
        ev = []
        hist = None
        steps = 2
        for it in range(1 + steps):
            kc.reset()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            for lo, n in batches:
                kc.count_uniform(packed[lo * wpr:], n, L, lo * kpr)
            e[1].record()
            hist, _ = kc.finalize(0, want_last_put=False)
            e[2].record()
            if it:
                ev.append(e)
        torch.cuda.synchronize()
        k1 = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        k2 = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        distinct = kc.distinct()
        cov = int((hist * np.arange(256, dtype=np.uint64)).sum())
        sat = int(hist[255])
        bytes_per_read = kpr * (80 if mer127 else 48) + (L + 3) // 4
        ok = int(hist.sum()) == distinct and (cov == n_kmers if sat == 0 else cov <= n_kmers)
        return {"workload": f"{n_reads} resident reads x {L} bp, genome {genome}, err {err}, K={K} ({'four' if mer127 else 'two'}-word k-mers, {kpr} k-mers a read), -p {P}",
                "ms_per_pass": k1 + k2, "reads_per_sec": n_reads / ((k1 + k2) * 1e-3), "k1_scatter_ms": k1, "k2_count_ms": k2, "distinct_kmers": distinct,
                "algorithmic_bytes_per_read": bytes_per_read, "roofline_frac_k2": n_reads * bytes_per_read / (k2 * 1e-3) / 1e9 / 8000.0,
                "roofline_frac_both_kernels": n_reads * bytes_per_read / ((k1 + k2) * 1e-3) / 1e9 / 8000.0,
                "conservation": {"histogram_sum_equals_distinct": int(hist.sum()) == distinct, "kmer_occurrences_in": n_kmers, "sum_of_coverage_histogram": cov,
                                 "saturated_nodes": sat, "ok": ok}}
    finally:
        kc.close()


def pass_ragged(torch, api, packed, n_reads, L, P, genome, err, batch_reads, device, K, min_len, seed, uniform_k1_ms=None):
    """Pass 1 over the SAME resident reads trimmed to lengths uniform in [min_len, L] -- every batch ragged (word_off / kmer_base index arrays,
    pg_set_read_len_bound(L)), cut by the RAGGED form of the tiled K1 (round 6; round 5 dropped such batches to a one-lane-a-read kernel).
    Timed like the headline pass (1 warm-up, 2 timed passes, inputs resident), conservation on the timed result; `k1_per_base_vs_uniform` =
    K1's time per BASE against the uniform pass's.  The reference chops reads of any length alike: prlHashReads.c:163-259,642-648."""
    mer127 = K > 63
    if min_len < K + 1 or min_len > L:
        return None
    wpr = (L + 31) // 32
    dev = packed.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    p2 = packed[: n_reads * wpr].view(n_reads, wpr)
    batches, n_kmers, n_bases, ord_base = [], 0, 0, 0
    for lo in range(0, n_reads, batch_reads):
        n = min(batch_reads, n_reads - lo)
        lens = torch.randint(min_len, L + 1, (n,), device=dev, generator=g, dtype=torch.int64)
        nw = (lens + 31) // 32
        off = torch.cumsum(nw, 0) - nw
        total = int((off[-1] + nw[-1]).item())
        ridx = torch.repeat_interleave(torch.arange(n, device=dev), nw, output_size=total)
        k = torch.arange(total, device=dev) - off[ridx]
        vals = p2[lo + ridx, k]
        valid = lens[ridx] - 32 * k
        shift = torch.clamp(64 - 2 * valid, min=0)
        mask = torch.where(valid >= 32, torch.full_like(vals, -1), -torch.bitwise_left_shift(torch.ones_like(vals), shift))
        words = torch.zeros(total + 8, dtype=torch.int64, device=dev)
        words[:total] = vals & mask
        kb = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        kb[1:] = torch.cumsum(lens - K + 1, 0)
        nk = int(kb[-1].item())
        batches.append((words, off.contiguous(), kb, n, nk, ord_base))
        ord_base += nk
        n_kmers += nk
        n_bases += int(lens.sum().item())
        del ridx, k, vals, valid, shift, mask, lens, nw
    expected = min(n_kmers, genome + n_bases * err * min(K, L - K + 1))
    log2_slots = 20
    while (1 << log2_slots) * 0.6 < expected:
        log2_slots += 1
    kc = api.KmerCounter(K, n_sets=P, mer127=mer127, log2_slots=log2_slots, device=device.index if hasattr(device, "index") else device, engine=2)
    try:
        api._check(api.lib().pg_expect_kmers(kc.h, n_kmers), "pg_expect_kmers")
        kc.set_autogrow(False)
        kc.set_read_len_bound(L)
        ev, hist, steps = [], None, 2
        for it in range(1 + steps):
            kc.reset()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            for words, off, kb, n, nk, ob in batches:
                kc.count_ragged(words, off, kb, n, nk, ord_base=ob)
            e[1].record()
            hist, _ = kc.finalize(0, want_last_put=False)
            e[2].record()
            if it:
                ev.append(e)
        torch.cuda.synchronize()
        k1 = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        k2 = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        distinct = kc.distinct()
        cov = int((hist * np.arange(256, dtype=np.uint64)).sum())
        sat = int(hist[255])
        ok = int(hist.sum()) == distinct and (cov == n_kmers if sat == 0 else cov <= n_kmers)
        out = {"workload": f"the same {n_reads} resident reads trimmed to {min_len} - {L} bp (uniform lengths, every batch ragged), K={K}, -p {P}",
               "bases": n_bases, "kmer_occurrences": n_kmers, "ms_per_pass": k1 + k2, "reads_per_sec": n_reads / ((k1 + k2) * 1e-3),
               "k1_scatter_ms": k1, "k2_count_ms": k2, "k1_ps_per_base": k1 * 1e9 / n_bases, "distinct_kmers": distinct,
               "k1_kernel": "skm_scatter_seg_kernel<..., RAGGED> (tiles sized for the longest read)",
               "conservation": {"histogram_sum_equals_distinct": int(hist.sum()) == distinct, "kmer_occurrences_in": n_kmers, "sum_of_coverage_histogram": cov,
                                "saturated_nodes": sat, "ok": ok}}
        if uniform_k1_ms:
            uni = uniform_k1_ms * 1e9 / (n_reads * L)
            out["uniform_k1_ps_per_base"] = uni
            out["k1_per_base_vs_uniform"] = out["k1_ps_per_base"] / uni
        return out
    finally:
        kc.close()


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
    ap.add_argument("--whole-reads", type=int, default=4_000_000, help="reads of the whole-command / CPU-baseline FASTQ (0 = skip both)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the whole-command run and the reference run (kernel work only)")
    ap.add_argument("--no-big", action="store_true", help="skip the 60 M-read -a 16 command (19 GB of FASTQ on local disk, ~25 s)")
    ap.add_argument("--no-extras", action="store_true", help="skip the export + sort and the PCIe-inclusive measurements")
    ap.add_argument("--no-k127", action="store_true", help="skip the K = 127 pass over the same reads (the `k127` entry of the line)")
    ap.add_argument("--ragged", action="store_true", help="the `ragged` entry even with --no-extras")
    ap.add_argument("--ragged-min-len", type=int, default=100, help="the `ragged` entry: the same reads trimmed to lengths uniform in [this, read_len] (0 = skip)")
    ap.add_argument("--exchange", default="lib", help="N > 1: lib = pg_count_reads_sharded (librccl through the C ABI), torch = torch.distributed all_to_all")
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
    # The executable's runs (and the reference's, = the CPU baseline) come FIRST, before this process has a context on the GPU: a child
    # that allocates its record pool beside a parent holding a context on the same GPU pays seconds for it (round 3's driver run:
    # device_context_s 2.5 - 4 s; on its own 0.03 s) -- that is the neighbour's cost, not the command's.  Nothing of this touches
    # the timed region below.
    commands = {}
    if world == 1 and args.engine == 2 and not args.no_cpu_baseline and args.whole_reads > 0:
        # the reference's workers each scan the whole k-mer buffer (prlHashReads.c:79-90), so its pass 1 stops scaling long before a
        # 100+-core host is used up (-p 256 was slower than -p 16 in round 1); it runs at the same -p as the executable, which also
        # makes its files comparable byte for byte
        wc, cpu = whole_command(args, args.sets)
        commands["whole_command"] = wc
        if cpu:
            commands["cpu_baseline"] = cpu
        if not args.no_big:
            big = big_command(args)
            if big:
                commands.update(big)
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

    # N > 1: the library's own communicator (librccl through the C ABI): rank 0 makes the id, torch's store carries it
    comm, exchange = None, "none"
    if world > 1 and args.engine == 2:
        exchange = args.exchange
        if (args.comm != "nccl" or args.share_gpu) and exchange == "lib":
            # RCCL refuses ranks that share a GPU (test set-ups only): the library's own sharded pass -- the same pipelined
            # pg_count_reads_sharded, the same routing, flags and appends -- over a transport that stages the exchange through the
            # host and torch.distributed's gloo all-to-all (pg_comm_create_host)
            def host_a2a(send, so, sc, recv, ro, rc):
                assert all(so[i] + sc[i] == so[i + 1] for i in range(world - 1)) and all(ro[i] + rc[i] == ro[i + 1] for i in range(world - 1))
                st_ = torch.frombuffer(send, dtype=torch.uint8)[so[0]: so[0] + sum(sc)] if sum(sc) else torch.empty(0, dtype=torch.uint8)
                rt_ = torch.frombuffer(recv, dtype=torch.uint8)[ro[0]: ro[0] + sum(rc)] if sum(rc) else torch.empty(0, dtype=torch.uint8)
                dist.all_to_all_single(rt_, st_, list(rc), list(sc))
            comm = api.Comm.host(world, rank, local, host_a2a)
            exchange = "lib-host"
        if exchange == "lib":
            # No substitute: a run asked to time the library's exchange that cannot create the library's communicator FAILS with the
            # RCCL error (round 5 fell back to torch.distributed here and carried on -- the line would then have timed torch's RCCL
            # under the library's name).  `--exchange torch` is how torch's all-to-all is asked for.  The ranks agree first, so that a
            # rank whose create failed does not leave the others hanging in their first collective.
            err = ""
            try:
                box = [api.Comm.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                comm = api.Comm.rccl(world, rank, local, box[0])
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                print(f"[bench rank {rank}] pg_comm_create failed{': ' + err if err else ' on another rank'} -- not substituting torch.distributed "
                      f"(--exchange torch asks for it)", file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
            # the rank count the LIBRARY's communicator reports (not torch's): goes into the line
            if comm.size != world or comm.rank != rank:
                print(f"[bench rank {rank}] the library's communicator reports rank {comm.rank} of {comm.size}, torch {rank} of {world}", file=sys.stderr)
                sys.exit(3)

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
    if engine == 2:
        # the input size is known up front, as call_pregraph knows it from the file sizes: the partition count and the record
        # pool follow it.  N > 1: it follows the whole job -- a partition holds what ALL ranks send to it (every rank must cut
        # with the same geometry); the export array (log2_slots) follows this rank's share
        # (N > 1, round 6: a rank STORES only the partitions it owns -- pg_expect's n_owners -- so cursors, chunk table and pool follow 1 / N of the job)
        api._check(api.lib().pg_expect(kc.h, n_kmers * world, n_reads * world, 0, world), "pg_expect")
    kc.set_autogrow(False)
    wpr = (L + 31) // 32
    batches = [(lo, min(args.batch_reads, n_reads - lo)) for lo in range(0, n_reads, args.batch_reads)]
    ord0 = rank * n_kmers
    ord0_last = ord0 + n_kmers                     # 1 + the ordinal of the last k-mer occurrence of this rank's reads
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
            elif engine == 2 and comm is not None:
                # partition engine across GPUs: owner(partition) = partition mod world; super-k-mer records travel.
                # One collective call of the library per batch: cut + route, counts, records (ncclSend/ncclRecv group), append.
                kc.count_sharded(comm, view, n, L, ord0 + lo * kpr)
            elif engine == 2:
                # the same steps spelled out with torch.distributed as the transport
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
        kc._last_hist, _ = kc.finalize(0, want_last_put=False)
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
    pipe0 = comm.pipeline_stats() if comm is not None else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    pipe1 = comm.pipeline_stats() if comm is not None else None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    distinct = kc.distinct()                       # also raises if the set overflowed
    # ---- conservation check on the TIMED result (milliseconds; every rank, summed): the coverage histogram adds up to the
    # distinct k-mers, the nodes' coverage fields add up to the k-mer occurrences that went in (exact while no node
    # saturates at 255, a lower bound otherwise), the latest put any set saw is the last k-mer of the last read
    hist_t, _ = kc._last_hist, None
    chk = kc.checksum() if engine == 2 else None
    cons_local = [int(hist_t.sum()), int((hist_t * np.arange(256, dtype=np.uint64)).sum()), int(hist_t[255]),
                  int(chk[6]) if chk is not None else -1, int(chk[7]) if chk is not None else -1, distinct]
    if world > 1:
        t = torch.tensor(cons_local, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        cons_local = [int(x) for x in t.tolist()]
        distinct = cons_local[5]
    last_timed = kc.last_put() if engine == 2 else None           # K3: a second expansion of every record (untimed)
    max_last = int(last_timed.max()) if last_timed is not None else -1
    if world > 1 and last_timed is not None:
        t = torch.tensor([max_last], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_last = int(t.item())
    total_kmers = n_kmers * world
    conservation = {
        "checked_on": "the result of the last timed step",
        "histogram_sum": cons_local[0], "distinct": distinct, "histogram_sum_equals_distinct": cons_local[0] == distinct,
        "kmer_occurrences_in": total_kmers, "sum_of_coverage_histogram": cons_local[1], "saturated_nodes": cons_local[2],
        "occurrences_conserved": (cons_local[1] == total_kmers) if cons_local[2] == 0 else (cons_local[1] <= total_kmers),
        "occurrence_check": "exact (no node saturated)" if cons_local[2] == 0 else "lower bound (saturated nodes count 255)",
        "sum_of_exported_coverage_fields": cons_local[3] if cons_local[3] >= 0 else None,
        "export_agrees_with_histogram": (cons_local[3] == cons_local[1] and cons_local[4] == cons_local[2]) if cons_local[3] >= 0 else None,
        "max_last_put": max_last if max_last >= 0 else None,
        "max_last_put_equals_last_ordinal_plus_1": (max_last == ord0_last) if max_last >= 0 and world == 1 else None,
    }
    conservation["ok"] = all(v is not False for v in conservation.values())
    digest_timed = [int(x) for x in chk[:6]] if chk is not None else None

    # ---- outside the timed steps (N = 1): the hand-over call_pregraph does after pass 1, and a PCIe-inclusive pass
    extras, st_snapshot = {}, None
    if world == 1 and engine == 2 and not args.no_extras:
        # (c) one pass with the batches coming from pinned host memory: two device buffers, copies on their own stream
        nb = min(args.batch_reads, n_reads)
        host = torch.empty(n_reads * wpr + 8, dtype=torch.int64, pin_memory=True)      # the whole packed input, page-locked
        host.copy_(packed)
        dbuf = [torch.empty(nb * wpr + 8, dtype=torch.int64, device=dev) for _ in range(2)]
        cstream = torch.cuda.Stream()
        copied = [torch.cuda.Event() for _ in range(2)]
        used = [torch.cuda.Event() for _ in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kc.reset()
        for i, (lo, n) in enumerate(batches):
            b = i & 1
            with torch.cuda.stream(cstream):
                if i >= 2:
                    cstream.wait_event(used[b])
                dbuf[b][: n * wpr + 8].copy_(host[lo * wpr: (lo + n) * wpr + 8], non_blocking=True)
                copied[b].record(cstream)
            torch.cuda.current_stream().wait_event(copied[b])
            kc.count_uniform(dbuf[b], n, L, ord0 + lo * kpr)
            used[b].record(torch.cuda.current_stream())
        kc.finalize(0, want_last_put=False)
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - t0
        # the same reads through other buffers in other batch boundaries' timing: the digest of the distinct k-mers must not move
        conservation["digest_equal_resident_vs_pcie_pass"] = [int(x) for x in kc.checksum()[:6]] == digest_timed
        conservation["ok"] = conservation["ok"] and conservation["digest_equal_resident_vs_pcie_pass"]
        extras["pcie_inclusive_ms_per_pass"] = dt_h * 1e3
        extras["pcie_inclusive_reads_per_sec"] = n_reads / dt_h
        extras["pcie_note"] = f"the packed reads ({n_reads * wpr * 8 / 1e9:.1f} GB) start in pinned host memory; batches of {nb} reads are copied on a second stream into two device buffers while the previous batch is cut"
        del host, dbuf
        # (b) per-set counts + the decision about the last put (what call_pregraph runs instead of K3)
        t0 = time.perf_counter()
        cnts = kc.set_counts()
        need = bool(api.lib().pg_host_last_put_matters(cnts.ctypes.data, P, 0, int(mer127)))
        extras["set_counts_ms"] = (time.perf_counter() - t0) * 1e3
        extras["last_put_needed"] = need
        t0 = time.perf_counter()
        kc.last_put()
        extras["last_put_kernel_ms"] = (time.perf_counter() - t0) * 1e3          # paid only when needed
        # (a) export + sort: the distinct k-mers in (set, first ordinal) order for the layout replay, as call_pregraph hands
        # them over: pg_export_take_ws (the export array itself, and the record pool as scratch) + pg_sort_records_ws.  Last, because the
        # context is spent afterwards.
        import ctypes
        st_snapshot = kc.stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ptr, got, ws, ws_bytes = ctypes.c_void_p(0), ctypes.c_uint64(0), ctypes.c_void_p(0), ctypes.c_uint64(0)
        api._check(api.lib().pg_export_take_ws(kc.h, ctypes.byref(ptr), ctypes.byref(got), ctypes.byref(ws), ctypes.byref(ws_bytes)), "pg_export_take_ws")
        sort_ok = True
        if got.value:       # sorted inside the record pool pass 1 is done with: no allocation on the way
            api._check(api.lib().pg_sort_records_ws(ptr, got.value, int(mer127), ws, ws_bytes, kc._stream()), "pg_sort_records_ws")
        torch.cuda.synchronize()
        extras["export_sort_ms"] = (time.perf_counter() - t0) * 1e3
        extras["export_sorted_on_device"] = bool(sort_ok)
        assert got.value == distinct
        torch.cuda.synchronize()
        api.hip_free(ptr)
        api.hip_free(ws)

    k127 = k31 = ragged = None
    want_ragged = world == 1 and engine == 2 and args.ragged_min_len > 0 and (not args.no_extras or args.ragged)
    want_k127 = world == 1 and engine == 2 and K <= 63 and L >= 128 and not args.no_extras and not args.no_k127
    st_final = None
    if want_ragged or want_k127:
        st_final = st_snapshot if extras else kc.stats()
        kc.close()                                              # (its pools go; the passes below make their own)
        torch.cuda.empty_cache()
    if want_ragged:
        try:
            ragged = pass_ragged(torch, api, packed, n_reads, L, P, args.genome, args.err, args.batch_reads, dev, K, args.ragged_min_len, args.seed + 77,
                                 uniform_k1_ms=sum(e0.elapsed_time(e1) for e0, e1, _ in ev) / args.steps if ev else None)
        except Exception as e:
            ragged = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if want_k127:
        try:
            k127 = pass_other_k(torch, api, packed, n_reads, L, P, args.genome, args.err, args.batch_reads, local, K=127)
        except Exception as e:
            k127 = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE.json configs[1] at its full size: 10 M x 100 bp over 4.6 Mb, err 0.005, K = 31 -- its own reads (a fraction of a second to draw)
        try:
            del packed
            torch.cuda.empty_cache()
            c1 = dict(n=10_000_000, L=100, genome=4_600_000, err=0.005, K=31)
            p1 = gen_packed_reads(torch, dev, c1["genome"], c1["n"], c1["L"], c1["err"], 20260926)
            k31 = pass_other_k(torch, api, p1, c1["n"], c1["L"], P, c1["genome"], c1["err"], args.batch_reads, local, K=c1["K"])
            if k31:
                k31["config"] = "BASELINE.json configs[1]"
            del p1
        except Exception as e:
            k31 = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        ms = dt / args.steps * 1e3
        total_reads = n_reads * world
        rec = {
            "metric": f"pregraph_pass1_reads_per_sec_K{K}", "value": total_reads / (dt / args.steps), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "distinct_kmers_per_sec": distinct / (dt / args.steps), "conservation": conservation,
            "kmer_occurrences_per_sec": n_kmers * world / (dt / args.steps),
            "config": {"workload": workload_label(n_reads, L, args.genome, args.err, K, P, world),
                       "reads_per_gpu": n_reads, "read_len": L, "K": K, "genome": args.genome, "err": args.err,
                       "distinct_kmers": distinct, "table_slots_log2": log2_slots,
                       "engine": engine,
                       "parallelism": ("single GPU, " + ("super-k-mer partitions counted in LDS" if engine == 2 else "fused extract+insert into one DRAM set"))
                       if world == 1 else (f"owner(record) = minimizer partition mod {world} (a hash of the k-mer's minimizer -- NOT north_star's high-bit key range: "
                                            f"the first base of a canonical k-mer is skewed 7:5:3:1, SURVEY.md 8e), super-k-mer records in a direct all-to-all; "
                                            f"after pass 1: owner(distinct k-mer) = reference set id mod {world}" if engine == 2
                                           else f"set-id owner, k-mer records over RCCL all-to-all x{world}"),
                       "exchange": {"lib": "pg_count_reads_sharded (librccl ncclSend/ncclRecv group through the C ABI)",
                                    "lib-host": "pg_count_reads_sharded over pg_comm_create_host (test transport: staged through the host, gloo all-to-all)",
                                    "none": None}.get(exchange, exchange)},
        }
        if comm is not None:
            rec["config"]["exchange_stats_rank0"] = comm.stats()
            d = {k: pipe1[k] - pipe0[k] for k in ("exchange_ms", "bytes_sent", "host_waits", "repeated_cuts", "rounds")}
            # (rank 0's numbers; the ranks cut equal batches of one read distribution)
            rec["exchange_ms"] = d["exchange_ms"] / args.steps           # device time of the record exchanges per step (events on the exchange stream)
            rec["bytes_sent_per_rank"] = d["bytes_sent"] / args.steps    # to the other ranks, per step
            rec["exchange"] = {"transport": comm.transport, "library_comm_ranks": comm.size, "rounds_per_step": d["rounds"] / args.steps, "host_waits_per_round": d["host_waits"] / max(d["rounds"], 1),
                               "repeated_cuts": d["repeated_cuts"], "owner_region_records": pipe1["owner_region_records"],
                               "exchange_GBps_per_rank": (d["bytes_sent"] / 1e9) / (d["exchange_ms"] / 1e3) if d["exchange_ms"] > 0 else None,
                               "exchange_over_step": d["exchange_ms"] / args.steps / ms,
                               "overlap": "round i's records travel on their own stream while batch i + 1 is cut and round i - 1 is appended (exchange.hip)"}
        if extras:
            rec["pass1_hand_over"] = extras
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
                # Partition engine: two kernels share the pass.  K1 (skm_scatter_seg_kernel) reads the packed reads and
                # writes super-k-mer records; K2 (skm_count_kernel) reads the records, counts every partition in LDS and writes
                # the distinct k-mers.  `achieved` follows the contract: SURVEY.md 8d's algorithmic bytes per read (one node
                # read + one node write per k-mer occurrence + the packed read) x the reads one launch of the dominant kernel
                # processes / its duration.  The bytes that formulation really moves (passes x record bytes) are listed too.
                st = st_final if st_final is not None else (st_snapshot if extras else kc.stats())
                rec_bytes = st["records"] * st["unit_bytes"]
                k1_bytes = n_reads * wpr * 8 + rec_bytes
                k2_bytes = rec_bytes + distinct * (kc.nw + 2) * 8
                k1_s, k2_s = sum(dur) / args.steps, sum(dur2) / args.steps
                alg_pass = n_reads * bytes_per_read
                if k1_s >= k2_s:
                    kernel, achieved = f"skm_scatter_seg_kernel<{nwk}>", alg_pass / k1_s / 1e9
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
            traffic_note = None
            if os.path.exists(tf):
                # PMC bytes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes over this command at its full 200 M reads) are kept per read in
                # profiles/pmc_traffic.json together with a hash of the kernels' sources they were taken on: counters of another library are
                # not this run's traffic (null + "stale").  FETCH_SIZE of K2 is doubled as MI355X_MICROARCH.md prescribes for 16-byte-a-lane
                # streamed reads on gfx950 (its record reads); K1's 8-byte loads and WRITE_SIZE are uncalibrated there and stay as counted.
                try:
                    tj = json.load(open(tf))
                    kn = kernel.split("<")[0]
                    sha = library_source_sha()
                    if tj.get("library_sha") != sha:
                        traffic_note = f"stale: profiles/pmc_traffic.json was taken on library {tj.get('library_sha')}, this is {sha}"
                    elif not (L == 150 and K == 63 and abs(args.genome * 300 - n_reads * L) <= 0.5 * n_reads * L):
                        traffic_note = "the counters were taken at 150 bp, K = 63, 300x: not this workload"
                    elif kn + "_bytes_per_read" in tj:
                        reads_per_launch = n_reads if kn == "skm_count_kernel" else n_reads / len(batches)
                        traffic = tj[kn + "_bytes_per_read"] * reads_per_launch
                        traffic_note = tj.get("note")
                except Exception:
                    traffic = None
            # `bound` names the roofline the contract prices this path against (HBM bytes of a hash-table
            # formulation).  The kernel itself moves 0.15x those bytes and is limited elsewhere, see `limiter`.
            limiter = ("instruction issue at four waves a SIMD: ~182 wave-level vector instructions per read (0.48 of the issue rate), SQ_WAIT_ANY 0.60 of the wave cycles -- dependent LDS "
                       "round trips between five workgroup barriers a partition (the 88 KB LDS set allows one 1024-lane workgroup per CU; a workgroup's occurrence phase ends with its longest "
                       "probe sequence), 0.97 bank-conflict cycles per LDS issue cycle on the random 8-byte accesses; HBM ~8 % of peak (profiles/r06_final_pmc_sq_200M_K63.json, "
                       "profiles/pmc_traffic.json, profiles/r06_final_k2_phase_cycles_200M_K63.txt, DESIGN.md 3)") if engine == 2 else "random-atomic rate of the DRAM-resident set"
            # what the counters say about the same launch: PMC bytes / launch time against the same peak (never `frac`)
            counter_frac = (traffic / (avg_ms * 1e-3) / 1e9 / 8000.0) if traffic else None
            if engine == 2:
                # The formulation's OWN rooflines ("a sort/partition formulation should instead count passes x record
                # bytes and say so"): (1) memory -- the bytes it cannot avoid moving (packed reads in, super-k-mer records out; records
                # back in, distinct k-mers out) at HBM peak against the time of both kernels; (2) instruction issue, its declared
                # limiter -- the wave-level vector instructions K2 issues (SQ_INSTS_VALU of a committed PMC pass, per read) against
                # what 256 CUs x 4 SIMDs issue at 2.4 GHz with four cycles a wave64 instruction.
                floor_ms = (k1_bytes + k2_bytes) / 8000e9 * 1e3
                extra["own_formulation"] = {"definition": "passes x record bytes: reads in + records out (K1), records in + distinct k-mers out (K2)",
                                            "bytes_per_step": k1_bytes + k2_bytes, "bytes_per_read": (k1_bytes + k2_bytes) / n_reads,
                                            "hbm_floor_ms": floor_ms, "frac_of_hbm_both_kernels": floor_ms / ((k1_s + k2_s) * 1e3),
                                            "k2_frac_of_hbm": (k2_bytes / 8000e9) / k2_s}
                try:
                    tj = json.load(open(tf))
                    ipr = tj.get("skm_count_kernel_valu_insts_per_read_K127" if mer127 else "skm_count_kernel_valu_insts_per_read")
                    if tj.get("library_sha") != library_source_sha():
                        ipr = None
                    if ipr and L == 150 and K in (63, 127) and abs(args.genome * 300 - n_reads * L) <= 0.5 * n_reads * L:   # (the counters were taken at this read geometry and coverage)
                        issue_peak = 256 * 4 * 2.4e9 / 4
                        extra["valu_issue_frac"] = ipr * n_reads / issue_peak / k2_s
                        extra["valu_issue_note"] = (f"{ipr:.0f} wave-level vector instructions per read (SQ_INSTS_VALU, {tj.get('sq_source', 'profiles/')}) x reads / "
                                                    f"{issue_peak / 1e9:.0f} G wave instructions/s / K2's time")
                except Exception:
                    pass
            rec["roofline"] = {"bound": "hbm", "bound_note": "hbm-equivalent (contract): SURVEY.md 8d prices the path as a hash table in HBM; this formulation "
                                                             "moves a fraction of those bytes (see traffic / hbm_counter_frac) and is limited as `limiter_bound` says",
                               "limiter_bound": "valu-issue/lds" if engine == 2 else "random-atomic rate",
                               "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                               "traffic": traffic, "traffic_note": traffic_note, "hbm_counter_frac": counter_frac, "kernel": kernel, "launches": launches, "avg_launch_ms": avg_ms,
                               "algorithmic_bytes_per_launch": per_launch, "limiter": limiter, **extra}
            if ragged is not None:
                rec["ragged"] = ragged
            if k127 is not None:
                rec["k127"] = k127
            if k31 is not None:
                rec["k31_config1"] = k31
            rec.update(commands)                                      # whole_command, cpu_baseline, whole_command_60M*: run before the pass (above)
        print(json.dumps(rec), flush=True)
    kc.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
