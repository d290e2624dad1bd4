#!/usr/bin/env python3
"""Whole-`pregraph`-command comparison on one of BASELINE.json's parity configurations (default configs[1]:
E. coli-scale 10 M x 100 bp, K = 31): writes the synthetic FASTQ, runs this repository's executable (MI355X) and, when
oracle/_ref travelled with the snapshot, the reference binary on the same file, compares the five output files byte for
byte (edge.gz after decompression) and prints both sets of phase timings.  Run on the GPU box:

    python scripts/whole_command_config.py --out gpurun_out/config2 [--reads 10000000 --ref-threads 16]
"""
import argparse, gzip, hashlib, json, os, re, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapdenovo2_amd import synth, api


write_fastq_fast = synth.write_fastq_fast


def gpu_codes(genome_len, n_reads, read_len, err, seed):
    return synth.gpu_reads_codes(genome_len, n_reads, read_len, err, seed)


def md5s(prefix):
    out = {}
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        out[ext] = hashlib.md5(open(f"{prefix}.{ext}", "rb").read()).hexdigest()
    h = hashlib.md5()
    with gzip.open(prefix + ".edge.gz", "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    out["edge"] = h.hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/config2")
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--seed", type=int, default=20260926)
    ap.add_argument("--kmer", type=int, default=31)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--ref-threads", type=int, default=16)
    ap.add_argument("--skip-ref", action="store_true")
    ap.add_argument("--expect", default="", help="a result.json of an earlier run on the same input: compare md5s with its ref_md5")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    fq, cfg = os.path.join(a.out, "reads.fq"), os.path.join(a.out, "lib.cfg")
    t = time.time()
    write_fastq_fast(fq, gpu_codes(a.genome, a.reads, a.read_len, a.err, a.seed))
    synth.write_config(cfg, fq, a.read_len)
    res = {"workload": f"{a.reads} x {a.read_len} bp, genome {a.genome}, err {a.err}, K={a.kmer}, -p {a.sets}", "fastq_s": time.time() - t}
    t = time.time()
    r = subprocess.run([api.binary(False), "pregraph", "-s", cfg, "-K", str(a.kmer), "-o", os.path.join(a.out, "amd"), "-p", str(a.sets)],
                       capture_output=True, text=True, env=dict(os.environ, PG_HOST_VERBOSE="1"))
    res["amd_wall_s"] = time.time() - t
    res["amd_rc"] = r.returncode
    res["amd_log"] = [l for l in r.stderr.splitlines() if "Time spent" in l or "node(s) allocated" in l or "read(s) processed" in l or "edge(s)" in l or "pre-arc" in l or "tip scan" in l or "[cli]" in l or "edges:" in l or "replay set" in l or "reader:" in l]
    if r.returncode == 0:
        res["amd_md5"] = md5s(os.path.join(a.out, "amd"))
    ref = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
    if os.path.exists(ref) and not a.skip_ref:
        t = time.time()
        r = subprocess.run([ref, "pregraph", "-s", cfg, "-K", str(a.kmer), "-o", os.path.join(a.out, "ref"), "-p", str(a.sets)], capture_output=True, text=True)
        res["ref_wall_s_same_p"] = time.time() - t
        res["ref_log"] = [l for l in r.stderr.splitlines() if "Time spent" in l or "node(s) allocated" in l]
        res["ref_md5"] = md5s(os.path.join(a.out, "ref"))
        res["identical"] = res.get("amd_md5") == res["ref_md5"]
    if a.expect and os.path.exists(a.expect):
        want = json.load(open(a.expect)).get("ref_md5")
        res["identical_to_expected_ref_md5"] = (want is not None and res.get("amd_md5") == want)
    for f in ("reads.fq",):
        os.remove(os.path.join(a.out, f))
    for pre in ("amd", "ref"):
        for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc", "edge.gz"):
            p = os.path.join(a.out, f"{pre}.{ext}")
            if os.path.exists(p):
                os.remove(p)
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(a.out, "result.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
