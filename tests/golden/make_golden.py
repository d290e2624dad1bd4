#!/usr/bin/env python3
"""Regenerates tests/golden/ by running the reference binaries built by oracle/Makefile.ref (oracle/_ref) on
synthetic inputs from soapdenovo2_amd.synth.  The reference ships no tests or vectors of its own,
so these files ARE the parity pin.  Only data is stored: input parameters (cases.json), the reference's output
files for the small cases, and md5 digests for the larger ones.

    python tests/golden/make_golden.py
    python tests/golden/make_golden.py --quirk rq_bam_odd     # add / refresh one reader corner case, everything else stays
    python tests/golden/make_golden.py --case l1500k_k31      # ... or one case
"""
import gzip, hashlib, json, os, shutil, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from soapdenovo2_amd import synth

# name: genome, reads, len, err, seed, K, runs [(P, D, a, mer127)], keep full files?
CASES = {
    # (round 6: -p beyond 8 -- 16, 37, 64, 255: the set id is an unsigned char and fixes every output byte, pregraph.c:142-220,
    #  prlHashReads.c:66-126 -- and -d together with -a)
    "t6k_k31":  dict(G=30000, N=6000, L=100, err=0.005, seed=20260926, K=31,
                     runs=[(1,0,0,0), (8,0,0,0), (7,0,0,0), (3,1,0,0), (2,0,1,0), (16,0,0,0), (255,0,0,0), (37,1,1,0)], full=[(1,0,0,0), (8,0,0,0)]),
    "t8k_k63":  dict(G=40000, N=8000, L=150, err=0.003, seed=3, K=63,
                     runs=[(2,0,0,0), (8,0,0,0), (2,0,0,1), (2,0,1,0), (2,0,1,1), (3,1,0,0), (3,2,0,1), (37,0,0,0), (64,0,0,0), (16,1,1,0)], full=[(2,0,0,0)]),
    "t6k_k127": dict(G=40000, N=6000, L=250, err=0.002, seed=5, K=127, runs=[(3,0,0,1), (3,0,1,1), (3,1,0,1), (16,0,0,1), (64,0,0,1), (255,0,0,1)], full=[]),
    "t5k_k24":  dict(G=20000, N=5000, L=80, err=0.01, seed=12, K=24, runs=[(8,0,0,0)], full=[]),
    "m100k_k31": dict(G=500000, N=100000, L=100, err=0.005, seed=7, K=31, runs=[(8,0,0,0)], full=[]),
    "m60k_k63": dict(G=400000, N=60000, L=150, err=0.002, seed=8, K=63, runs=[(8,0,0,0), (8,0,0,1)], full=[]),
    # structured genomes (synth.genome_model): two haplotypes with SNP pairs K + 2 apart -> length-1 edges, i.e. the
    # (K+1)-mer table of node2edge.c:481-542 and, at K = 127, the `char` length overflow of kmer.c:532; repeats longer than K
    "d8k_k127": dict(G=40000, N=8000, L=250, err=0.001, seed=21, K=127, model="diploid", runs=[(3,0,0,1), (8,0,0,1), (3,1,0,1)], full=[]),
    "r8k_k127": dict(G=40000, N=8000, L=250, err=0.001, seed=22, K=127, model="repeat", runs=[(3,0,0,1)], full=[]),
    "d8k_k63":  dict(G=40000, N=8000, L=150, err=0.002, seed=23, K=63, model="diploid", runs=[(5,0,0,0), (5,0,0,1), (5,1,0,0), (4,0,1,1)], full=[]),
    # -a pools under LOAD: the smallest pool is 16.7 M slots a set (prlHashReads.c:372-390), so the -a runs above sit below 0.1 % load and never
    # collide.  Here two sets of 33.5 M slots take ~17 M k-mers each (~52 %): probe clusters, kick-free first-come-first-served linear probing as
    # the device layout (dev_graph.hpp: layout_static) has to reproduce it, a cluster that wraps round the end of a table
    "l1500k_k31": dict(G=12000000, N=1500000, L=100, err=0.005, seed=77, K=31, runs=[(2,0,1,0)], full=[]),
    # trimmed reads (round 6): lengths uniform in [min_len, L] (synth.ragged_lens) -- every batch is ragged, the tiles of the super-k-mer cutter
    # are sized for the longest read and pass 2 threads the reads where pass 1 left them; prlHashReads.c:163-259,642-648 chops any read alike
    "g120k_k63": dict(G=600000, N=120000, L=150, min_len=100, err=0.002, seed=31, K=63, runs=[(8,0,0,0), (8,0,1,0)], full=[]),
    "g40k_k127": dict(G=300000, N=40000, L=250, min_len=160, err=0.002, seed=32, K=127, runs=[(5,0,0,1)], full=[]),
    "g60k_k31": dict(G=300000, N=60000, L=100, min_len=40, err=0.004, seed=33, K=31, runs=[(8,1,0,0)], full=[]),
    # long reads (round 6): max_rd_len is whatever the config says (lib.c:163), and a read of 5 - 6 k bases fills a tile of the super-k-mer cutter on its own, one of
    # 9 k is more than a tile holds (partition_kernels.hip: launch_tiled returns 1, skm_scatter_kernel takes the batch a lane a read); pass 2 walks ~5.9 k k-mers a read
    "x500_k63": dict(G=300000, N=500, L=6000, err=0.002, seed=41, K=63, runs=[(4,0,0,0), (4,0,1,0)], full=[]),
    "x400_k127": dict(G=300000, N=400, L=5000, err=0.002, seed=42, K=127, runs=[(3,0,0,1)], full=[]),
    # (6000 bases still fit a tile of ONE read; 9000 do not)
    "y300_k63": dict(G=300000, N=300, L=9000, err=0.002, seed=43, K=63, runs=[(4,0,0,0)], full=[]),
}
EXTS = ("kmerFreq", "preGraphBasic", "vertex", "edge", "preArc")


def tag(name, run):
    P, D, a, m = run
    return f"{name}_p{P}_d{D}_a{a}_{'127' if m else '63'}"


def main():
    digests = {}
    only_quirks = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--quirk"]
    only_cases = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--case"]
    old = json.load(open(os.path.join(HERE, "cases.json"))) if (only_quirks or only_cases) else None
    with tempfile.TemporaryDirectory() as td:
        for name, c in ({k: v for k, v in CASES.items() if k in only_cases} if (only_quirks or only_cases) else CASES).items():
            cfg = synth.make_case(td, name, c["G"], c["N"], c["L"], c["err"], c["seed"], model=c.get("model", "uniform"), K=c["K"], min_len=c.get("min_len", 0))
            for run in c["runs"]:
                P, D, a, m = run
                t = tag(name, run)
                pre = os.path.join(td, t)
                binary = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer" if m else "SOAPdenovo-63mer")
                cmd = [binary, "pregraph", "-s", cfg, "-K", str(c["K"]), "-o", pre, "-p", str(P)]
                if D: cmd += ["-d", str(D)]
                if a: cmd += ["-a", str(a)]
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                with open(pre + ".edge", "wb") as f:          # compare decompressed: gzip bytes depend on the zlib build
                    f.write(gzip.open(pre + ".edge.gz", "rb").read())
                digests[t] = {e: hashlib.md5(open(f"{pre}.{e}", "rb").read()).hexdigest() for e in EXTS}
                # downstream pin (P3): the reference's contig stage on the reference's pregraph files
                subprocess.run([binary, "contig", "-g", pre], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                digests[t]["contig"] = hashlib.md5(open(pre + ".contig", "rb").read()).hexdigest()
                # -R (repeat resolution by reads): two more files from pass 2, and the contig stage that consumes them
                preR = pre + "_R"
                subprocess.run([preR if x == pre else x for x in cmd] + ["-R"],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                for e in ("path", "markOnEdge"):
                    digests[t][e] = hashlib.md5(open(f"{preR}.{e}", "rb").read()).hexdigest()
                subprocess.run([binary, "contig", "-g", preR, "-R"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                digests[t]["contigR"] = hashlib.md5(open(preR + ".contig", "rb").read()).hexdigest()
                digests[t]["length1_edges"] = sum(1 for l in gzip.open(pre + ".edge.gz", "rt") if l.startswith(">length 1,"))
                if list(run) in [list(r) for r in c["full"]]:
                    for e in EXTS:
                        if e == "edge":
                            with gzip.GzipFile(os.path.join(HERE, f"{t}.edge.gz"), "wb", mtime=0) as g:
                                g.write(open(pre + ".edge", "rb").read())
                        else:
                            shutil.copy(f"{pre}.{e}", os.path.join(HERE, f"{t}.{e}"))
                print("golden", t, flush=True)
        # reader corner cases: K = 31, -p 3; also pin the reference's "read(s) processed" count
        import re
        quirks = {}
        for name in (only_quirks or ([] if only_cases else synth.QUIRK_CASES)):
            cfg = synth.make_quirk_case(td, name)
            pre = os.path.join(td, name)
            binary = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-63mer")
            out = subprocess.run([binary, "pregraph", "-s", cfg, "-K", "31", "-o", pre, "-p", "3"], check=True, capture_output=True, text=True)
            with open(pre + ".edge", "wb") as f:
                f.write(gzip.open(pre + ".edge.gz", "rb").read())
            digests[name] = {e: hashlib.md5(open(f"{pre}.{e}", "rb").read()).hexdigest() for e in EXTS}
            m = re.search(r"Time spent on hashing reads: \d+s, (\d+) read\(s\) processed", out.stderr)
            m2 = re.search(r"(\d+) node\(s\) allocated, (\d+) kmer\(s\) in reads", out.stderr)
            quirks[name] = {"reads_processed": int(m.group(1)), "nodes": int(m2.group(1)), "kmers": int(m2.group(2))}
            print("golden", name, quirks[name], flush=True)
    if old is not None:
        old["md5"].update(digests)
        old["quirks"].update(quirks)
        digests, quirks = old["md5"], old["quirks"]
    cases = {k: {kk: vv for kk, vv in v.items()} for k, v in CASES.items()}
    json.dump({"cases": cases, "md5": digests, "quirks": quirks}, open(os.path.join(HERE, "cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
