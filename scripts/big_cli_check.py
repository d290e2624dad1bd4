#!/usr/bin/env python3
"""Larger-than-golden check of the executable (run on the GPU box): one synthetic FASTQ through the default partition engine
and through the global-set engine (PG_ENGINE=1); both must write the same five files.  Prints timings and md5s.

    python scripts/big_cli_check.py --reads 30000000 --read-len 100 --genome 15000000 --kmer 31
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soapdenovo2_amd import synth, api
from scripts.whole_command_config import write_fastq_fast, gpu_codes, md5s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/big")
    ap.add_argument("--reads", type=int, default=30_000_000)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--genome", type=int, default=15_000_000)
    ap.add_argument("--err", type=float, default=0.004)
    ap.add_argument("--kmer", type=int, default=31)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--single", action="store_true", help="only the default engine, no comparison")
    ap.add_argument("--variant", action="append", default=[], help="extra run of the default engine with these variables, e.g. PG_NO_THP=1,PG_GROW_VERBOSE=1")
    ap.add_argument("--expect", default="", help="result.json of an earlier run with the same arguments: only the default engine runs and its md5s are compared")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    fq, cfg = os.path.join(a.out, "reads.fq"), os.path.join(a.out, "lib.cfg")
    write_fastq_fast(fq, gpu_codes(a.genome, a.reads, a.read_len, a.err, 7))
    synth.write_config(cfg, fq, a.read_len)
    os.sync()                                       # the generator's write-back is not part of the commands being timed
    res = {"workload": vars(a)}
    runs = (("partitions", {}), ("global_set", {"PG_ENGINE": "1"}))
    if a.expect or a.single:
        runs = runs[:1]
    for v in a.variant:
        runs = runs + ((v, dict(kv.split("=", 1) for kv in v.split(","))),)
    for ri, (tag, env) in enumerate(runs):
        pre = tag if tag in ("partitions", "global_set") else "variant%d" % ri
        t = time.time()
        r = subprocess.run([api.binary(False), "pregraph", "-s", cfg, "-K", str(a.kmer), "-o", os.path.join(a.out, pre), "-p", str(a.sets)],
                           capture_output=True, text=True, env={**os.environ, "PG_HOST_VERBOSE": "1", **env})
        res[tag] = {"wall_s": time.time() - t, "rc": r.returncode,
                    "log": [l for l in r.stderr.splitlines() if "[cli]" in l or "Time spent on" in l or "replay set" in l or "node(s) allocated" in l or "edge(s)" in l or "pre-arc" in l or "again" in l or l.startswith("grow ") or l.startswith("reader:")]}
        open(os.path.join(os.path.dirname(a.out.rstrip("/")) or ".", "stderr_%s.txt" % pre), "w").write(r.stderr)
        if r.returncode == 0:
            res[tag]["md5"] = md5s(os.path.join(a.out, pre))
        else:
            res[tag]["stderr_tail"] = r.stderr[-1500:]
    if a.expect and os.path.exists(a.expect):
        want = json.load(open(a.expect))["partitions"]["md5"]
        res["same_as_expected"] = res["partitions"].get("md5") == want
    elif not a.single:
        res["engines_agree"] = res["partitions"].get("md5") is not None and res["partitions"].get("md5") == res["global_set"].get("md5")
    for f in os.listdir(a.out):
        if f not in ("result.json",):
            os.remove(os.path.join(a.out, f))
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(a.out, "result.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
