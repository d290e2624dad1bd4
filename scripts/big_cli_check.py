#!/usr/bin/env python3
"""Larger-than-golden check of the executable: one synthetic FASTQ (scripts/synth_fastq.cpp: bytes that depend on the arguments
only, so the same file can be made wherever the reference binary has the memory and the time to run) through
`SOAPdenovo-63mer|127mer pregraph`, md5s compared with the reference's.

    # where the reference fits (this takes the reference tens of minutes; no GPU needed):
    python scripts/big_cli_check.py --reference --reads 60000000 --out /tmp/big60 --save profiles/r03_ref_60M_K63.json
    # on the GPU box:
    python scripts/big_cli_check.py --reads 60000000 --expect profiles/r03_ref_60M_K63.json --out gpurun_out/big60

Without --expect the run is refused: a big run whose files are compared with nothing proves nothing (--unverified says so
explicitly and marks the result).  Exit code 1 when the md5s differ or the command fails.
"""
import argparse, gzip, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GEN = os.path.join(ROOT, "soapdenovo2_amd", "bin", "synth_fastq")
KEEP = ("sharded:", "[cli]", "finish: waited", "pass 2 routed", "pass 2 direct", "graph lane", "arena (", "Time spent on", "replay set", "node(s) allocated", "edge(s)", "pre-arc", "again", "reader:", "tip scan", "tips decided", "edges:", "layout", "vertex")


def md5s(prefix):
    out = {}
    for ext in ("kmerFreq", "preGraphBasic", "vertex", "preArc"):
        h = hashlib.md5()
        with open(f"{prefix}.{ext}", "rb") as f:
            for chunk in iter(lambda: f.read(1 << 24), b""):
                h.update(chunk)
        out[ext] = h.hexdigest()
    h = hashlib.md5()
    with gzip.open(prefix + ".edge.gz", "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    out["edge"] = h.hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/big")
    ap.add_argument("--reads", type=int, default=30_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=100_000_000)
    ap.add_argument("--err", type=float, default=0.001)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--kmer", type=int, default=63)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--a-gb", type=int, default=0, help="the reference's -a (0 = growable sets)")
    ap.add_argument("--min-len", type=int, default=0, help="ragged reads: lengths uniform in [min_len, read_len] (synth_fastq's min_len; 0 = one length)")
    ap.add_argument("--reference", action="store_true", help="run oracle/_ref instead of the executable and --save its md5s")
    ap.add_argument("--save", default="", help="with --reference: where the expectation goes")
    ap.add_argument("--expect", default="", help="JSON written by --reference --save for the same arguments")
    ap.add_argument("--unverified", action="store_true", help="run without an expectation (the result says so)")
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for the command (repeatable)")
    ap.add_argument("--keep-fastq", action="store_true")
    ap.add_argument("--tag", default="", help="suffix of result.json / stderr.txt (several runs into one --out)")
    ap.add_argument("--rocprof", default="", help="run the command under rocprofv3 with these arguments (e.g. '--kernel-trace --stats'), output next to result.json")
    a = ap.parse_args()
    key = {k: getattr(a, k) for k in ("reads", "read_len", "genome", "err", "seed", "kmer", "sets", "a_gb")}
    if a.min_len:
        key["min_len"] = a.min_len                                 # (absent from the expectation files of one-length runs)
    want, want_kind = None, None
    if not a.reference:
        if a.expect:
            e = json.load(open(a.expect))
            if not a.min_len and e.get("workload", {}).get("min_len"):      # (the expectation says how the reads were trimmed)
                a.min_len = e["workload"]["min_len"]
                key["min_len"] = a.min_len
            if e.get("workload") != key or not e.get("md5"):
                sys.exit(f"{a.expect} holds no md5s for these arguments: {e.get('workload')} vs {key}")
            want = e["md5"]
            want_kind = e.get("kind", "reference")
        elif not a.unverified:
            sys.exit("no --expect: refusing a run that is compared with nothing (--unverified to insist)")
    os.makedirs(a.out, exist_ok=True)
    fq, cfg = os.path.abspath(os.path.join(a.out, "reads.fq")), os.path.join(a.out, "lib.cfg")
    t = time.time()
    gen_args = [str(a.genome), str(a.reads), str(a.read_len), str(a.err), str(a.seed), str(a.min_len)]
    made_with = open(fq + ".args").read().split() if os.path.exists(fq + ".args") else None
    if not (a.keep_fastq and os.path.exists(fq) and (made_with is None or made_with == gen_args)):      # (a kept file is reused only for the arguments it was made with)
        subprocess.check_call([GEN, fq, str(a.genome), str(a.reads), str(a.read_len), str(a.err), str(a.seed)] + ([str(os.cpu_count() or 8), str(a.min_len)] if a.min_len else []))
        open(fq + ".args", "w").write(" ".join(gen_args))
    open(cfg, "w").write(f"max_rd_len={a.read_len}\n[LIB]\navg_ins=200\nreverse_seq=0\nasm_flags=3\nrank=1\nq={fq}\n")
    os.sync()
    res = {"workload": key, "fastq_bytes": os.path.getsize(fq), "generate_s": round(time.time() - t, 1)}
    mer127 = a.kmer > 63
    if a.reference:
        binary = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer" if mer127 else "SOAPdenovo-63mer")
    else:
        binary = os.path.join(ROOT, "soapdenovo2_amd", "bin", "SOAPdenovo-127mer" if mer127 else "SOAPdenovo-63mer")
    pre = os.path.join(a.out, "ref" if a.reference else "amd")
    cmd = [binary, "pregraph", "-s", cfg, "-K", str(a.kmer), "-o", pre, "-p", str(a.sets)] + (["-a", str(a.a_gb)] if a.a_gb else [])
    env = dict(os.environ, PG_HOST_VERBOSE="1")
    env.update(dict(kv.split("=", 1) for kv in a.env))
    cwd = None
    if a.rocprof:
        prof_dir = os.path.abspath(os.path.join(a.out, "prof" + a.tag))
        cmd = ["rocprofv3"] + a.rocprof.split() + ["--output-format", "csv", "-d", prof_dir, "--"] + [os.path.abspath(c) if os.path.exists(c) else c for c in cmd]
        cmd[cmd.index("-o") + 1] = os.path.abspath(pre)
        env["TMPDIR"] = "/tmp"
        cwd = "/tmp"
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=cwd)
    res.update({"command": " ".join(cmd[1:]), "binary": os.path.relpath(binary, ROOT), "wall_s": round(time.time() - t, 2), "rc": r.returncode,
                "log": [l for l in r.stderr.splitlines() if any(k in l for k in KEEP)]})
    open(os.path.join(a.out, f"stderr{a.tag}.txt"), "w").write(r.stderr)
    ok = r.returncode == 0
    if ok:
        res["md5"] = md5s(pre)
        if want is not None:
            res["expect"] = a.expect
            # an expectation made by the reference itself, or ("kind": "self") by an earlier run of this executable
            verdict = "identical_to_reference" if want_kind == "reference" else "identical_to_earlier_run"
            res[verdict] = res["md5"] == want
            ok = res[verdict]
        elif not a.reference:
            res["unverified"] = True
    else:
        res["stderr_tail"] = r.stderr[-1500:]
    for f in os.listdir(a.out):
        full = os.path.join(a.out, f)
        if os.path.isfile(full) and not f.startswith(("result", "stderr")) and not (a.keep_fastq and f in ("reads.fq", "reads.fq.args", "lib.cfg")):
            os.remove(full)
    print(json.dumps(res, indent=1))
    json.dump(res, open(os.path.join(a.out, f"result{a.tag}.json"), "w"), indent=1)
    if a.reference and a.save and ok:
        json.dump({"workload": key, "md5": res["md5"], "reference_wall_s": res["wall_s"], "log": res["log"],
                   "made_by": "scripts/big_cli_check.py --reference (oracle/_ref, built from /root/reference by oracle/Makefile.ref)"}, open(a.save, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
