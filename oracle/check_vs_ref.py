#!/usr/bin/env python3
"""Pin the CPU restatement (oracle/liboracle.so) against the real reference (oracle/_ref) on a matrix of
synthetic cases.  Test infrastructure; run from the repo root:  python oracle/check_vs_ref.py [--big]"""
import filecmp, gzip, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from soapdenovo2_amd import synth
from oracle_binding import run_oracle

CASES = [  # name, G, N, L, err, seed, K, [(P, D, a, mer127)]
    ("t6k_k31", 30000, 6000, 100, 0.005, 20260926, 31, [(1,0,0,0), (3,0,0,0), (7,0,0,0), (8,0,0,0), (3,1,0,0), (2,0,1,0), (8,0,0,1)]),
    ("t8k_k63", 40000, 8000, 150, 0.003, 3, 63, [(2,0,0,0), (5,0,0,0), (2,0,0,1), (8,2,0,0)]),
    ("t6k_k127", 40000, 6000, 250, 0.002, 5, 127, [(3,0,0,1), (1,0,0,1)]),
    ("t4k_k13", 3000, 4000, 60, 0.01, 11, 13, [(4,0,0,0)]),
    ("t5k_k24", 20000, 5000, 80, 0.01, 12, 24, [(8,0,0,0)]),   # even K -> 25
]
BIG = [("m100k_k31", 500000, 100000, 100, 0.005, 7, 31, [(8,0,0,0), (1,0,0,0)]),
       ("m60k_k63", 400000, 60000, 150, 0.002, 8, 63, [(8,0,0,0), (8,0,0,1)])]

def main():
    cases = CASES + (BIG if "--big" in sys.argv else [])
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for name, G, N, L, err, seed, K, runs in cases:
            cfg = synth.make_case(td, name, G, N, L, err, seed)
            codes = synth.reads_codes(G, N, L, err, seed)
            for P, D, a, m127 in runs:
                tag = f"{name}_p{P}_d{D}_a{a}_{'127' if m127 else '63'}"
                ref = os.path.join(td, "ref_" + tag); ora = os.path.join(td, "ora_" + tag)
                binary = os.path.join(ROOT, "oracle", "_ref", "SOAPdenovo-127mer" if m127 else "SOAPdenovo-63mer")
                cmd = [binary, "pregraph", "-s", cfg, "-K", str(K), "-o", ref, "-p", str(P)]
                if D: cmd += ["-d", str(D)]
                if a: cmd += ["-a", str(a)]
                subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                run_oracle(codes, K, P, ora, D=D, a_gb=a, mer127=bool(m127))
                ok = all(filecmp.cmp(f"{ora}.{e}", f"{ref}.{e}", shallow=False) for e in ("kmerFreq", "vertex", "preGraphBasic"))
                ok = ok and open(ora + ".edge", "rb").read() == gzip.open(ref + ".edge.gz", "rb").read()
                print(("OK  " if ok else "FAIL"), tag, flush=True)
                bad += not ok
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
