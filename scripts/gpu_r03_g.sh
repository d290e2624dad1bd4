# round 3, call 7: K2 virtual-lane tiles after the fix (conservation must hold), default direct chunks, all GPU tests under PG_K2_VT=4
mkdir -p gpurun_out/r3g
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3g/$tag.log 2> gpurun_out/r3g/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run vt2 PG_K2_VT=2
run vt4 PG_K2_VT=4
run nodirect PG_DIRECT_CHUNKS=0
timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3g/k127.log 2> gpurun_out/r3g/k127.err; echo "k127 rc=$?"
PG_K2_VT=4 timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3g/k127_vt4.log 2> gpurun_out/r3g/k127_vt4.err; echo "k127 vt4 rc=$?"
timeout 600 python bench.py --kmer 31 --reads 10000000 --read-len 100 --genome 4600000 --err 0.005 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3g/k31.log 2> gpurun_out/r3g/k31.err; echo "k31 rc=$?"
PG_K2_VT=4 timeout 600 python bench.py --kmer 31 --reads 10000000 --read-len 100 --genome 4600000 --err 0.005 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3g/k31_vt4.log 2> gpurun_out/r3g/k31_vt4.err; echo "k31 vt4 rc=$?"
PG_K2_VT=4 timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/pytest_vt4.log 2>&1; echo "pytest vt4 rc=$?"; grep -E "passed|failed" gpurun_out/r3g/pytest_vt4.log | tail -2
D=/tmp/pgbig60
Bc="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $Bc --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3g/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $Bc --expect profiles/r03_ref_60M_K63.json --tag _a0 > gpurun_out/r3g/a.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $Bc --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3g/c.log 2>&1; echo "big60 -a 16 rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3g/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3g/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
for f in sorted(glob.glob("gpurun_out/r3g/big60/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    for l in j["log"]:
        if "K6" in l or "layout" in l or "[cli] p" in l or "tips decided" in l or "Time spent on rem" in l: print("    ", l)
PY
