# round 3, call 6: K1 floor (no stores / no reservation), K1 direct chunks, K2 virtual-lane tiles, -a 16 with the pool reused for the image
mkdir -p gpurun_out/r3f
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3f/$tag.log 2> gpurun_out/r3f/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run k1_nostore PG_K1DBG=1
run k1_noreserve PG_K1DBG=2
run direct4 PG_DIRECT_CHUNKS=4
run direct6 PG_DIRECT_CHUNKS=6
run vt2 PG_K2_VT=2
run vt4 PG_K2_VT=4
run vt4_direct4 PG_K2_VT=4 PG_DIRECT_CHUNKS=4

PG_K2_VT=4 timeout 900 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or growth_and_batch or ragged or cli_matches_reference_files or sharded_pass1 or last_put or full_size" > gpurun_out/r3f/pytest_vt4.log 2>&1; echo "pytest vt4 rc=$?"; grep -E "passed|failed" gpurun_out/r3f/pytest_vt4.log | tail -2
timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3f/k127.log 2> gpurun_out/r3f/k127.err; echo "k127 rc=$?"
PG_K2_VT=4 timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3f/k127_vt4.log 2> gpurun_out/r3f/k127_vt4.err; echo "k127 vt4 rc=$?"
D=/tmp/pgbig60
Bc="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $Bc --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _warm > gpurun_out/r3f/w.log 2>&1; echo "big60 -a 16 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $Bc --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3f/c.log 2>&1; echo "big60 -a 16 rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3f/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3f/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
for f in sorted(glob.glob("gpurun_out/r3f/big60/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    for l in j["log"]:
        if "K6" in l or "layout" in l or "[cli] p" in l or "tips decided" in l: print("    ", l)
PY
