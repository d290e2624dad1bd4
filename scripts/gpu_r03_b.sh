# round 3, call 2: tips decided on the device + the vertex list from the device; -a without any host copy of the sets
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3b/pytest.log | tail -3
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3b/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _devtips > gpurun_out/r3b/a.log 2>&1; echo "big60 device tips rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _replay --env SOAPDENOVO2_AMD_TIPS=replay > gpurun_out/r3b/b.log 2>&1; echo "big60 replay tips rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3b/c.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16trace --rocprof "--kernel-trace --stats" > gpurun_out/r3b/d.log 2>&1; echo "big60 -a 16 trace rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3b/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3b/big60/ 2>/dev/null
for f in $(find $D/prof_a16trace -name "*kernel_stats.csv"); do cp $f gpurun_out/r3b/kernel_stats_60M_a16.csv; done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow ") or "at 0." in l: continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
head -12 gpurun_out/r3b/kernel_stats_60M_a16.csv | cut -c1-200
