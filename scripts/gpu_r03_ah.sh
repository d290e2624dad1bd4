# round 3, call 37: the graph stages keep the device blocks they release and hand them out again (backend_hip.hpp: devcache): the whole GPU suite, 60 M and 200 M reads
mkdir -p gpurun_out/r3ah
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3ah/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3ah/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3ah/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0 > gpurun_out/r3ah/a.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0_plain --env SOAPDENOVO2_AMD_DEVICE_CACHE=0 > gpurun_out/r3ah/b.log 2>&1; echo "big60 plain allocations rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3ah/c.log 2>&1; echo "big60 -a 16 rc=$?"
rm -rf $D/reads.fq
D2=/tmp/pgbig200
Bc="--reads 200000000 --out $D2 --keep-fastq"
timeout 1200 python scripts/big_cli_check.py $Bc --expect profiles/r03_hostreplay_200M_K63.json --tag _warm > gpurun_out/r3ah/w200.log 2>&1; echo "big200 warm rc=$?"
timeout 1200 python scripts/big_cli_check.py $Bc --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3ah/a200.log 2>&1; echo "big200 rc=$?"
rm -f $D2/reads.fq
mkdir -p gpurun_out/r3ah/big; cp $D/result*.json gpurun_out/r3ah/big/ 2>/dev/null; for f in $D2/result*.json; do cp $f gpurun_out/r3ah/big/200_$(basename $f); done; cp $D/stderr_a0.txt gpurun_out/r3ah/big/stderr60_a0.txt; cp $D2/stderr_a0.txt gpurun_out/r3ah/big/stderr200_a0.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ah/big/*result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), j.get("identical_to_earlier_run"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "cli\] \|Time spent on constructing edges\|at .*device context" gpurun_out/r3ah/big/stderr200_a0.txt gpurun_out/r3ah/big/stderr60_a0.txt | head -24
