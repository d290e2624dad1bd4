# round 3, call 33: the reads of pass 1 stay on the device for pass 2 (one-length inputs, no -R): golden cases, corner cases (the move to the host store), 60 M reads
mkdir -p gpurun_out/r3af
timeout 1500 python -m pytest tests -m gpu -x -q -k "cli_ or device_pass2 or linked_into or call_pregraph_twice" > gpurun_out/r3af/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3af/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _warm > gpurun_out/r3af/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3af/a.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16_host --env SOAPDENOVO2_AMD_KEEP_ON_HOST=1 > gpurun_out/r3af/b.log 2>&1; echo "big60 -a 16 kept on the host rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0 > gpurun_out/r3af/c.log 2>&1; echo "big60 rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3af/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3af/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3af/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "^reader:\|cli\] \|at .*device context\|Time spent on threading" gpurun_out/r3af/big60/stderr_a16.txt gpurun_out/r3af/big60/stderr_a16_host.txt | head -40
