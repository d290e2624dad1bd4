# round 3, call 23: the sharded command with regroup / sort / layout inside each rank's record pool (no allocation behind a release of that size)
mkdir -p gpurun_out/r3w
timeout 1500 python -m pytest tests -m gpu -x -q -k "sharded or last_put_on_demand or call_pregraph_twice or linked_into" > gpurun_out/r3w/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3w/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3w/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > gpurun_out/r3w/s2.log 2>&1; echo "big60 two ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > gpurun_out/r3w/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _one > gpurun_out/r3w/a.log 2>&1; echo "big60 one rank rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3w/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3w/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3w/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "growable sets on device\|K6 on device\|cli\] \|rank .* (device" gpurun_out/r3w/big60/stderr_sh2.txt gpurun_out/r3w/big60/stderr_sh3a16.txt | head -40
