# round 3, call 12: growable sets laid out side by side (one stream a set)
mkdir -p gpurun_out/r3l
timeout 1200 python -m pytest tests -m gpu -x -q -k "layout_on_the_device or last_put_on_demand or sharded_matches" > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3l/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3l/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _dev > gpurun_out/r3l/a.log 2>&1; echo "big60 device layout rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _l4 --env SOAPDENOVO2_AMD_LAYOUT_LANES=4 > gpurun_out/r3l/b.log 2>&1; echo "big60 4 lanes rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _l2 --env SOAPDENOVO2_AMD_LAYOUT_LANES=2 > gpurun_out/r3l/c.log 2>&1; echo "big60 2 lanes rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > gpurun_out/r3l/s2.log 2>&1; echo "big60 two ranks rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3l/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3l/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3l/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), j.get("identical_to_earlier_run"), "gen", j.get("generate_s"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "growable sets on device\|k-mer set layout on the device\|cli\] layout" gpurun_out/r3l/big*/stderr*.txt | head -30
