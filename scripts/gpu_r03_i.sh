# round 3, call 9: growable (-a 0) sets laid out on the device (dev_rehash.hpp) -- golden cases, then the 60 M-read command against the reference's files
mkdir -p gpurun_out/r3i
timeout 1200 python -m pytest tests -m gpu -x -q -k "layout_on_the_device or cli_matches_reference_files or sharded_matches or last_put_on_demand or corner_cases" > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3i/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3i/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _dev > gpurun_out/r3i/a.log 2>&1; echo "big60 device layout rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _host --env SOAPDENOVO2_AMD_LAYOUT=host > gpurun_out/r3i/h.log 2>&1; echo "big60 host layout rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > gpurun_out/r3i/s2.log 2>&1; echo "big60 two ranks rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3i/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3i/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3i/big60/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow ") or "at 0." in l: continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "growable sets on device\|k-mer set layout on the device\|rank .* (device" gpurun_out/r3i/big60/stderr*.txt | head -20
