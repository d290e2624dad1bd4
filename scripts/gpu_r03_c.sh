# round 3, call 3: graph kernels on per-set base pointers; the sharded run (regroup by set owner, per-rank layout) with ranks on one GPU
mkdir -p gpurun_out/r3c
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3c/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3c/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _one > gpurun_out/r3c/a.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3c/c.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > gpurun_out/r3c/s2.log 2>&1; echo "big60 two ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > gpurun_out/r3c/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3c/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3c/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c/big60/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow ") or "at 0." in l: continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "K6 on device\|rank .* (device" gpurun_out/r3c/big60/stderr*.txt | head -20
