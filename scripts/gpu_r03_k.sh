# round 3, call 11: the growable layout with change / cluster lists -- golden cases, the 60 M-read command against the reference, the 200 M-read
# command against round 3's host-replay run of the same input
mkdir -p gpurun_out/r3k
timeout 1200 python -m pytest tests -m gpu -x -q -k "layout_on_the_device or last_put_on_demand or sharded_matches" > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3k/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3k/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _dev > gpurun_out/r3k/a.log 2>&1; echo "big60 device layout rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3k/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3k/big60/ 2>/dev/null
D=/tmp/pgbig200
timeout 1500 python scripts/big_cli_check.py --reads 200000000 --out $D --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3k/b200.log 2>&1; echo "big200 rc=$?"
mkdir -p gpurun_out/r3k/big200; cp $D/result*.json $D/stderr*.txt gpurun_out/r3k/big200/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3k/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), j.get("identical_to_earlier_run"), "gen", j.get("generate_s"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow ") or "at 0." in l: continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "growable sets on device\|k-mer set layout on the device" gpurun_out/r3k/big*/stderr*.txt | head -20
