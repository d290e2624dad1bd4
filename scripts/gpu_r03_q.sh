# round 3, call 17: the list of clusters from the flags by a prefix sum in the big rounds
mkdir -p gpurun_out/r3q
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3q/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _dbg --env SOAPDENOVO2_AMD_LAYOUT_LANES=1 --env PG_RH_DEBUG=1 > gpurun_out/r3q/a.log 2>&1; echo "big60 1 lane debug rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _l1 --env SOAPDENOVO2_AMD_LAYOUT_LANES=1 > gpurun_out/r3q/b.log 2>&1; echo "big60 1 lane rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _l8 > gpurun_out/r3q/c.log 2>&1; echo "big60 8 lanes rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3q/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3q/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3q/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "growable sets on device\|growable layout, lane\|cli\] layout" gpurun_out/r3q/big*/stderr*.txt | head -40
grep "^rh size" gpurun_out/r3q/big60/stderr_dbg.txt | head -150 | tail -75
