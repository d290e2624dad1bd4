# round 3, call 28: why the 4 M-read command spends 4 s in "layout + tips + edges" (verbose stages, device and host layout, twice each)
mkdir -p gpurun_out/r3aa
D=/tmp/pgbig4
B="--reads 4000000 --out $D --keep-fastq --unverified"
for t in dev1 dev2; do timeout 600 python scripts/big_cli_check.py $B --tag _$t > gpurun_out/r3aa/$t.log 2>&1; echo "$t rc=$?"; done
for t in host1; do timeout 600 python scripts/big_cli_check.py $B --tag _$t --env SOAPDENOVO2_AMD_LAYOUT=host > gpurun_out/r3aa/$t.log 2>&1; echo "$t rc=$?"; done
timeout 600 python scripts/big_cli_check.py $B --tag _dbg --env PG_RH_DEBUG=1 --env SOAPDENOVO2_AMD_LAYOUT_LANES=1 > gpurun_out/r3aa/dbg.log 2>&1; echo "dbg rc=$?"
mkdir -p gpurun_out/r3aa/big4; cp $D/result*.json $D/stderr*.txt gpurun_out/r3aa/big4/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3aa/big4/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], j.get("md5", {}).get("vertex"))
PY
grep -h "growable\|layout\|tips decided\|Time spent on\|edges:" gpurun_out/r3aa/big4/stderr_dev1.txt | head -20
grep -h "growable\|layout\|Time spent on rebuilding" gpurun_out/r3aa/big4/stderr_dev2.txt gpurun_out/r3aa/big4/stderr_host1.txt | head
grep "^rh size" gpurun_out/r3aa/big4/stderr_dbg.txt | awk '{print $3, $4, $5, $6}' | sort | uniq -c | sort -k2n | tail -30
