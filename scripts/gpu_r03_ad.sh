# round 3, call 31: eight copying threads in the reader; <o>.edge.gz formatted and deflated beside pass 2
mkdir -p gpurun_out/r3ad
timeout 1500 python -m pytest tests -m gpu -x -q -k "cli_ or linked_into or call_pregraph_twice" > gpurun_out/r3ad/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3ad/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _warm > gpurun_out/r3ad/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3ad/a.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16_inline --env SOAPDENOVO2_AMD_EDGE_FILE_INLINE=1 > gpurun_out/r3ad/b.log 2>&1; echo "big60 -a 16 inline edges rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0 > gpurun_out/r3ad/c.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0b > gpurun_out/r3ad/d.log 2>&1; echo "big60 again rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3ad/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3ad/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ad/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "^reader:\|cli\] \|at .*device context\|finish: waited\|edges:" gpurun_out/r3ad/big60/stderr_a16.txt gpurun_out/r3ad/big60/stderr_a0b.txt | head -40
