# round 3, call 24: the AVX2 FASTQ record on the GPU box; reader corner cases through the executable
mkdir -p gpurun_out/r3x
timeout 1200 python -m pytest tests -m gpu -x -q -k "corner_cases or cli_matches_reference_files or fasta_and_reference or degenerate" > gpurun_out/r3x/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3x/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3x/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _simd > gpurun_out/r3x/a.log 2>&1; echo "big60 simd rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _scalar --env SOAPDENOVO2_AMD_PARSE_SIMD=0 > gpurun_out/r3x/b.log 2>&1; echo "big60 scalar rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3x/c.log 2>&1; echo "big60 -a 16 rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3x/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3x/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3x/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "^reader:\|cli\] parse\|cli\] layout\|cli\] pass 2\|at .*device context" gpurun_out/r3x/big*/stderr*.txt | head -40
