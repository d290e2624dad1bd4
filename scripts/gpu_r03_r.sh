# round 3, call 18: the mapped reader on the GPU box (16 cores): corner-case files through the executable, the 60 M-read command both ways
mkdir -p gpurun_out/r3r
timeout 1200 python -m pytest tests -m gpu -x -q -k "corner_cases or cli_matches_reference_files or fasta_and_reference or degenerate or option_semantics" > gpurun_out/r3r/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3r/pytest.log | tail -5
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _warm > gpurun_out/r3r/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _map > gpurun_out/r3r/a.log 2>&1; echo "big60 mapped rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _copy --env SOAPDENOVO2_AMD_READER=copy > gpurun_out/r3r/b.log 2>&1; echo "big60 copy rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _map2 > gpurun_out/r3r/c.log 2>&1; echo "big60 mapped again rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3r/d.log 2>&1; echo "big60 -a 16 rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3r/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3r/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3r/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "^reader:\|cli\] parse\|cli\] layout\|cli\] pass 2\|at .*device context" gpurun_out/r3r/big*/stderr*.txt | head -40
nproc; cat /sys/fs/cgroup/cpu.max
