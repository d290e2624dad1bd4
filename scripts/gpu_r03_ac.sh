# round 3, call 30: the reader with threads that live as long as the file (16 granted cores of 256); the parser's own scaling on this box
mkdir -p gpurun_out/r3ac
D=/tmp/pgbig60
B="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _warm > gpurun_out/r3ac/w.log 2>&1; echo "big60 warm rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16 > gpurun_out/r3ac/a.log 2>&1; echo "big60 -a 16 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --expect profiles/r03_ref_60M_K63.json --tag _a0 > gpurun_out/r3ac/b.log 2>&1; echo "big60 rc=$?"
timeout 900 python scripts/big_cli_check.py $B --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _a16b > gpurun_out/r3ac/c.log 2>&1; echo "big60 -a 16 again rc=$?"
g++ -O3 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I soapdenovo2_amd/csrc -I include scripts/parse_bench.cpp -lz -o /tmp/parse_bench 2> gpurun_out/r3ac/parse_bench_build.err && timeout 300 /tmp/parse_bench $D/reads.fq > gpurun_out/r3ac/parse_bench.txt 2>&1; echo "parse_bench rc=$?"
rm -rf $D/reads.fq
mkdir -p gpurun_out/r3ac/big60; cp $D/result*.json $D/stderr*.txt gpurun_out/r3ac/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ac/big*/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
PY
grep -h "^reader:\|cli\] parse\|at .*device context" gpurun_out/r3ac/big*/stderr*.txt | head -20
cat gpurun_out/r3ac/parse_bench.txt
