# round 3, call 38: where the 200 M-read command's edge stage spends the 1.3 s it has over the -a run (vertex list?)
mkdir -p gpurun_out/r3ai
D2=/tmp/pgbig200
timeout 600 python scripts/big_cli_check.py --reads 200000000 --out $D2 --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3ai/a200.log 2>&1; echo "big200 rc=$?"
grep -h "vertex list\|vertex writer\|Time spent on constructing edges\|edges:\|tips decided\|cli\] layout" $D2/stderr_a0.txt
cp $D2/stderr_a0.txt gpurun_out/r3ai/
