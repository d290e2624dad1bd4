# round 3, call 32: the 200 M-read command (configs[2] at full size, 63 GB of FASTQ) with the round's reader and layouts, against round 3's earlier runs of the same input
mkdir -p gpurun_out/r3ae
df -h /tmp | tail -1; free -g | head -2
D=/tmp/pgbig200
Bc="--reads 200000000 --out $D --keep-fastq"
timeout 1200 python scripts/big_cli_check.py $Bc --a-gb 64 --expect profiles/r03_hostreplay_200M_K63_a64.json --tag _a64 > gpurun_out/r3ae/a64.log 2>&1; echo "big200 -a 64 rc=$?"
timeout 1200 python scripts/big_cli_check.py $Bc --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3ae/a0.log 2>&1; echo "big200 -a 0 rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3ae/big200; cp $D/result*.json $D/stderr*.txt gpurun_out/r3ae/big200/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ae/big200/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "gen", j.get("generate_s"), "identical to the earlier run", j.get("identical_to_earlier_run"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow ") or "lane" in l: continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
