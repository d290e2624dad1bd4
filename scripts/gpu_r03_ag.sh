# round 3, call 35: the bench line's new fields on a small pass (the commands run as in the default line); the 200 M-read command with the reads kept on the device
mkdir -p gpurun_out/r3ag
timeout 900 python bench.py --reads 20000000 --genome 10000000 --steps 1 --warmup 1 --whole-reads 400000 > gpurun_out/r3ag/bench_small.log 2> gpurun_out/r3ag/bench_small.err; echo "bench small rc=$?"
python - <<PY
import json
l = [x for x in open("gpurun_out/r3ag/bench_small.log") if x.startswith("{")][-1]; j = json.loads(l)
print("value", round(j["value"] / 1e6, 1), "ok", j["conservation"]["ok"], "whole", {k: j["whole_command"].get(k) for k in ("wall_s", "files_identical_to_reference")})
for k in ("whole_command_60M_a16", "whole_command_60M"):
    b = j.get(k) or {}
    print("   ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "reader_s", "layout_s", "files_identical_to_reference", "stages_s")})
PY
D=/tmp/pgbig200
Bc="--reads 200000000 --out $D --keep-fastq"
timeout 1200 python scripts/big_cli_check.py $Bc --expect profiles/r03_hostreplay_200M_K63.json --tag _warm > gpurun_out/r3ag/w.log 2>&1; echo "big200 warm rc=$?"
timeout 1200 python scripts/big_cli_check.py $Bc --expect profiles/r03_hostreplay_200M_K63.json --tag _a0 > gpurun_out/r3ag/a0.log 2>&1; echo "big200 -a 0 rc=$?"
timeout 1200 python scripts/big_cli_check.py $Bc --a-gb 64 --expect profiles/r03_hostreplay_200M_K63_a64.json --tag _a64 > gpurun_out/r3ag/a64.log 2>&1; echo "big200 -a 64 rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3ag/big200; cp $D/result*.json $D/stderr*.txt gpurun_out/r3ag/big200/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3ag/big200/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "gen", j.get("generate_s"), "identical to the earlier run", j.get("identical_to_earlier_run"))
    for l in j["log"]:
        if "cli]" in l and "at " not in l: print("    ", l)
PY
