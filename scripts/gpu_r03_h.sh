# round 3, call 8: K2 tiles at the restored partition count, the 200 M-read command (configs[2] at full size), the default bench line
mkdir -p gpurun_out/r3h
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3h/$tag.log 2> gpurun_out/r3h/$tag.err; echo "$tag rc=$?"; }
run base PG_NONE=1
run vt4 PG_K2_VT=4
run vt2 PG_K2_VT=2
timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3h/k127.log 2> gpurun_out/r3h/k127.err; echo "k127 rc=$?"
PG_K2_VT=4 timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3h/k127_vt4.log 2> gpurun_out/r3h/k127_vt4.err; echo "k127 vt4 rc=$?"
df -h /tmp | tail -1; free -g | head -2
D=/tmp/pgbig200
Bc="--reads 200000000 --out $D --keep-fastq --unverified"
timeout 1200 python scripts/big_cli_check.py $Bc --a-gb 64 --tag _a64 > gpurun_out/r3h/a64.log 2>&1; echo "big200 -a 64 rc=$?"
timeout 1200 python scripts/big_cli_check.py $Bc --tag _a0 > gpurun_out/r3h/a0.log 2>&1; echo "big200 -a 0 rc=$?"
rm -f $D/reads.fq
mkdir -p gpurun_out/r3h/big200; cp $D/result*.json $D/stderr*.txt gpurun_out/r3h/big200/ 2>/dev/null
timeout 1500 python bench.py > gpurun_out/r3h/bench_default.log 2> gpurun_out/r3h/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"], "parts", r.get("partitions"))
        if "whole_command" in j:
            w = j["whole_command"]; print("   whole", {k: w.get(k) for k in ("reads", "wall_s", "reference_wall_s", "files_identical_to_reference")})
            print("   big", j.get("whole_command_60M_a16"))
            print("   cpu", j.get("cpu_baseline"))
    except Exception as e:
        pass
for f in sorted(glob.glob("gpurun_out/r3h/big200/result*.json")):
    j = json.load(open(f))
    print(f.split("/")[-1], "rc", j["rc"], "wall", j["wall_s"], "gen", j.get("generate_s"), j.get("md5"))
    for l in j["log"]:
        if l.startswith("replay set") or l.startswith("grow "): continue
        print("    ", l)
    if j["rc"]: print(j.get("stderr_tail"))
PY
tail -3 gpurun_out/r3h/bench_default.err
