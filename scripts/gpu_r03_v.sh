# round 3, call 22: K1 tile sizes below 24 reads; K = 127 and 100 bp reads with small tiles
mkdir -p gpurun_out/r3v
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > gpurun_out/r3v/$tag.log 2> gpurun_out/r3v/$tag.err; echo "$tag rc=$?"; }
run r12 PG_K1_R=12
run r16 PG_K1_R=16
run r20 PG_K1_R=20
run r24 PG_K1_R=24
for r in 0 16 24 32 48; do
PG_K1_R=$r timeout 600 python bench.py --kmer 127 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3v/k127_r$r.log 2> gpurun_out/r3v/k127_r$r.err; echo "k127 r$r rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/r3v/*.log")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("/")[-1][:-4].ljust(18), "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "both", round(r["pass1_both_kernels_frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        pass
PY
