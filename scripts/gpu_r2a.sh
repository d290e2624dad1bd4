set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q > gpurun_out/r2a/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest_new.log
B="python bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline"
$B > gpurun_out/r2a/b20_new.log 2>&1
PG_K2V=1 $B > gpurun_out/r2a/b20_k2old.log 2>&1
PG_K1V=1 $B > gpurun_out/r2a/b20_k1old.log 2>&1
PG_DBG=2 $B > gpurun_out/r2a/b20_dbg2.log 2>&1
PG_K2CFG=1 $B > gpurun_out/r2a/b20_cfg1.log 2>&1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/b200_new.log 2>&1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -- python $GRAFT_REPO_ROOT/bench.py --reads 20000000 --genome 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r2a/prof -name "*stats*" | head; for f in $(find gpurun_out/r2a/prof -name "*kernel_stats.csv"); do head -8 $f; done
tail -3 gpurun_out/r2a/pytest_new.log
for f in b20_new b20_k2old b20_k1old b20_cfg1 b200_new; do python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/r2a/$f.log") if x.startswith("{")][-1]; j=json.loads(l); r=j.get("roofline",{})
    print("$f", round(j["ms_per_step"],1), "k1", round(r.get("k1_scatter_ms_per_step",0),1), "k2", round(r.get("k2_count_ms_per_step",0),1), "frac", round(r.get("frac",0),3), "distinct", j["config"]["distinct_kmers"])
except Exception as e: print("$f", "ERR", e)
PY
done
grep "K2 phase" gpurun_out/r2a/b20_dbg2.log | head -12
