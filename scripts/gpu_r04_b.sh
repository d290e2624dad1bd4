# round 4, call 2: the GPU suite on the lanes (graph stages on all ranks), the pipelined sharded pass 1 and the 2-process gloo bench; K2 A/B of the early
# tile request and the persistent-workgroup count on the new kernel; phase timers and SQ counters of the new K2; the sharded 60 M-read command with and
# without overlap; the default bench line
O=gpurun_out/r4b; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $O/pytest.log | tail -8
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { tag=$1; shift; env "$@" timeout 600 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench $tag rc=$?"; }
run base PG_NOP=1
run opt3 PG_K2_OPT=3
run opt3_wg1 PG_K2_OPT=3 PG_K2_WG_PER_CU=1
run opt3_wg3 PG_K2_OPT=3 PG_K2_WG_PER_CU=3
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --kmer 127"
run k127_base PG_NOP=1
run k127_opt3 PG_K2_OPT=3
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        l = [x for x in open(f) if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
        print(f.split("bench_")[1], "pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
for k in 63 127; do
PG_DBG=2 timeout 300 python bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep "K2 phase" | tail -12 > $O/k2_phase_cycles_20M_k$k.txt
done
cat $O/k2_phase_cycles_20M_k63.txt
cd /tmp && export TMPDIR=/tmp
for k in 63 127; do
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/pmc_sq_k$k -- python $R/bench.py --kmer $k --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/pmc_sq_k$k.log 2>&1
done
cd $R
for k in 63 127; do python scripts/pmc_summary.py $O/pmc_sq_k$k $O/pmc_sq_k$k.json > $O/pmc_sq_k$k.txt 2>&1; done
python - <<PY
import json
for k in (63, 127):
    try:
        j = json.load(open("$O/pmc_sq_k%d.json" % k))
        for name, v in j.items():
            if "skm_count" in name or "skm_scatter_seg" in name: print(k, name[:60], {a: round(b) for a, b in v.items()})
    except Exception as e: print("pmc", k, "ERR", e)
PY
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
# the sharded command at 60 M reads: three ranks with -a 16, two with growable sets; the same without overlap
D=/tmp/pgbig60
C="--reads 60000000 --out $D --keep-fastq"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _one > $O/one.log 2>&1; echo "big60 one rank rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16 --env SOAPDENOVO2_AMD_DEVICES=0,0,0 > $O/s3.log 2>&1; echo "big60 -a 16 three ranks rc=$?"
timeout 900 python scripts/big_cli_check.py $C --a-gb 16 --expect profiles/r03_ref_60M_K63_a16.json --tag _sh3a16_serial --env SOAPDENOVO2_AMD_DEVICES=0,0,0 --env PG_PIPE_SERIAL=1 > $O/s3s.log 2>&1; echo "big60 -a 16 three ranks, no overlap rc=$?"
timeout 900 python scripts/big_cli_check.py $C --expect profiles/r03_ref_60M_K63.json --tag _sh2 --env SOAPDENOVO2_AMD_DEVICES=0,0 > $O/s2.log 2>&1; echo "big60 two ranks rc=$?"
rm -rf $D/reads.fq
mkdir -p $O/big60; cp $D/result*.json $D/stderr*.txt $O/big60/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/big60/result*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    if j["rc"]: print(j.get("stderr_tail"))
    print("   ", [l for l in j["log"] if "[cli]" in l and "rank" not in l][-7:])
PY
grep -h "graph lane\|records exchanged\|exchange:" $O/big60/stderr_sh3a16.txt $O/big60/stderr_sh2.txt | head -20
timeout 1500 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench default rc=$?"
python - <<PY
import json
try:
    l = [x for x in open("$O/bench_default.log") if x.startswith("{")][-1]; j = json.loads(l); r = j["roofline"]
    print("default: pass", round(j["ms_per_step"], 1), "k1", round(r["k1_scatter_ms_per_step"], 1), "k2", round(r["k2_count_ms_per_step"], 1), "frac", round(r["frac"], 3), "ok", j["conservation"]["ok"])
    print("  own", r.get("own_formulation"), "valu", r.get("valu_issue_frac"))
    print("  k127", j.get("k127"))
    for k in ("whole_command", "whole_command_60M_a16", "whole_command_60M", "whole_command_k127_20M", "whole_command_200M_a40"):
        b = j.get(k) or {}
        print("  ", k, {q: b.get(q) for q in ("wall_s", "device_context_s", "files_identical_to_reference", "reference_wall_s", "skipped", "rc")})
    print("  cpu", j.get("cpu_baseline"))
except Exception as e:
    print("default bench ERR", e)
PY
tail -3 $O/bench_default.err
