# round 4, call 17: pass 2 with blocks of 4 / 8 k-mers looked up together, now that its probes are global loads (as flat loads each block waited for everything)
O=gpurun_out/r4q; mkdir -p $O
R=$GRAFT_REPO_ROOT
D=/tmp/pgbig60
cd /tmp && export TMPDIR=/tmp
for bl in 1 4 8; do
timeout 900 python $R/scripts/big_cli_check.py --reads 60000000 --a-gb 16 --out $D --keep-fastq --tag _bl$bl --env SOAPDENOVO2_AMD_P2_BLOCK=$bl --expect $R/profiles/r03_ref_60M_K63_a16.json --rocprof "--kernel-trace --stats --output-format csv" > $R/$O/bl$bl.log 2>&1; echo "block $bl rc=$?"
for f in $(find $D -name "*kernel_stats.csv" -newer $R/$O/bl$bl.log 2>/dev/null; find $D -name "*kernel_stats.csv" 2>/dev/null | head -1); do grep "p2_thread" $f | cut -c1-140; done | sort -u
find $D -name "*kernel_stats.csv" -exec cp {} $R/$O/kernel_stats_bl$bl.csv \; ; find $D -type d -name "*rocprof*" -exec rm -rf {} + 2>/dev/null; find $D -name "*.csv" -delete; find $D -name "*.db" -delete
done
cp $D/result*.json $R/$O/; rm -rf $D
cd $R
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/result_*.json")):
    j = json.load(open(f)); print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"), [l for l in j["log"] if "pass 2 batches" in l])
PY
