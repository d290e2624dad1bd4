# round 4, call 8: where a command's slow first second comes from.  scripts/startup_probe.hip (HIP start-up, 4 x 20 GB device
# allocations, page-locked buffers, each timed) on a fresh box, right behind 63 GB written through the page cache, after sync,
# after the command has read the file; the command itself three times with its marks.
O=gpurun_out/r4h; mkdir -p $O
P=soapdenovo2_amd/bin/startup_probe
mem() { grep -E "MemTotal|MemFree|MemAvailable|^Cached|Dirty|Writeback:" /proc/meminfo | tr -s ' ' | tr '\n' ';'; echo; }
{
echo "== box"; nproc; df -h /tmp | tail -1; mem
echo "== probe, fresh box"; $P 20; echo "== probe again"; $P 20
D=/tmp/pgbig200; mkdir -p $D
echo "== generating"; /usr/bin/time -f "generator %e s" soapdenovo2_amd/bin/synth_fastq $D/reads.fq 100000000 200000000 150 0.001 7; ls -la $D/reads.fq; mem
echo "== probe, right behind the generator"; $P 20; mem
echo "== sync"; /usr/bin/time -f "sync %e s" sync; mem
echo "== probe, after sync"; $P 20
} > $O/probe.log 2>&1
C="--reads 200000000 --a-gb 40 --out $D --keep-fastq --expect profiles/r04_ref_200M_K63_a40.json"
for t in 1 2 3; do timeout 600 python scripts/big_cli_check.py $C --tag _$t > $O/run$t.log 2>&1; echo "run $t rc=$?"; done
{ echo "== probe, after three commands"; mem; $P 20; } >> $O/probe.log 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("/tmp/pgbig200/result_*.json")):
    j = json.load(open(f))
    print(f, "rc", j["rc"], "wall", j["wall_s"], "identical", j.get("identical_to_reference"))
    print("   ", [l for l in j["log"] if "[cli]   at" in l or "pass 1" in l][:8])
PY
mkdir -p $O/big200; cp $D/result*.json $O/big200/; rm -rf $D
cat $O/probe.log
