#!/bin/bash
# A/B of pass 2's lookup table (round 6): the 200 M-read command under rocprofv3 --kernel-trace --stats, one run an arm; prints p2_thread_kernel's and
# p2_look_build's totals.  Arms: "name:ENV=V+ENV=V" (no env = the default).  The PG_P2_* knobs (table size, answers a round) are
# measure knobs: such an arm also needs LD_PRELOAD=$PWD/soapdenovo2_amd/libsoapdenovo2_amd_measure.so (make MEASURE=1).   gpurun -- 'bash scripts/p2_look_ab.sh <tag> <arm> ...'
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
READS=${READS:-200000000}; AGB=${AGB:-40}; EXPECT=${EXPECT:-profiles/r04_ref_200M_K63_a40.json}
ROCPROF=${ROCPROF:---kernel-trace --stats}          # ROCPROF="--kernel-trace --pmc FETCH_SIZE": bytes fetched per kernel instead of times
mkdir -p /tmp/big_ab
for arm in "$@"; do
    name=${arm%%:*}; env=""; [ "$name" != "$arm" ] && env=${arm#*:}
    envs=""; for kv in ${env//+/ }; do envs="$envs --env $kv"; done
    timeout 900 python scripts/big_cli_check.py --reads "$READS" --a-gb "$AGB" --expect "$EXPECT" --out /tmp/big_ab --keep-fastq --tag "_$name" --rocprof "$ROCPROF" $envs > "$O/$name.log" 2>&1
    cp /tmp/big_ab/result_$name.json "$O/" 2>/dev/null
    if [ "${ROCPROF#*--pmc}" != "$ROCPROF" ]; then
        python scripts/pmc_summary.py /tmp/big_ab/prof_$name "$O/pmc_$name.json" > "$O/pmc_$name.txt" 2>&1
        python -c "
import json; j = json.load(open('$O/pmc_$name.json'))
for k, v in j.items():
    if 'p2_thread' in k or 'p2_look' in k: print('$name', k[:50], {c: x for c, x in v.items()})
"
        continue
    fi
    f=$(find /tmp/big_ab/prof_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$O/kernel_stats_$name.csv"
    python - "$O" "$name" <<'PY'
import csv, json, sys
o, name = sys.argv[1], sys.argv[2]
j = json.load(open(f"{o}/result_{name}.json"))
rows = {r["Name"].split("(")[0].replace("void pg::", ""): r for r in csv.DictReader(open(f"{o}/kernel_stats_{name}.csv"))}
out = {"arm": name, "identical": j.get("identical_to_reference"), "wall_s": j.get("wall_s")}
for k, r in rows.items():
    if k.startswith(("p2_thread", "p2_look", "skm_answer", "p2_route", "p2_answer")): out[k] = {"calls": int(r["Calls"]), "total_ms": round(int(r["TotalDurationNs"]) / 1e6, 1), "max_ms": round(int(r["MaxNs"]) / 1e6, 2)}
for l in j.get("log", []):
    if "lookup table" in l or "pass 2 batches" in l or "pass 2 routed:" in l: out.setdefault("log", []).append(l.strip()[:160])
print(json.dumps(out))
PY
done
