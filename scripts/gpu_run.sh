#!/bin/bash
# One parameterised recipe for the GPU box (replaces the 74 one-off scripts of rounds 2 - 4):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'
# Everything lands under gpurun_out/<tag>/.  Steps:
#   tests[:<pytest -k expression>]   the -m gpu suite (or a subset)
#   poison[:<pytest -k expression>]  the -m gpu suite (or a subset) with PG_ARENA_POISON=1: every arena block filled with 0xA5 as it is cut
#   smoke                            __graft_entry__.smoke()
#   bench[:<extra bench.py args>]    one bench.py line -> bench<i>.json (default arguments = the driver's run)
#   stats[:<bench args>]             rocprofv3 --kernel-trace --stats of a kernels-only bench run -> kernel_stats<i>.csv
#   pmc:<counters>[:<bench args>]    one rocprofv3 --pmc pass (comma-separated counters) of a short kernels-only run -> pmc<i>_*.csv
#   phases[:<bench args>]            K2's phase cycles from the -DPG_MEASURE library (SOAPDENOVO2_AMD_LIB=..._measure.so, PG_K2_TIMERS=1)
#   cli:<reads>:<a_gb>:<expect json>:<tag>[:ENV=V+ENV=V]   the executable on a synth_fastq file of <reads> x 150 bp (kept in /tmp across the steps of a call),
#                                    md5s against the reference's (scripts/big_cli_check.py) -> result_<tag>.json, stderr_<tag>.txt
#   cliprof:<reads>:<a_gb>:<expect json>:<tag>   the same command under `rocprofv3 --kernel-trace --stats` -> kernel_stats_cli_<tag>.csv (how
#                                    profiles/r05_kernel_stats_cli_*.csv were made)
#   sh:<command>                     anything else, logged to sh<i>.log
# A step's bench arguments use ',' for ' ' (gpurun hands one string to bash).
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
i=0
KERNELS_ONLY="--no-cpu-baseline --no-extras"
for step in "$@"; do
    i=$((i + 1))
    kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
    t0=$(date +%s)
    case $kind in
    tests)
        if [ -n "$arg" ]; then timeout 1200 python -m pytest tests -m gpu -x -q -k "$arg" > "$O/tests$i.log" 2>&1; else timeout 1200 python -m pytest tests -m gpu -x -q > "$O/tests$i.log" 2>&1; fi
        echo "[$i] tests rc=$? $(tail -1 "$O/tests$i.log" | cut -c1-160)";;
    poison)
        if [ -n "$arg" ]; then PG_ARENA_POISON=1 timeout 1400 python -m pytest tests -m gpu -x -q -k "$arg" > "$O/poison$i.log" 2>&1; else PG_ARENA_POISON=1 timeout 1400 python -m pytest tests -m gpu -x -q > "$O/poison$i.log" 2>&1; fi
        echo "[$i] poison rc=$? $(grep -E "passed|failed" "$O/poison$i.log" | tail -1 | cut -c1-160)";;
    smoke)
        timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke$i.log" 2>&1; echo "[$i] smoke rc=$? $(tail -1 "$O/smoke$i.log")";;
    bench)
        timeout 1500 python bench.py ${arg//,/ } > "$O/bench$i.json" 2> "$O/bench$i.err"; echo "[$i] bench rc=$?"
        python scripts/bench_digest.py "$O/bench$i.json";;
    stats)
        rm -rf /tmp/prof_$i; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$i -o s -- python bench.py $KERNELS_ONLY --no-k127 ${arg//,/ } > "$O/stats$i.json" 2> "$O/stats$i.err"; echo "[$i] stats rc=$?"
        f=$(find /tmp/prof_$i -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats$i.csv" && head -8 "$O/kernel_stats$i.csv" | cut -c1-200
        python scripts/bench_digest.py "$O/stats$i.json";;
    pmc)
        ctrs=${arg%%:*}; bargs=""; [ "$ctrs" != "$arg" ] && bargs=${arg#*:}
        rm -rf /tmp/pmc_$i; timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc ${ctrs//,/ } -d /tmp/pmc_$i -o p -- python bench.py $KERNELS_ONLY --no-k127 --steps 1 --warmup 0 ${bargs//,/ } > "$O/pmc$i.json" 2> "$O/pmc$i.err"; echo "[$i] pmc $ctrs rc=$?"
        f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python scripts/pmc_summary.py /tmp/pmc_$i "$O/pmc${i}_summary.json" > "$O/pmc${i}_summary.txt" 2>&1 && grep -A9 "skm_count_kernel\|skm_scatter_seg" "$O/pmc${i}_summary.txt" | head -40 | cut -c1-160;;
    phases)
        SOAPDENOVO2_AMD_LIB=$PWD/soapdenovo2_amd/libsoapdenovo2_amd_measure.so PG_K2_TIMERS=1 timeout 600 python bench.py $KERNELS_ONLY --no-k127 --steps 1 --warmup 0 ${arg//,/ } > "$O/phases$i.json" 2> "$O/phases$i.txt"; echo "[$i] phases rc=$?"
        grep "K2 phase" "$O/phases$i.txt" | tail -12;;
    cli)
        IFS=: read -r c_reads c_agb c_exp c_tag c_env <<< "$arg"
        envs=""; for kv in ${c_env//+/ }; do envs="$envs --env $kv"; done
        mkdir -p /tmp/big_$c_reads
        timeout 1500 python scripts/big_cli_check.py --reads "$c_reads" --a-gb "$c_agb" --expect "$c_exp" --out /tmp/big_$c_reads --keep-fastq --tag "_$c_tag" $envs > "$O/cli$i.log" 2>&1; echo "[$i] cli $c_tag rc=$?"
        cp /tmp/big_$c_reads/result_$c_tag.json /tmp/big_$c_reads/stderr_$c_tag.txt "$O/" 2>/dev/null
        python -c "
import json,sys
j=json.load(open('$O/result_$c_tag.json')); print('   wall', j.get('wall_s'), 'identical', j.get('identical_to_reference'))
for l in j.get('log',[]):
    if any(k in l for k in ('[cli] ','reader:','arena','waited','routed:','edges: device')): print('     ', l[:200])
";;
    cliprof)
        IFS=: read -r c_reads c_agb c_exp c_tag <<< "$arg"
        mkdir -p /tmp/big_$c_reads
        timeout 1500 python scripts/big_cli_check.py --reads "$c_reads" --a-gb "$c_agb" --expect "$c_exp" --out /tmp/big_$c_reads --keep-fastq --tag "_$c_tag" --rocprof "--kernel-trace --stats" > "$O/cliprof$i.log" 2>&1; echo "[$i] cliprof $c_tag rc=$?"
        f=$(find /tmp/big_$c_reads/prof_$c_tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats_cli_$c_tag.csv" && head -8 "$O/kernel_stats_cli_$c_tag.csv" | cut -c1-200
        cp /tmp/big_$c_reads/result_$c_tag.json "$O/" 2>/dev/null;;
    sh)
        timeout 1500 bash -c "$arg" > "$O/sh$i.log" 2>&1; echo "[$i] sh rc=$? $(tail -3 "$O/sh$i.log" | cut -c1-200)";;
    *) echo "[$i] unknown step $step";;
    esac
    echo "[$i] $kind took $(( $(date +%s) - t0 )) s"
done
