set -u
O=gpurun_out/r6j; mkdir -p $O
export PG_ARENA_TRACE=1 PG_HOST_VERBOSE=1
run() { # tag reads expect extra...
  tag=$1; reads=$2; exp=$3; shift 3
  python scripts/big_cli_check.py --reads $reads --expect $exp --out /tmp/leg_$reads --tag _$tag --keep-fastq "$@" > $O/leg_$tag.log 2>&1; echo "$tag rc=$?"
  cp /tmp/leg_$reads/stderr_$tag.txt $O/trace_$tag.txt; cp /tmp/leg_$reads/result_$tag.json $O/
}
run 60M_a16 60000000 profiles/r03_ref_60M_K63_a16.json --a-gb 16
run 60M_a0 60000000 profiles/r03_ref_60M_K63.json
rm -rf /tmp/leg_60000000
run 20M_K127 20000000 profiles/r04_ref_20M_K127.json --kmer 127
run 20M_ragged 20000000 profiles/r06_ref_20M_ragged_K63.json --a-gb 16
rm -rf /tmp/leg_20000000
run 200M_a40 200000000 profiles/r04_ref_200M_K63_a40.json --a-gb 40
rm -rf /tmp/leg_200000000
python scripts/big_cli_check.py --reads 10000000 --read-len 100 --genome 4600000 --err 0.005 --kmer 31 --expect profiles/r05_ref_10M_K31.json --out /tmp/leg_10M --tag _10M_k31 > $O/leg_10M_k31.log 2>&1; echo "10M rc=$?"; cp /tmp/leg_10M/stderr_10M_k31.txt $O/trace_10M_k31.txt; cp /tmp/leg_10M/result_10M_k31.json $O/
