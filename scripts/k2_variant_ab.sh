# timing-only A/B of -D variants of the measure library (built as soapdenovo2_amd/libx_<name>.so): K2's launch time; results may be WRONG by construction
for v in "$@"; do
  for cfg in "" "--reads 10000000 --read-len 100 --kmer 31 --genome 4600000 --err 0.005 --seed 20260926"; do
    SOAPDENOVO2_AMD_LIB=$PWD/soapdenovo2_amd/$v PG_K2_TIMERS=1 python bench.py --no-cpu-baseline --no-extras --steps 2 $cfg 2>/tmp/err_$$.txt | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', '$cfg'[:12], 'K2 ms', round(r['avg_launch_ms'],2), 'conservation', j['conservation']['ok'])"
    grep "dedupe\|flatten\|occurrences" /tmp/err_$$.txt | tail -3
  done
done
