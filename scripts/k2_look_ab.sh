for v in look2 look3 measure; do
  for cfg in "--kmer 127" "" "--reads 10000000 --read-len 100 --kmer 31 --genome 4600000 --err 0.005 --seed 20260926"; do
    SOAPDENOVO2_AMD_LIB=$PWD/soapdenovo2_amd/libsoapdenovo2_amd_$v.so python bench.py --no-cpu-baseline --no-extras $cfg 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', '$cfg'[:12], 'K2 ms', round(r['avg_launch_ms'],2), 'frac', round(r['frac'],4), 'K1', round(j['ms_per_step']-r['avg_launch_ms'],2))"
  done
done
