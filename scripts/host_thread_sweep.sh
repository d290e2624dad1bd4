# sweep of SOAPDENOVO2_AMD_HOST_THREADS on the config-2 whole command (run on the GPU box)
trap "rm -rf gpurun_out/thr" EXIT
python - <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
import scripts.whole_command_config as w
os.makedirs("gpurun_out/thr", exist_ok=True)
w.write_fastq_fast("gpurun_out/thr/reads.fq", w.gpu_codes(4600000, 10000000, 100, 0.005, 20260926))
from soapdenovo2_amd import synth
synth.write_config("gpurun_out/thr/lib.cfg", "gpurun_out/thr/reads.fq", 100)
PY
for T in 256 128 64 32; do
  echo "== threads $T"
  S=$(date +%s.%N)
  SOAPDENOVO2_AMD_HOST_THREADS=$T PG_HOST_VERBOSE=1 soapdenovo2_amd/bin/SOAPdenovo-63mer pregraph -s gpurun_out/thr/lib.cfg -K 31 -o gpurun_out/thr/o$T -p 8 2>&1 | grep -E "Time spent|replay set 0"
  echo "wall $(echo "$(date +%s.%N) - $S" | bc) s"
done
md5sum gpurun_out/thr/o*.preArc gpurun_out/thr/o*.vertex | awk '{print $1}' | sort | uniq -c
grep -i huge /proc/meminfo | head -3; cat /sys/kernel/mm/transparent_hugepage/enabled
