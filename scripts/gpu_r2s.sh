mkdir -p gpurun_out/r2s
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -k "sharded or last_put" > gpurun_out/r2s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2s/pytest.log
grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl" gpurun_out/r2s/pytest.log | tail -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 2 --warmup 1 --reads 4000000 --genome 4000000 --comm gloo --share-gpu 2>/dev/null | tail -1 | cut -c1-400
python bench.py --reads 20000000 --genome 10000000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['pass1_hand_over'])"
