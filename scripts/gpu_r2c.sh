mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_pregraph.py -m gpu -q -x -s -k "sharded or last_put or set_counts" > gpurun_out/r2c/pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest2.log
grep -v "amdgpu.ids" gpurun_out/r2c/pytest2.log | tail -40
