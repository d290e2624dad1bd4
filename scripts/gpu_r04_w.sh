# round 4, the budget's last call: parity of every counting kernel (general and one-K, both flavours, both engines) and of the toggled commands with the record geometry as
# constants in the one-K kernels
O=gpurun_out/r4w; mkdir -p $O
timeout 170 python -m pytest tests -m gpu -x -q -k "count_matches_oracle or round3_switches" > $O/pytest_sub.log 2>&1; echo "pytest subset rc=$?"; tail -1 $O/pytest_sub.log
