# round 4: what is left of the budget -- as much of the rest of the GPU suite as fits (the counting kernels and the toggled commands ran in the call before)
O=gpurun_out/r4x; mkdir -p $O
timeout 160 python -u -m pytest tests -m gpu -x -q -k "not count_matches_oracle and not round3_switches and not bench_two_ranks and not waypoints" > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$? (124 = the time ran out)"; tail -3 $O/pytest_rest.log | cut -c1-200
