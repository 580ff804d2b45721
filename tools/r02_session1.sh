#!/usr/bin/env bash
# round-2 GPU session 1: new hot chain kernel -- correctness, timing, A/B, rehearsal of the self-launcher
tools/gpu_session.sh \
  "probe_dep|120|tools/issue_probe dep > gpurun_out/issue_probe_dep.csv; tools/issue_probe 2x2 >> gpurun_out/issue_probe_dep.csv; tools/issue_probe rsq >> gpurun_out/issue_probe_dep.csv; cat gpurun_out/issue_probe_dep.csv" \
  "smoke|300|python3 -c 'import __graft_entry__ as g; g.smoke()'" \
  "bench_leg|400|python3 bench.py > gpurun_out/bench_cassie_leg.json; cut -c1-1500 gpurun_out/bench_cassie_leg.json" \
  "bench_leg_old|200|IKGPU_CHAIN_HOT=0 python3 bench.py --no-cpu > gpurun_out/bench_cassie_leg_general.json; cut -c1-400 gpurun_out/bench_cassie_leg_general.json" \
  "iter_sweep|300|python3 tools/iter_sweep.py" \
  "parity_probe|400|python3 tools/parity_probe.py cassie_leg" \
  "tests_new|900|python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_ur10.py tests/test_gpu_edges.py -x -q -m gpu -s" \
  "rehearsal|400|python3 bench.py --launcher --gather full --no-cpu > gpurun_out/bench_rehearsal.json; cut -c1-600 gpurun_out/bench_rehearsal.json" \
  "bench_ur5_clamp|300|python3 bench.py --workload ur5_clamp > gpurun_out/bench_ur5_clamp.json; cut -c1-600 gpurun_out/bench_ur5_clamp.json" \
  "tests_all|1100|python3 -m pytest tests -x -q -m gpu"
