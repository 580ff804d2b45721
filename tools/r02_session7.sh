#!/usr/bin/env bash
tools/gpu_session.sh \
  "posture|300|python3 bench.py --workload cassie_demo_posture > gpurun_out/bench_cassie_demo_posture.json; python3 -c 'import json; d=json.load(open(\"gpurun_out/bench_cassie_demo_posture.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"parity_vs_cpu\"])'" \
  "tests_posture|600|python3 -m pytest tests/test_gpu_tree_posture.py tests/test_gpu_tree_fixed_base.py -x -q -m gpu" \
  "pmc_posture|500|tools/pmc_session.sh cassie_demo_posture pmc_posture"
