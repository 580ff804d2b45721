#!/usr/bin/env bash
tools/gpu_session.sh \
  "tests_pik|600|python3 -m pytest tests/test_gpu_pik.py -x -q -m gpu" \
  "bench_pik|400|python3 bench.py --workload cassie_demo_pik > gpurun_out/bench_cassie_demo_pik.json; python3 -c 'import json; d=json.load(open(\"gpurun_out/bench_cassie_demo_pik.json\")); print(d[\"value\"], d[\"ms_per_step\"], d[\"config\"][\"kernel\"], d[\"parity_vs_cpu\"])'" \
  "bench_pik_generic|400|IKGPU_PIK_KERNEL=generic python3 bench.py --workload cassie_demo_pik --no-cpu | cut -c1-200" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
