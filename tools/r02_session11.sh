#!/usr/bin/env bash
tools/gpu_session.sh \
  "tests_generic|900|python3 -m pytest tests/test_gpu_generic.py tests/test_gpu_constraints.py tests/test_gpu_pik.py -x -q -m gpu" \
  "generic_demo|300|IKGPU_DLS_KERNEL=generic python3 bench.py --workload cassie_demo --no-cpu | cut -c1-300" \
  "generic_pik|300|IKGPU_PIK_KERNEL=generic python3 bench.py --workload cassie_demo_pik --no-cpu | cut -c1-300" \
  "pmc_generic|500|IKGPU_DLS_KERNEL=generic tools/pmc_session.sh cassie_demo pmc_generic" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
