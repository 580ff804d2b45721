#!/usr/bin/env bash
tools/gpu_session.sh \
  "stamps|200|IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 uniform" \
  "sweep|200|python3 tools/iter_sweep.py; python3 tools/iter_sweep.py ur5 tool0 | head -12" \
  "pmc_leg|500|tools/pmc_session.sh cassie_leg pmc_leg" \
  "pmc_ur5|500|tools/pmc_session.sh ur5 pmc_ur5" \
  "stats_leg|300|tools/stats_session.sh cassie_leg --no-cpu" \
  "stats_ur5|300|tools/stats_session.sh ur5 --no-cpu" \
  "tests_chain|900|python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_ur10.py -x -q -m gpu"
