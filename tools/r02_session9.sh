#!/usr/bin/env bash
tools/gpu_session.sh \
  "tests_cons|600|python3 -m pytest tests/test_gpu_constraints.py -x -q -m gpu" \
  "timing|400|python3 tools/constraint_timing.py 2>&1 | grep -v amdgpu.ids; IKGPU_DLS_KERNEL=generic python3 tools/constraint_timing.py 2>&1 | grep demo_right" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
