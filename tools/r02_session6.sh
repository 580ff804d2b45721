#!/usr/bin/env bash
tools/gpu_session.sh \
  "tree_ab|400|for w in cassie_full_body cassie_demo cassie_demo_posture; do for st in 1 0; do echo \"== \$w stage=\$st\"; IKGPU_TREE_STAGE_TARGETS=\$st python3 bench.py --workload \$w --no-cpu | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"].get(\"kernel_ms\"))'; done; done" \
  "tests_tree|900|python3 -m pytest tests/test_gpu_tree_posture.py tests/test_gpu_tree_fixed_base.py tests/test_gpu_generic.py tests/test_gpu_pik.py tests/test_gpu_full_size.py -x -q -m gpu" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
