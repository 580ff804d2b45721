#!/usr/bin/env bash
tools/gpu_session.sh \
  "tree_ab|500|for v in '' _hotlog _hotlog_fold; do for w in cassie_full_body cassie_demo; do echo \"== lib\$v \$w\"; IKGPU_LIB=\$PWD/ik_amd/libikgpu\$v.so python3 bench.py --workload \$w --no-cpu | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"].get(\"kernel_ms\"))'; done; done" \
  "tests_hotlog|900|IKGPU_LIB=\$PWD/ik_amd/libikgpu_hotlog_fold.so python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_tree_posture.py tests/test_gpu_generic.py -x -q -m gpu"
