#!/usr/bin/env bash
tools/gpu_session.sh \
  "stamps|200|for m in uniform near; do IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 \$m; IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 200 \$m; done" \
  "sweep|200|python3 tools/iter_sweep.py" \
  "bench_leg|300|python3 bench.py > gpurun_out/bench_cassie_leg.json; cut -c1-300 gpurun_out/bench_cassie_leg.json" \
  "tests_full|900|python3 -m pytest tests/test_gpu_full_size.py -x -q -m gpu -s" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_full_size.py"
