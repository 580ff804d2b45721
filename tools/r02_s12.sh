#!/usr/bin/env bash
tools/gpu_session.sh \
  "stamps|200|IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 uniform" \
  "sweep|200|python3 tools/iter_sweep.py | head -12; python3 tools/iter_sweep.py ur5 tool0 | head -10" \
  "bench_leg|300|python3 bench.py > gpurun_out/bench_cassie_leg.json; cut -c1-330 gpurun_out/bench_cassie_leg.json" \
  "tests_chain|900|python3 -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_ur10.py -x -q -m gpu"
