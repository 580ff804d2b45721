#!/usr/bin/env bash
tools/gpu_session.sh \
  "stamps|200|IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 uniform | head -3" \
  "bench_leg|300|python3 bench.py > gpurun_out/bench_cassie_leg.json; cut -c1-330 gpurun_out/bench_cassie_leg.json" \
  "bench_fb|300|python3 bench.py --workload cassie_full_body --no-cpu | cut -c1-300" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
