#!/usr/bin/env bash
tools/gpu_session.sh \
  "stamps|200|for m in uniform near; do IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 \$m; done; IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 2 near"
