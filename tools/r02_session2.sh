#!/usr/bin/env bash
# round-2 GPU session 2: PMC counters of the hot chain kernel, A/B of the register-pinning variants, full GPU test suite
tools/gpu_session.sh \
  "ab_pin|200|for v in '' pin22; do lib=ik_amd/libikgpu\${v:+_\$v}.so; echo \"== \$lib\"; IKGPU_LIB=\$PWD/\$lib python3 tools/iter_sweep.py | grep '^65536,\\(50\\|200\\)\\|^262144'; done" \
  "pmc_leg|500|tools/pmc_session.sh cassie_leg pmc_leg" \
  "stats_leg|300|tools/stats_session.sh cassie_leg --no-cpu" \
  "tests_all|1000|python3 -m pytest tests -x -q -m gpu"
