#!/usr/bin/env bash
# The round's measurement matrix on one box: every bench workload (bench line + rocprofv3 kernel stats), the self-launcher
# rehearsals, the stamps of the hot chain kernel.  Outputs under gpurun_out/; copy what is to be judged into profiles/.
steps=()
for w in cassie_leg ur5 ur10 ur5_clamp ur10_clamp cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pik; do
  steps+=("stats_$w|300|tools/stats_session.sh $w")
done
tools/gpu_session.sh "${steps[@]}" \
  "launcher|300|python3 bench.py --launcher --no-cpu > gpurun_out/bench_launcher_n1.json; cut -c1-250 gpurun_out/bench_launcher_n1.json" \
  "launcher_full|300|python3 bench.py --launcher --gather full --no-cpu > gpurun_out/bench_launcher_n1_gather_full.json; cut -c1-250 gpurun_out/bench_launcher_n1_gather_full.json" \
  "launcher_compact|300|python3 bench.py --launcher --gather compact --no-cpu > gpurun_out/bench_launcher_n1_gather_compact.json; cut -c1-250 gpurun_out/bench_launcher_n1_gather_compact.json" \
  "strong_n1|300|python3 bench.py --scaling strong --no-cpu > gpurun_out/bench_strong_n1.json; cut -c1-250 gpurun_out/bench_strong_n1.json"
