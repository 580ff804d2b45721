#!/usr/bin/env bash
# The round's measurement matrix on one box: every bench workload (bench line + rocprofv3 kernel stats), PMC passes of the
# register-resident kernels, the self-launcher rehearsals, the stamps of the hot chain kernel, the issue-cost probe.
# Outputs under gpurun_out/; tools/r02_collect.sh copies what is to be judged into profiles/.
steps=()
for w in cassie_leg ur5 ur10 ur5_clamp ur10_clamp cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pinned cassie_demo_pinned_posture cassie_demo_pik; do
  steps+=("stats_$w|300|tools/stats_session.sh $w")
done
for w in cassie_leg ur5 cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pinned cassie_demo_pik; do
  steps+=("pmc_$w|400|tools/pmc_session.sh $w pmc_$w")
done
tools/gpu_session.sh "${steps[@]}" \
  "launcher|300|python3 bench.py --launcher --no-cpu > gpurun_out/bench_launcher_n1.json; cut -c1-250 gpurun_out/bench_launcher_n1.json" \
  "launcher_full|300|python3 bench.py --launcher --gather full --no-cpu > gpurun_out/bench_launcher_n1_gather_full.json; cut -c1-250 gpurun_out/bench_launcher_n1_gather_full.json" \
  "launcher_compact|300|python3 bench.py --launcher --gather compact --no-cpu > gpurun_out/bench_launcher_n1_gather_compact.json; cut -c1-250 gpurun_out/bench_launcher_n1_gather_compact.json" \
  "strong_n1|300|python3 bench.py --scaling strong --no-cpu > gpurun_out/bench_strong_n1.json; cut -c1-250 gpurun_out/bench_strong_n1.json" \
  "stamps|200|IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 50 uniform > gpurun_out/loop_stamps.txt; IKGPU_LIB=\$PWD/ik_amd/libikgpu_stamp.so python3 tools/loop_stamps.py 200 uniform >> gpurun_out/loop_stamps.txt; cat gpurun_out/loop_stamps.txt" \
  "sweep|200|python3 tools/iter_sweep.py > gpurun_out/iter_sweep.txt; cat gpurun_out/iter_sweep.txt" \
  "constraints|300|python3 tools/constraint_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/constraint_timing.txt; IKGPU_DLS_KERNEL=generic python3 tools/constraint_timing.py 2>&1 | grep demo_right >> gpurun_out/constraint_timing.txt; cat gpurun_out/constraint_timing.txt" \
  "probe|200|tools/issue_probe > gpurun_out/issue_probe.csv; tail -3 gpurun_out/issue_probe.csv"
