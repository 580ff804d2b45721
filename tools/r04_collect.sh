#!/usr/bin/env bash
# Copies round 4's measurement matrix (tools/r04_measure_all.sh) from gpurun_out/ into profiles/ (r04_*) and folds every PMC session
# into ik_amd/kernel_stats.json (tools/pmc_to_stats.py: one session, one build, stamped).
set -e
cd "$(dirname "$0")/.."
mkdir -p profiles/r04_pmc
for w in cassie_leg ur5 ur10 ur5_clamp ur10_clamp arm7 ur5_two_tasks cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pinned cassie_demo_pinned_posture cassie_demo_pik cassie_two_feet_pik ur5_pos_then_ori_pik cassie_three_feet; do
  [ -s gpurun_out/bench_$w.json ] && cp gpurun_out/bench_$w.json profiles/r04_bench_$w.json
  f=$(ls -t gpurun_out/stats_$w/runc/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/r04_kernel_stats_$w.csv
done
for s in cassie_leg cassie_leg_refill cassie_leg_lockstep cassie_leg_general ur5 arm7 ur5_two_tasks cassie_full_body cassie_demo cassie_demo_tree cassie_demo_coop cassie_demo_posture cassie_demo_posture_tree cassie_demo_pinned cassie_demo_pinned_tree cassie_demo_pinned_posture cassie_demo_pik cassie_two_feet_pik ur5_pos_then_ori_pik cassie_three_feet; do
  [ -d gpurun_out/r04_pmc_$s ] && python3 tools/pmc_to_stats.py gpurun_out/r04_pmc_$s profiles/r04_pmc $s | cut -c1-400
done
for f in r04_refill_timing_chain.txt r04_refill_timing_tree.txt r04_refill_timing_static.txt r04_creation_timing.txt r04_bench_cassie_full_body_static.json r04_constraint_timing.txt r04_generic_forms_rows_31_32.txt r04_host_entry.txt r04_chain_builds.txt r04_generic_forms.txt r04_bench_launcher_n1.json r04_bench_launcher_n1_gather_full.json r04_pik_timing.txt r04_host_entry_tails.txt r04_generic_forms_dual.txt r04_two_phase_probe_chain.txt r04_two_phase_probe_tree.txt r04_refill_batch.txt r04_bench_default.json; do
  [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
[ -s gpurun_out/parity_counts.json ] && cp gpurun_out/parity_counts.json profiles/r04_parity_counts.json
ls profiles | grep r04 | wc -l
