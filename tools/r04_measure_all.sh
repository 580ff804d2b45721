#!/usr/bin/env bash
# Round 4's measurement matrix, in parts (a gpurun call lasts at most 20 minutes):
#   tools/r04_measure_all.sh stats     bench line + rocprofv3 --kernel-trace --stats of every bench workload
#   tools/r04_measure_all.sh pmc1|pmc2 PMC sessions (tools/pmc_session.sh: fresh directories, one build per session), incl. the
#                                      A/B variants an environment switch selects at problem creation
#   tools/r04_measure_all.sh misc      refill / host entry / chain builds / generic forms timings, launcher rehearsals, probes
# Outputs under gpurun_out/; tools/r04_collect.sh folds them into profiles/ and ik_amd/kernel_stats.json.
part="${1:-stats}"
steps=()
case "$part" in
stats)
  for w in cassie_leg ur5 ur10 ur5_clamp ur10_clamp arm7 ur5_two_tasks cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pinned cassie_demo_pinned_posture cassie_demo_pik cassie_two_feet_pik ur5_pos_then_ori_pik cassie_three_feet; do
    steps+=("stats_$w|240|tools/stats_session.sh $w")
  done ;;
pmc1)
  steps+=("pmc_cassie_leg|300|tools/pmc_session.sh cassie_leg r04_pmc_cassie_leg")
  steps+=("pmc_cassie_leg_general|300|IKGPU_CHAIN_HOT=0 tools/pmc_session.sh cassie_leg r04_pmc_cassie_leg_general")
  steps+=("pmc_ur5|300|tools/pmc_session.sh ur5 r04_pmc_ur5")
  steps+=("pmc_arm7|300|tools/pmc_session.sh arm7 r04_pmc_arm7")
  steps+=("pmc_ur5_two_tasks|300|tools/pmc_session.sh ur5_two_tasks r04_pmc_ur5_two_tasks")
  steps+=("pmc_cassie_full_body|300|tools/pmc_session.sh cassie_full_body r04_pmc_cassie_full_body")
  steps+=("pmc_cassie_demo|300|tools/pmc_session.sh cassie_demo r04_pmc_cassie_demo")
  steps+=("pmc_cassie_two_feet_pik|300|tools/pmc_session.sh cassie_two_feet_pik r04_pmc_cassie_two_feet_pik")
  steps+=("pmc_ur5_pos_then_ori_pik|300|tools/pmc_session.sh ur5_pos_then_ori_pik r04_pmc_ur5_pos_then_ori_pik")
  steps+=("pmc_cassie_three_feet|300|tools/pmc_session.sh cassie_three_feet r04_pmc_cassie_three_feet") ;;
pmc2)
  steps+=("pmc_cassie_demo_tree|300|IKGPU_TREE_STATIC_ROWS=0 tools/pmc_session.sh cassie_demo r04_pmc_cassie_demo_tree")
  steps+=("pmc_cassie_demo_coop|400|IKGPU_DLS_KERNEL=generic IKGPU_GENERIC_STATIC=0 tools/pmc_session.sh cassie_demo r04_pmc_cassie_demo_coop")
  steps+=("pmc_cassie_demo_posture|300|tools/pmc_session.sh cassie_demo_posture r04_pmc_cassie_demo_posture")
  steps+=("pmc_cassie_demo_posture_tree|300|IKGPU_TREE_STATIC_ROWS=0 tools/pmc_session.sh cassie_demo_posture r04_pmc_cassie_demo_posture_tree")
  steps+=("pmc_cassie_demo_pinned|300|tools/pmc_session.sh cassie_demo_pinned r04_pmc_cassie_demo_pinned")
  steps+=("pmc_cassie_demo_pinned_tree|300|IKGPU_TREE_STATIC_ROWS=0 tools/pmc_session.sh cassie_demo_pinned r04_pmc_cassie_demo_pinned_tree")
  steps+=("pmc_cassie_demo_pinned_posture|300|tools/pmc_session.sh cassie_demo_pinned_posture r04_pmc_cassie_demo_pinned_posture")
  steps+=("pmc_cassie_demo_pik|300|tools/pmc_session.sh cassie_demo_pik r04_pmc_cassie_demo_pik") ;;
pmc3)   # the default stop rule on config 4's batch: the lane-refill kernel against the lock-step kernel (IKGPU_REFILL=0)
  steps+=("pmc_cassie_leg_refill|300|IKGPU_REFILL=1 tools/pmc_session.sh cassie_leg r04_pmc_cassie_leg_refill --stop-rule --batch 262144")
  steps+=("pmc_cassie_leg_lockstep|300|IKGPU_REFILL=0 tools/pmc_session.sh cassie_leg r04_pmc_cassie_leg_lockstep --stop-rule --batch 262144") ;;
misc)
  steps+=("refill_chain|200|python3 tools/refill_timing.py > gpurun_out/r04_refill_timing_chain.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_refill_timing_chain.txt | tail -4")
  steps+=("refill_tree|200|python3 tools/refill_timing.py full_body x > gpurun_out/r04_refill_timing_tree.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_refill_timing_tree.txt | tail -4")
  steps+=("refill_static|300|python3 tools/refill_timing.py demo x > gpurun_out/r04_refill_timing_static.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_refill_timing_static.txt | tail -4")
  steps+=("creation|600|python3 tools/creation_timing.py > gpurun_out/r04_creation_timing.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_creation_timing.txt")
  steps+=("fullbody_static|400|IKGPU_TREE_STATIC_ROWS=18 python3 bench.py --workload cassie_full_body --no-cpu --timed-only 2>/dev/null | grep '^{' > gpurun_out/r04_bench_cassie_full_body_static.json; cut -c1-400 gpurun_out/r04_bench_cassie_full_body_static.json")
  steps+=("constraints|600|python3 tools/constraint_timing.py 2>&1 | grep -v amdgpu.ids | grep cholqr > gpurun_out/r04_constraint_timing.txt; cat gpurun_out/r04_constraint_timing.txt")
  steps+=("rows_31_32|900|FORMS=coop,static python3 tools/generic_forms.py rows_31 rows_32 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_generic_forms_rows_31_32.txt; cat gpurun_out/r04_generic_forms_rows_31_32.txt")
  steps+=("host_entry|200|python3 tools/host_entry_timing.py > gpurun_out/r04_host_entry.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_host_entry.txt | tail -4")
  steps+=("chain_builds|200|python3 tools/chain_builds_timing.py > gpurun_out/r04_chain_builds.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_chain_builds.txt")
  steps+=("generic_forms|400|FORMS=coop,static python3 tools/generic_forms.py shared_joints com_of_the_arm moving_reference_prismatic demo_task_set com_under_feet three_feet_frames feet_frames_beyond_the_register_solve rows_16 nv_30 fixed_two_feet_priorities com_in_foot_frame > gpurun_out/r04_generic_forms.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_generic_forms.txt")
  steps+=("launcher|300|python3 bench.py --launcher --no-cpu > gpurun_out/r04_bench_launcher_n1.json; cut -c1-250 gpurun_out/r04_bench_launcher_n1.json")
  steps+=("launcher_full|300|python3 bench.py --launcher --gather full --no-cpu > gpurun_out/r04_bench_launcher_n1_gather_full.json; cut -c1-250 gpurun_out/r04_bench_launcher_n1_gather_full.json")
  steps+=("pik_timing|300|python3 tools/pik_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_pik_timing.txt; cat gpurun_out/r04_pik_timing.txt")
  steps+=("host_tails|200|python3 tools/host_entry_tails.py 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_host_entry_tails.txt; cat gpurun_out/r04_host_entry_tails.txt")
  steps+=("two_phase_chain|300|python3 tools/two_phase_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_two_phase_probe_chain.txt; tail -3 gpurun_out/r04_two_phase_probe_chain.txt")
  steps+=("two_phase_tree|300|python3 tools/two_phase_probe.py tree 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_two_phase_probe_tree.txt; tail -3 gpurun_out/r04_two_phase_probe_tree.txt")
  steps+=("refill_batch|600|bash tools/refill_batch_sweep.sh > gpurun_out/r04_refill_batch.txt 2>&1; tail -3 gpurun_out/r04_refill_batch.txt")
  steps+=("bench_default|300|python3 bench.py > gpurun_out/r04_bench_default.json 2>/dev/null; cut -c1-300 gpurun_out/r04_bench_default.json")
  steps+=("forms_dual|300|IKGPU_STATIC_FORM=dual FORMS=static python3 tools/generic_forms.py three_feet_frames rows_16 feet_frames_beyond_the_register_solve nv_30 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_generic_forms_dual.txt; cat gpurun_out/r04_generic_forms_dual.txt") ;;
esac
tools/gpu_session.sh "${steps[@]}"
