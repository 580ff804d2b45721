#!/usr/bin/env bash
# Copies the measurement matrix of tools/r02_measure_all.sh from gpurun_out/ into profiles/ (r02_*) and folds the PMC passes into
# ik_amd/kernel_stats.json.
set -e
cd "$(dirname "$0")/.."
for w in cassie_leg ur5 ur10 ur5_clamp ur10_clamp cassie_full_body cassie_demo cassie_demo_posture cassie_demo_pinned cassie_demo_pinned_posture cassie_demo_pik; do
  [ -f gpurun_out/bench_$w.json ] && cp gpurun_out/bench_$w.json profiles/r02_bench_$w.json
  f=$(ls -t gpurun_out/stats_$w/runc/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" profiles/r02_kernel_stats_$w.csv
done
for f in bench_launcher_n1 bench_launcher_n1_gather_full bench_launcher_n1_gather_compact bench_strong_n1; do [ -f gpurun_out/$f.json ] && cp gpurun_out/$f.json profiles/r02_$f.json; done
for f in loop_stamps.txt iter_sweep.txt constraint_timing.txt; do [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/r02_$f; done
[ -f gpurun_out/issue_probe.csv ] && cp gpurun_out/issue_probe.csv profiles/r02_issue_probe.csv
[ -f gpurun_out/issue_probe_dep.csv ] && cp gpurun_out/issue_probe_dep.csv profiles/r02_issue_probe_dep.csv
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_leg "dls_chain<NJ=7,full>" profiles/r02_pmc leg | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_ur5 "dls_chain<NJ=6,full>" profiles/r02_pmc ur5 | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_full_body "dls_tree<NJ=7,chains=2,base_task>" profiles/r02_pmc full | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis>" profiles/r02_pmc demo | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo_posture "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis,posture>" profiles/r02_pmc posture | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo_pinned "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis,constraint_rows=3>" profiles/r02_pmc pinned | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo_pik "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis,pik_levels=2>" profiles/r02_pmc pik | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo_pinned_posture "dls_tree<NJ=7,chains=1,base_task,base_reference,align_axis,posture,constraint_rows=3>" profiles/r02_pmc pinned_posture | cut -c1-80
python3 tools/pmc_to_stats.py gpurun_out/pmc_cassie_demo_generic "dls_generic<M=10,nv=22,joints=17>" profiles/r02_pmc generic_demo | cut -c1-80
