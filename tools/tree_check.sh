set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_tree_posture.py tests/test_gpu_constraints.py tests/test_gpu_pik.py tests/test_gpu_parity.py tests/test_gpu_refill.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/tree_check_tests.txt; rc=$?
cat gpurun_out/tree_check_tests.txt
[ $rc -eq 0 ] || exit $rc
for w in cassie_full_body cassie_demo_posture cassie_demo_pinned cassie_demo_pinned_posture cassie_demo_pik; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['config'].get('kernel'), d['ms_per_step'])" || exit 1
done
IKGPU_TREE_STATIC_ROWS=0 timeout -k 10 200 python bench.py --workload cassie_demo --no-cpu --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('demo tree', d['config'].get('kernel'), d['ms_per_step'])"
