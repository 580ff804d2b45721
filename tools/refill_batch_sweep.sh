set -e
for b in 1 16 32; do echo "== IKGPU_REFILL_BATCH=$b chain"; IKGPU_REFILL_BATCH=$b timeout -k 10 200 python tools/refill_timing.py cassie_fixed LeftFootFront 2>&1 | grep -v amdgpu | grep -E "==|B= 262144 max_it=100|B=1048576 max_it=100" | head -6; done
for b in 1 16 32; do echo "== IKGPU_REFILL_BATCH=$b tree"; IKGPU_REFILL_BATCH=$b timeout -k 10 250 python tools/refill_timing.py full_body x 2>&1 | grep -v amdgpu | grep -E "==|B= 262144 max_it=100|B=1048576 max_it=100"; done
