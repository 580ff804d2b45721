#!/usr/bin/env bash
# Per-phase cycle counters of the cooperative kernels: a VARIANT build beside the production library (never over it),
# selected with IKGPU_LIB.
cd "$GRAFT_REPO_ROOT"
lib=$(tools/build_variant.sh coopprof -DIKGPU_COOP_PROFILE | tail -1)
IKGPU_LIB="$lib" python tools/coop_profile.py pik
IKGPU_LIB="$lib" python tools/coop_profile.py
for c in com_under_feet feet_frames_beyond_the_register_solve posture_regulariser; do IKGPU_LIB="$lib" python tools/coop_profile.py dls $c; done
