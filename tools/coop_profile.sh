#!/usr/bin/env bash
cd "$GRAFT_REPO_ROOT"
touch ik_amd/csrc/kernels.hip; make -s -C ik_amd/csrc KERNEL_EXTRA="-DIKGPU_COOP_PROFILE" 2>&1 | tail -3
python tools/coop_profile.py pik
python tools/coop_profile.py
