"""ik_amd -- MI355X-native batched damped-least-squares inverse kinematics.

One hot path of dazzmo/ik -- ik::dls() (reference ik/ik/dls.cpp:5-78), and its sibling ik::pik()
(reference ik/ik/pik.cpp:31-103) -- as hand-written gfx950 kernels behind a C ABI (include/ikgpu.h, ik_amd/libikgpu.so).  This package is the host-side
mirror of the reference API for that path plus the ctypes plumbing; it fails loudly when the
native library is missing and has no CPU fallback.
"""
from .api import (AlignAxisTask, AlignAxisType, CentreOfMassTask, FrameConstraint, FrameTask, InverseKinematicsProblem, KinematicType, Model, PostureTask, Problem, SE3,  # noqa: F401
                  dls, dls_batch, dls_data, dls_parameters, evaluate_batch, inverse_kinematics_visitor, never_stop_visitor,
                  pik, pik_batch, pik_data, pik_parameters, plan, precompile, task_frames_fk_batch)
from .capi import IkgpuError  # noqa: F401
