"""ctypes binding of the C ABI in include/ikgpu.h (ik_amd/libikgpu.so).

Plumbing only: every computation happens behind the C ABI, in the hand-written gfx950 kernels.
If the shared library is missing this module raises -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IKGPU_LIB: another build of the same library (A/B of kernel variants: `make -C ik_amd/csrc OUT=... OBJ=... KERNEL_EXTRA=-D...`
# writes it beside the production file instead of over it)
LIB_PATH = os.environ.get("IKGPU_LIB") or os.path.join(_HERE, "libikgpu.so")

OK, ERR_INVALID, ERR_PARSE, ERR_UNSUPPORTED, ERR_DEVICE = 0, 1, 2, 3, 4
JOINT_UNIVERSE, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FREEFLYER = 0, 1, 2, 3
POSITION, ORIENTATION, FULL = 0, 1, 2
ALIGN_AXIS_X, ALIGN_AXIS_Y, ALIGN_AXIS_Z, POSTURE_ROW, CENTRE_OF_MASS = 3, 4, 5, 6, 7
SOA, AOS = 0, 1
TARGETS_POSE7 = 0x100
ROOT_FIXED, ROOT_FREEFLYER = 0, 1

# Every symbol include/ikgpu.h declares (tests check the library exports all of them).
SYMBOLS = [
    "ikgpu_abi_version", "ikgpu_last_error", "ikgpu_dls_params_default",
    "ikgpu_model_from_urdf", "ikgpu_model_create", "ikgpu_model_destroy", "ikgpu_model_get_flat",
    "ikgpu_model_frame_id", "ikgpu_model_joint_id",
    "ikgpu_problem_create", "ikgpu_problem_destroy", "ikgpu_problem_rows", "ikgpu_problem_kernel",
    "ikgpu_problem_plan",
    "ikgpu_dls_solve_batch", "ikgpu_dls_solve_batch_host", "ikgpu_evaluate_batch", "ikgpu_task_frames_fk_batch",
    "ikgpu_pik_params_default", "ikgpu_pik_solve_batch", "ikgpu_pik_solve_batch_host", "ikgpu_pik_kernel",
    "ikgpu_problem_create_constrained", "ikgpu_problem_plan_constrained", "ikgpu_problem_support",
    "ikgpu_problem_precompile", "ikgpu_rtc_worker_compile",
    "ikgpu_shard_range", "ikgpu_shard_slot_layout", "ikgpu_shard_slot_bytes", "ikgpu_shard_group_create", "ikgpu_shard_group_destroy",
    "ikgpu_shard_group_size", "ikgpu_shard_group_problem", "ikgpu_shard_group_uses_rccl", "ikgpu_shard_group_stream",
    "ikgpu_shard_group_last_issue_us",
    "ikgpu_dls_solve_batch_sharded", "ikgpu_shard_group_synchronize", "ikgpu_targets_from_pose7",
]
MAX_PIK_LEVELS, MAX_PIK_DA = 8, 128


class FlatModel(C.Structure):
    _fields_ = [("njoints", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32), ("nframes", C.c_int32),
                ("joint_type", C.POINTER(C.c_int32)), ("joint_parent", C.POINTER(C.c_int32)),
                ("joint_idx_q", C.POINTER(C.c_int32)), ("joint_idx_v", C.POINTER(C.c_int32)),
                ("joint_placement", C.POINTER(C.c_double)), ("joint_axis", C.POINTER(C.c_double)),
                ("lower", C.POINTER(C.c_double)), ("upper", C.POINTER(C.c_double)),
                ("frame_parent", C.POINTER(C.c_int32)), ("frame_placement", C.POINTER(C.c_double)),
                ("joint_names", C.POINTER(C.c_char_p)), ("frame_names", C.POINTER(C.c_char_p)),
                ("joint_mass", C.POINTER(C.c_double)), ("joint_com", C.POINTER(C.c_double))]


class Task(C.Structure):
    _fields_ = [("frame", C.c_int32), ("reference", C.c_int32), ("type", C.c_int32), ("priority", C.c_int32),
                ("weight", C.c_double * 6)]


MAX_VISITOR_LEVELS = 8


class DlsParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("damping", C.c_double), ("step_length", C.c_double),
                ("stop_sq_tol", C.c_double), ("dq_sq_tol", C.c_double), ("num_level_tols", C.c_int32),
                ("level_sq_tol", C.c_double * MAX_VISITOR_LEVELS)]


class PikParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("step_length", C.c_double), ("stop_sq_tol", C.c_double),
                ("num_levels", C.c_int32), ("lam", C.c_double * MAX_PIK_LEVELS), ("da", C.POINTER(C.c_double))]


class IkgpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ikgpu error %d: %s" % (code, message))
        self.code = code
        self.message = message


_lib = None


def lib():
    """Load libikgpu.so (after torch, so that both share one HIP runtime). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ik_amd/libikgpu.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). ik_amd has no CPU or PyTorch fallback.")
    try:
        import torch  # noqa: F401  -- loads libamdhip64 first; libikgpu.so then binds to the same runtime
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.ikgpu_abi_version.restype = C.c_int
    L.ikgpu_last_error.restype = C.c_char_p
    L.ikgpu_dls_params_default.argtypes = [C.POINTER(DlsParams)]
    L.ikgpu_dls_params_default.restype = None
    L.ikgpu_model_from_urdf.argtypes = [C.c_char_p, sz, C.c_int, C.POINTER(vp)]
    L.ikgpu_model_create.argtypes = [C.POINTER(FlatModel), C.POINTER(vp)]
    L.ikgpu_model_destroy.argtypes = [vp]
    L.ikgpu_model_destroy.restype = None
    L.ikgpu_model_get_flat.argtypes = [vp, C.POINTER(FlatModel)]
    L.ikgpu_model_frame_id.argtypes = [vp, C.c_char_p]
    L.ikgpu_model_frame_id.restype = i32
    L.ikgpu_model_joint_id.argtypes = [vp, C.c_char_p]
    L.ikgpu_model_joint_id.restype = i32
    L.ikgpu_problem_create.argtypes = [vp, C.POINTER(Task), i32, i32, C.POINTER(vp)]
    L.ikgpu_problem_destroy.argtypes = [vp]
    L.ikgpu_problem_destroy.restype = None
    L.ikgpu_problem_rows.argtypes = [vp]
    L.ikgpu_problem_rows.restype = i32
    L.ikgpu_problem_kernel.argtypes = [vp]
    L.ikgpu_problem_kernel.restype = C.c_char_p
    L.ikgpu_problem_support.argtypes = [vp, vp]
    L.ikgpu_problem_support.restype = C.c_int
    L.ikgpu_problem_plan.argtypes = [vp, C.POINTER(Task), i32, C.c_char_p, sz]
    L.ikgpu_problem_create_constrained.argtypes = [vp, C.POINTER(Task), i32, C.POINTER(Task), i32, i32, C.POINTER(vp)]
    L.ikgpu_problem_plan_constrained.argtypes = [vp, C.POINTER(Task), i32, C.POINTER(Task), i32, C.c_char_p, sz]
    L.ikgpu_problem_precompile.argtypes = [vp, C.POINTER(Task), i32, C.POINTER(Task), i32, C.c_char_p, sz]
    L.ikgpu_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.ikgpu_shard_range.restype = None
    L.ikgpu_shard_slot_layout.argtypes = [i32, i64, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.ikgpu_shard_slot_layout.restype = sz
    L.ikgpu_shard_slot_bytes.argtypes = [i32, i64, i32]
    L.ikgpu_shard_slot_bytes.restype = sz
    L.ikgpu_shard_group_create.argtypes = [vp, C.POINTER(Task), i32, C.POINTER(Task), i32, C.POINTER(i32), i32, C.POINTER(vp)]
    L.ikgpu_shard_group_destroy.argtypes = [vp]
    L.ikgpu_shard_group_destroy.restype = None
    L.ikgpu_shard_group_size.argtypes = [vp]
    L.ikgpu_shard_group_size.restype = i32
    L.ikgpu_shard_group_problem.argtypes = [vp, i32]
    L.ikgpu_shard_group_problem.restype = vp
    L.ikgpu_shard_group_uses_rccl.argtypes = [vp]
    L.ikgpu_shard_group_uses_rccl.restype = i32
    L.ikgpu_shard_group_stream.argtypes = [vp, i32]
    L.ikgpu_shard_group_stream.restype = vp
    L.ikgpu_dls_solve_batch_sharded.argtypes = [vp, i64, C.POINTER(vp), C.POINTER(vp), C.POINTER(DlsParams), C.POINTER(vp)]
    L.ikgpu_shard_group_synchronize.argtypes = [vp]
    L.ikgpu_targets_from_pose7.argtypes = [i64, i32, vp, vp, C.c_int, vp]
    L.ikgpu_dls_solve_batch.argtypes = [vp, i64, vp, vp, C.POINTER(DlsParams), vp, vp, vp, C.c_int, vp]
    L.ikgpu_dls_solve_batch_host.argtypes = [vp, i64, vp, vp, C.POINTER(DlsParams), vp, vp, vp, C.c_int]
    L.ikgpu_pik_params_default.argtypes = [C.POINTER(PikParams), i32]
    L.ikgpu_pik_params_default.restype = None
    L.ikgpu_pik_solve_batch.argtypes = [vp, i64, vp, vp, C.POINTER(PikParams), vp, vp, vp, C.c_int, vp]
    L.ikgpu_pik_solve_batch_host.argtypes = [vp, i64, vp, vp, C.POINTER(PikParams), vp, vp, vp, C.c_int]
    L.ikgpu_pik_kernel.argtypes = [vp, C.POINTER(PikParams)]
    L.ikgpu_pik_kernel.restype = C.c_char_p
    L.ikgpu_evaluate_batch.argtypes = [vp, i64, vp, vp, vp, vp, C.c_int, vp]
    L.ikgpu_task_frames_fk_batch.argtypes = [vp, i64, vp, vp, C.c_int, vp]
    for name in ("ikgpu_model_from_urdf", "ikgpu_model_create", "ikgpu_model_get_flat", "ikgpu_problem_create",
                 "ikgpu_problem_plan", "ikgpu_dls_solve_batch", "ikgpu_dls_solve_batch_host", "ikgpu_evaluate_batch",
                 "ikgpu_task_frames_fk_batch", "ikgpu_pik_solve_batch", "ikgpu_pik_solve_batch_host",
                 "ikgpu_problem_create_constrained", "ikgpu_problem_plan_constrained"):
        getattr(L, name).restype = C.c_int
    if L.ikgpu_abi_version() != 2:
        raise ImportError("libikgpu.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise IkgpuError(rc, lib().ikgpu_last_error().decode("utf-8", "replace"))
