"""Deterministic synthetic IK workloads: counter-based SplitMix64 keyed by
(seed, problem index, component), so any slice of a batch can be generated independently --
each rank of a multi-GPU run generates exactly its own shard.  Host logic only (numpy)."""
import json
import os

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
MODELS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fixtures", "models")


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, index, component):
    """U[0,1) for every (index, component) pair; index: int array [B], component: int array [C] -> [B, C]."""
    idx = np.asarray(index, dtype=np.uint64)[:, None]
    comp = np.asarray(component, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = splitmix64(np.uint64(seed)) ^ splitmix64(idx * np.uint64(0xD1342543DE82EF95) + comp + np.uint64(1))
    z = splitmix64(key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, index, ncomp, lo, hi, stream=0):
    u = uniform01(seed, index, np.arange(ncomp) + 1000 * stream)
    return lo + (hi - lo) * u


def cassie_nominal(joint_names):
    """SRDF default pose of the 16 Cassie leg joints (fixtures/models/cassie.nominal.json)."""
    with open(os.path.join(MODELS_DIR, "cassie.nominal.json")) as fh:
        vals = json.load(fh)["joints"]
    # (joints a test added to the model -- "tail-joint*", tests/test_gpu_generic.py -- rest at zero)
    return np.array([vals.get(n, 0.0) for n in joint_names if n in vals or n.startswith("tail-joint")])


UR5_NOMINAL = np.array([0.0, -np.pi / 2, np.pi / 2, 0.0, np.pi / 2, 0.0])


def chain_workload(lower, upper, nominal, index, seed=0, mode="uniform", narrow=None):
    """Configurations for a fixed-base model.  Returns (q0 [B,nq], qstar [B,nq]); the target of
    problem b is FK(qstar[b]) so every target is reachable by construction.
      q0    = clamp(nominal + U(-0.1, 0.1))
      qstar = U(lower, upper)                       mode "uniform"
            = clamp(q0 + U(-0.15, 0.15))            mode "near"
    narrow: optional +-limit (rad) intersected with the model limits before sampling qstar."""
    lower, upper = np.asarray(lower, float), np.asarray(upper, float)
    nq = lower.size
    q0 = np.clip(nominal[None, :] + uniform(seed, index, nq, -0.1, 0.1, stream=0), lower, upper)
    if mode == "near":
        qs = np.clip(q0 + uniform(seed, index, nq, -0.15, 0.15, stream=1), lower, upper)
    else:
        lo, hi = lower, upper
        if narrow is not None:
            lo, hi = np.maximum(lower, -narrow), np.minimum(upper, narrow)
        qs = uniform(seed, index, nq, lo[None, :], hi[None, :], stream=1)
    return q0, qs


def freeflyer_workload(lower, upper, nominal, index, seed=0, integrate=None, mode="near"):
    """Shape F: free-flyer q = (x y z qx qy qz qw | joints).
      q0    : base at (0, 0, 1) + small noise, identity quaternion + small noise (normalised),
              joints = clamp(nominal + U(-0.1, 0.1))
      qstar : joints = clamp(q0 + U(-0.15, 0.15)) ("near") or U(lower, upper) ("uniform"),
              base  = q0's base moved by the body twist (U(+-0.1) m, U(+-0.2) rad) through `integrate`
    `integrate(q, v)` is the SE(3) configuration update (callers pass the oracle's or build targets on
    the device); returns (q0 [B, nq], qstar [B, nq])."""
    lower, upper = np.asarray(lower, float), np.asarray(upper, float)
    nj = lower.size - 7
    B = len(index)
    q0 = np.zeros((B, 7 + nj))
    q0[:, :3] = np.array([0.0, 0.0, 1.0]) + uniform(seed, index, 3, -0.02, 0.02, stream=2)
    quat = np.array([0.0, 0.0, 0.0, 1.0]) + uniform(seed, index, 4, -0.02, 0.02, stream=3)
    q0[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    q0[:, 7:] = np.clip(nominal[None, :] + uniform(seed, index, nj, -0.1, 0.1, stream=0), lower[7:], upper[7:])
    qs = q0.copy()
    if mode == "near":
        qs[:, 7:] = np.clip(q0[:, 7:] + uniform(seed, index, nj, -0.15, 0.15, stream=1), lower[7:], upper[7:])
    else:
        qs[:, 7:] = uniform(seed, index, nj, lower[None, 7:], upper[None, 7:], stream=1)
    v = np.zeros((B, 6 + nj))
    v[:, :3] = uniform(seed, index, 3, -0.1, 0.1, stream=4)
    v[:, 3:6] = uniform(seed, index, 3, -0.2, 0.2, stream=5)
    if integrate is None:
        integrate = freeflyer_integrate_batch
        return q0, integrate(qs, v)
    return q0, np.stack([integrate(qs[b], v[b]) for b in range(B)])


def freeflyer_integrate_batch(q, v):
    """Vectorised numpy q (+) v for a free-flyer model (input generation only): base pose composed with
    exp6 of the body twist v[:, :6], joints q + v.  Quaternion (x y z w), unit on output."""
    q, v = np.asarray(q, float), np.asarray(v, float)
    out = q.copy()
    out[:, 7:] = q[:, 7:] + v[:, 6:]
    vl, w = v[:, :3], v[:, 3:6]
    t2 = np.sum(w * w, axis=1)
    t = np.sqrt(t2)
    small = t < 1e-6
    ts = np.where(small, 1.0, t)
    a_v = np.where(small, 1.0 - t2 / 6.0, np.sin(ts) / ts)
    a_wxv = np.where(small, 0.5 - t2 / 24.0, (1.0 - np.cos(ts)) / np.where(small, 1.0, t2))
    a_w = np.where(small, 1.0 / 6.0 - t2 / 120.0, (1.0 - a_v) / np.where(small, 1.0, t2))
    trans = a_v[:, None] * vl + (a_w * np.sum(w * vl, axis=1))[:, None] * w + a_wxv[:, None] * np.cross(w, vl)
    # rotation increment as a quaternion: (sin(t/2) w/t, cos(t/2))
    half = 0.5 * t
    k = np.where(small, 0.5 - t2 / 48.0, np.sin(half) / ts)
    dq = np.concatenate([k[:, None] * w, np.cos(half)[:, None]], axis=1)
    x, y, z, ww = q[:, 3], q[:, 4], q[:, 5], q[:, 6]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)], 1),
                  np.stack([2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)], 1),
                  np.stack([2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)], 1)], 1)
    out[:, :3] = q[:, :3] + np.einsum("bij,bj->bi", R, trans)
    x2, y2, z2, w2 = dq[:, 0], dq[:, 1], dq[:, 2], dq[:, 3]
    qq = np.stack([ww * x2 + x * w2 + y * z2 - z * y2,
                   ww * y2 - x * z2 + y * w2 + z * x2,
                   ww * z2 + x * y2 - y * x2 + z * w2,
                   ww * w2 - x * x2 - y * y2 - z * z2], 1)
    out[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    return out
