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
    return np.array([vals[n] for n in joint_names if n in vals])


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
