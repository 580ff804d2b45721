"""The C oracle against the committed golden vectors (tests/golden/*.json, produced by the
independently written numpy twin -- tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import urdf_path

import oracle as O
import twin as T

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["S_cassie_leg", "U_ur5", "F_cassie_full"]


def load(name):
    with open(os.path.join(HERE, "golden", name + ".json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden_vectors(native_built, case):
    g = load(case)
    tm = T.load_urdf(os.path.join(os.path.dirname(urdf_path("x")), g["urdf"]), g["free_flyer"])
    om = O.OracleModel(O.flat_from_twin(tm))
    fids = [om.frame_id(f) for f in g["task_frames"]]
    tasks = O.make_tasks([(f, 0, 2, 0, None) for f in fids])
    for p in g["problems"]:
        q0, tg = np.array(p["q0"]), np.array(p["targets"])
        assert np.abs(O.fk(om, q0)[1][fids] - np.array(p["oMf_q0"])).max() < 1e-14
        e, J = O.evaluate(om, tasks, tg, q0)
        assert np.abs(e - p["e"]).max() < 1e-13 and np.abs(J - np.array(p["J"])).max() < 1e-12
        for k, key in ((1, "q_after_1"), (3, "q_after_3"), (50, "q_after_50")):
            q, ok, it = O.dls(om, tasks, tg, q0, O.params(k, 1e-2, 1.0, -1.0))
            assert np.abs(q - p[key]).max() < 1e-9, (key, np.abs(q - p[key]).max())
            assert not ok and it == k
        q, ok, it = O.dls(om, tasks, tg, q0, O.params())
        d = p["default_stop"]
        assert ok == d["success"] and it == d["iterations"] and np.abs(q - d["q"]).max() < 1e-9
        # first iteration's step direction, from the trace
        _, _, _, tr = O.dls(om, tasks, tg, q0, O.params(1, 1e-2, 1.0, -1.0), trace=True)
        M = len(p["e"])
        assert np.abs(tr[0, om.nq + M:] - p["dq"]).max() < 1e-10


def test_oracle_lie_maps_reproduce_golden(native_built):
    for row in load("lie_maps"):
        M = np.array(row["M"])
        assert np.abs(O.exp6(row["nu"]) - M).max() < 1e-14
        assert np.abs(O.log6(M) - row["log6"]).max() < 1e-11
        assert np.abs(O.Jlog6(M) - np.array(row["Jlog6"])).max() < 1e-10


def test_golden_files_are_reproducible(native_built, tmp_path):
    """make_golden.py is deterministic: regenerating S must give the committed numbers."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    fresh = mg.case("S_cassie_leg", "cassie_fixed.kin.urdf", False, ["LeftFootFront"], 8, 11)
    old = load("S_cassie_leg")
    for a, b in zip(fresh["problems"], old["problems"]):
        assert np.abs(np.array(a["q_after_50"]) - np.array(b["q_after_50"])).max() < 1e-12
        assert np.array_equal(np.array(a["q0"]), np.array(b["q0"]))
