#!/usr/bin/env python3
"""Writes fixtures/models/arm7.kin.urdf: a made-up 7-joint serial arm whose joint origins carry GENERAL rotations (rpy values that
are no multiples of pi/2) and whose joint axes are oblique unit vectors.

NOT a file of the reference and not a model of any real robot.  It exists because the reference's loader takes ANY URDF
(pinocchio::urdf::buildModelFromXML, reference ik_ros/src/cassie.cpp:34-35) while the fixture robots (Cassie, UR5 / UR10) are
all near-permutation kinematics: this arm has no structurally zero or unit placement entry, so it exercises the general chain
build (device/chain_solver.hpp) and a placement-structure code no pre-built kernel has (device/chain_hot.hpp).

    python fixtures/make_arm7_urdf.py
"""
import os

# (name, parent, child, xyz, rpy, axis, (lower, upper))
JOINTS = [
    ("j1", "base", "l1", (0.05, -0.02, 0.31), (0.11, -0.23, 0.37), (0.0, 0.6, 0.8), (-2.6, 2.6)),
    ("j2", "l1", "l2", (0.02, 0.12, 0.08), (-0.41, 0.52, 0.13), (0.8, 0.0, 0.6), (-1.9, 1.9)),
    ("j3", "l2", "l3", (-0.03, 0.04, 0.36), (0.27, 0.19, -0.58), (0.36, 0.48, 0.8), (-2.8, 2.8)),
    ("j4", "l3", "l4", (0.07, -0.05, 0.09), (1.02, -0.31, 0.22), (0.6, 0.8, 0.0), (-2.2, 0.4)),
    ("j5", "l4", "l5", (0.01, 0.03, 0.33), (-0.17, 0.44, 0.71), (0.0, 0.28, 0.96), (-2.9, 2.9)),
    ("j6", "l5", "l6", (-0.04, 0.06, 0.07), (0.63, 0.08, -0.35), (0.96, 0.0, 0.28), (-1.7, 2.1)),
    ("j7", "l6", "l7", (0.02, -0.01, 0.11), (-0.29, -0.47, 0.16), (0.48, 0.6, 0.64), (-3.0, 3.0)),
]
TOOL = ("tool_joint", "l7", "tool", (0.03, 0.02, 0.14), (0.21, -0.12, 0.33))
MASSES = {"base": 3.0, "l1": 2.7, "l2": 2.4, "l3": 2.1, "l4": 1.6, "l5": 1.2, "l6": 0.9, "l7": 0.5, "tool": 0.0}


def num(v):
    return " ".join(repr(float(x)) for x in v)


def main():
    out = ['<?xml version="1.0"?>',
           "<!-- Made-up 7-joint arm with general joint-origin rotations and oblique axes, written by fixtures/make_arm7_urdf.py.",
           "     NOT a file of the reference and not a real robot: a non-fixture chain for the general / run-time specialised chain kernels. -->",
           '<robot name="arm7">']
    for name in ["base"] + [j[2] for j in JOINTS] + ["tool"]:
        out += ['  <link name="%s">' % name, "    <inertial>", '      <origin rpy="0 0 0" xyz="0.01 -0.02 0.05"/>',
                '      <mass value="%r"/>' % MASSES[name], "    </inertial>", "  </link>"]
    for name, parent, child, xyz, rpy, axis, lim in JOINTS:
        out += ['  <joint name="%s" type="revolute">' % name, '    <origin rpy="%s" xyz="%s"/>' % (num(rpy), num(xyz)),
                '    <axis xyz="%s"/>' % num(axis), '    <parent link="%s"/>' % parent, '    <child link="%s"/>' % child,
                '    <limit lower="%r" upper="%r"/>' % lim, "  </joint>"]
    name, parent, child, xyz, rpy = TOOL
    out += ['  <joint name="%s" type="fixed">' % name, '    <origin rpy="%s" xyz="%s"/>' % (num(rpy), num(xyz)),
            '    <parent link="%s"/>' % parent, '    <child link="%s"/>' % child, "  </joint>", "</robot>"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "arm7.kin.urdf")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
