#!/usr/bin/env python3
"""Writes fixtures/models/ur10.kin.urdf: a UR10 arm authored from the public `ur_description` kinematic constants.

NOT a file of the reference (the reference ships ik/test/ur5.urdf only; BASELINE.json's config 5 names a UR10, SURVEY.md 8d
lists the constants).  Same joint layout as a UR5 -- six revolute joints, axes Z Y Y Y Z Y, the two pi/2 pitch offsets --
with the UR10's lengths:

    d1 = 0.1273   a2 = -0.612   a3 = -0.5723   d4 = 0.163941   d5 = 0.1157   d6 = 0.0922
    shoulder offset 0.220941, elbow offset -0.1719  =>  wrist-1 length d4 - elbow - shoulder = 0.1149

Zero pose of tool0 (the known-answer test in tests/test_ur10.py):  x = -a2 - a3 = 1.1843,  y = d4 + d6 = 0.256141,
z = d1 - d5 = 0.0116.

    python fixtures/make_ur10_urdf.py
"""
import os

D1, A2, A3, D4, D5, D6 = 0.1273, -0.612, -0.5723, 0.163941, 0.1157, 0.0922
SHOULDER_OFFSET, ELBOW_OFFSET = 0.220941, -0.1719
WRIST_1 = D4 - ELBOW_OFFSET - SHOULDER_OFFSET
PI = 3.14159265359

# (link, mass, centre of mass): the public ur_description inertial parameters of the UR10
LINKS = [("base_link", 4.0, (0, 0, 0)), ("shoulder_link", 7.778, (0, 0, 0)), ("upper_arm_link", 12.93, (0, 0, 0.306)),
         ("forearm_link", 3.87, (0, 0, 0.28615)), ("wrist_1_link", 1.96, (0, 0, 0)), ("wrist_2_link", 1.96, (0, 0, 0)),
         ("wrist_3_link", 0.202, (0, 0, 0)), ("ee_link", 0, (0, 0, 0)), ("base", 0, (0, 0, 0)), ("tool0", 0, (0, 0, 0)),
         ("world", 0, (0, 0, 0))]
# (name, type, parent, child, xyz, rpy, axis, (lower, upper))
JOINTS = [
    ("shoulder_pan_joint", "revolute", "base_link", "shoulder_link", (0, 0, D1), (0, 0, 0), (0, 0, 1), (-2 * PI, 2 * PI)),
    ("shoulder_lift_joint", "revolute", "shoulder_link", "upper_arm_link", (0, SHOULDER_OFFSET, 0), (0, PI / 2, 0), (0, 1, 0), (-2 * PI, 2 * PI)),
    ("elbow_joint", "revolute", "upper_arm_link", "forearm_link", (0, ELBOW_OFFSET, -A2), (0, 0, 0), (0, 1, 0), (-PI, PI)),
    ("wrist_1_joint", "revolute", "forearm_link", "wrist_1_link", (0, 0, -A3), (0, PI / 2, 0), (0, 1, 0), (-2 * PI, 2 * PI)),
    ("wrist_2_joint", "revolute", "wrist_1_link", "wrist_2_link", (0, WRIST_1, 0), (0, 0, 0), (0, 0, 1), (-2 * PI, 2 * PI)),
    ("wrist_3_joint", "revolute", "wrist_2_link", "wrist_3_link", (0, 0, D5), (0, 0, 0), (0, 1, 0), (-2 * PI, 2 * PI)),
    ("ee_fixed_joint", "fixed", "wrist_3_link", "ee_link", (0, D6, 0), (0, 0, PI / 2), None, None),
    ("base_link-base_fixed_joint", "fixed", "base_link", "base", (0, 0, 0), (0, 0, -PI), None, None),
    ("wrist_3_link-tool0_fixed_joint", "fixed", "wrist_3_link", "tool0", (0, D6, 0), (-PI / 2, 0, 0), None, None),
    ("world_joint", "fixed", "world", "base_link", (0, 0, 0), (0, 0, 0), None, None),
]


def num(v):
    return " ".join(repr(float(x)) for x in v)


def main():
    out = ['<?xml version="1.0"?>',
           "<!-- UR10 kinematic fixture authored from the public ur_description constants by fixtures/make_ur10_urdf.py.",
           "     NOT a file of the reference (which ships a UR5 only); used for BASELINE.json's config 5. -->",
           '<robot name="ur10">']
    for name, mass, com in LINKS:
        out += ['  <link name="%s">' % name, "    <inertial>", '      <origin rpy="0 0 0" xyz="%s"/>' % num(com),
                '      <mass value="%r"/>' % float(mass), "    </inertial>", "  </link>"]
    for name, typ, parent, child, xyz, rpy, axis, lim in JOINTS:
        out += ['  <joint name="%s" type="%s">' % (name, typ), '    <origin rpy="%s" xyz="%s"/>' % (num(rpy), num(xyz))]
        if axis:
            out.append('    <axis xyz="%d %d %d"/>' % axis)
        out += ['    <parent link="%s"/>' % parent, '    <child link="%s"/>' % child]
        if lim:
            out.append('    <limit lower="%r" upper="%r"/>' % lim)
        out.append("  </joint>")
    out.append("</robot>")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "ur10.kin.urdf")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
