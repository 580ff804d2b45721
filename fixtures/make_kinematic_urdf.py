#!/usr/bin/env python3
"""Derive kinematic-only model fixtures from the reference's robot descriptions.

Reads the three URDFs and the SRDF that ship with the reference (as *data*) and writes
stripped files that keep only what the IK hot path consumes: link names, and for every
robot-level joint its name / type / origin / axis / parent / child / position limits.
Of each link's <inertial> only the mass and the centre of mass (<origin>) are kept, for the centre-of-mass task;
visual, collision, rotational inertia, transmission and gazebo elements are dropped.  Attribute
strings are copied verbatim so that number parsing is bit-identical to parsing the
originals.  Run in the build container only (the reference does not travel):

    python fixtures/make_kinematic_urdf.py /root/reference fixtures/models
"""
import sys
import os
import xml.etree.ElementTree as ET

SOURCES = [
    ("cassie-description/urdf/cassie.urdf", "cassie.kin.urdf"),
    ("cassie-description/urdf/cassie_fixed.urdf", "cassie_fixed.kin.urdf"),
    ("ik/test/ur5.urdf", "ur5.kin.urdf"),
]
SRDF = ("cassie-description/srdf/cassie.srdf", "cassie.nominal.json")


def strip(src_path, dst_path, rel):
    root = ET.parse(src_path).getroot()
    out = []
    out.append('<?xml version="1.0"?>')
    out.append("<!-- kinematic-only fixture derived from the reference data file %s" % rel)
    out.append("     by fixtures/make_kinematic_urdf.py: links (names, mass, centre of mass) and robot-level joints only. -->")
    out.append('<robot name="%s">' % root.get("name"))
    for el in root:
        if el.tag == "link":
            inertial = el.find("inertial")
            if inertial is None or inertial.find("mass") is None:
                out.append('  <link name="%s"/>' % el.get("name"))
                continue
            # mass and centre of mass only (what pinocchio::centerOfMass reads); the rotational inertia is not on the path
            out.append('  <link name="%s">' % el.get("name"))
            out.append("    <inertial>")
            org = inertial.find("origin")
            if org is not None:
                out.append("      <origin %s/>" % " ".join('%s="%s"' % (k, org.get(k)) for k in ("rpy", "xyz") if org.get(k) is not None))
            out.append('      <mass value="%s"/>' % inertial.find("mass").get("value"))
            out.append("    </inertial>")
            out.append("  </link>")
    for el in root:
        if el.tag != "joint":
            continue
        out.append('  <joint name="%s" type="%s">' % (el.get("name"), el.get("type")))
        for tag, attrs in (("origin", ("rpy", "xyz")), ("axis", ("xyz",)),
                           ("parent", ("link",)), ("child", ("link",)),
                           ("limit", ("lower", "upper"))):
            sub = el.find(tag)
            if sub is None:
                continue
            a = " ".join('%s="%s"' % (k, sub.get(k)) for k in attrs if sub.get(k) is not None)
            out.append("    <%s %s/>" % (tag, a))
        out.append("  </joint>")
    out.append("</robot>")
    with open(dst_path, "w") as f:
        f.write("\n".join(out) + "\n")


def srdf_nominal(src_path, dst_path, rel):
    import json
    root = ET.parse(src_path).getroot()
    gs = root.find("group_state")
    vals = {j.get("name"): float(j.get("value")) for j in gs.findall("joint")}
    with open(dst_path, "w") as f:
        json.dump({"provenance": "group_state '%s' of reference data file %s" % (gs.get("name"), rel),
                   "joints": vals}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    ref, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for rel, name in SOURCES:
        strip(os.path.join(ref, rel), os.path.join(dst, name), rel)
    srdf_nominal(os.path.join(ref, SRDF[0]), os.path.join(dst, SRDF[1]), SRDF[0])
