#!/usr/bin/env python3
"""Extract the kinematic tree of a URDF (joint names, types, parent / child links, origins, axes) into the small JSON
model the motion converter consumes -- numbers only, no meshes, inertias or limits.

    python tools/urdf_to_kinematics.py <robot.urdf> humanoid_amp_amd/motions/models/g1_29dof.json

The shipped model was extracted from the reference's g1_model/urdf/g1_29dof_rev_1_0.urdf (Unitree G1, 29 DoF).
"""
import json
import sys
import xml.etree.ElementTree as ET


def floats(text, n):
    v = [float(x) for x in (text or "").split()]
    return v if len(v) == n else [0.0] * n


def main(src, dst):
    robot = ET.parse(src).getroot()
    joints = []
    for j in robot.findall("joint"):
        origin = j.find("origin")
        axis = j.find("axis")
        joints.append({
            "name": j.get("name"), "type": j.get("type"),
            "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": floats(origin.get("xyz") if origin is not None else "", 3),
            "rpy": floats(origin.get("rpy") if origin is not None else "", 3),
            "axis": floats(axis.get("xyz") if axis is not None else "", 3) if axis is not None else [0.0, 0.0, 0.0],
        })
    children = {j["child"] for j in joints}
    roots = sorted({j["parent"] for j in joints} - children)
    model = {"robot": robot.get("name"), "root_links": roots, "links": [l.get("name") for l in robot.findall("link")],
             "joints": joints}
    with open(dst, "w") as fh:
        json.dump(model, fh, indent=1)
    print(f"{dst}: {len(joints)} joints, {len(model['links'])} links, roots {roots}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
