"""Test double for the `pybullet` module (NOT product code): a kinematic arm with the same chain as
robotic_manipulator_rloa_amd/environment/synthetic.py, exposing just the calls the Environment adapter makes.
Lets the CPU suite drive environment/environment.py end to end on an image without PyBullet."""
import types

import numpy as np


def make_modules(n_joints=8):
    from robotic_manipulator_rloa_amd.environment.synthetic import LINKS, DT
    pb = types.ModuleType("pybullet")
    pb.error = type("error", (Exception,), {})
    pb.GUI, pb.DIRECT, pb.POSITION_CONTROL, pb.VELOCITY_CONTROL = 1, 2, 2, 0
    st = {"q": np.zeros(n_joints, np.float32), "qd": np.zeros(n_joints, np.float32), "cmd": {}, "bodies": {}, "next": 0,
          "calls": []}
    pb._state = st

    def _new_body(kind, pos):
        uid = st["next"]
        st["next"] += 1
        st["bodies"][uid] = (kind, np.array(pos, np.float32))
        return uid

    def link_positions():
        R, p = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        out = []
        for k in range(n_joints):
            c, s = np.float32(np.cos(st["q"][k])), np.float32(np.sin(st["q"][k]))
            N = np.empty_like(R)
            if k % 2 == 0:
                N[:, 0], N[:, 1], N[:, 2] = R[:, 0] * c + R[:, 1] * s, -R[:, 0] * s + R[:, 1] * c, R[:, 2]
            else:
                N[:, 0], N[:, 1], N[:, 2] = R[:, 0] * c - R[:, 2] * s, R[:, 1], R[:, 0] * s + R[:, 2] * c
            R = N
            p = p + R[:, 2] * (LINKS[k] if k < len(LINKS) else np.float32(0.03))   # gripper-ish stubs beyond the arm
            out.append(p.copy())
        return out

    pb.connect = lambda mode: (st["calls"].append(("connect", mode)) or 7)
    pb.disconnect = lambda cid: st["calls"].append(("disconnect", cid))
    pb.setGravity = lambda *a: None
    pb.setRealTimeSimulation = lambda *a: None
    pb.setAdditionalSearchPath = lambda *a: None

    def loadURDF(path, basePosition=(0, 0, 0), useFixedBase=0, globalScaling=1):
        if "broken" in path:
            raise pb.error("cannot load")
        return _new_body(path, basePosition)
    pb.loadURDF = loadURDF
    pb.loadSDF = lambda path: [loadURDF(path)]
    pb.getNumJoints = lambda uid: n_joints
    pb.getJointInfo = lambda uid, j: (j, f"joint_{j}".encode(), 0, 0, 0, 0, 0, 0, -3.0, 3.0, 0, 0, b"link", (0, 0, 1))
    pb.resetBasePositionAndOrientation = lambda *a: None

    def setJointMotorControl2(uid, j, controlMode=None, targetPosition=None, targetVelocity=None, force=None):
        st["cmd"][j] = (controlMode, targetPosition if targetPosition is not None else targetVelocity)
    pb.setJointMotorControl2 = setJointMotorControl2

    def stepSimulation(*a, **k):
        for j, (mode, val) in st["cmd"].items():
            if mode == pb.POSITION_CONTROL:
                st["q"][j], st["qd"][j] = val, 0.0
            else:
                st["q"][j] = np.float32(st["q"][j] + DT * np.float32(val))
                st["qd"][j] = val
    pb.stepSimulation = stepSimulation
    pb.getJointState = lambda uid, j: (float(st["q"][j]), float(st["qd"][j]), (0,) * 6, 0.0)
    pb.getLinkState = lambda uid, link: (tuple(float(v) for v in link_positions()[link]),)

    def getClosestPoints(a, b, distance=10.0, linkIndexA=-1, linkIndexB=-1):
        pa = link_positions()[linkIndexA]
        if a == b:
            d = float(np.linalg.norm(pa - link_positions()[linkIndexB]))
        else:
            kind, pos = st["bodies"][b]
            radius = 0.06 if "sphere" in kind else 0.0
            d = float(np.linalg.norm(pa - pos)) - radius
        return [] if d > distance else [(0,) * 8 + (d,)]
    pb.getClosestPoints = getClosestPoints
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "/fake/pybullet_data"
    return pb, pbd
