"""ONE table of robot presets, read by the framework's demos (rl_framework._DEMO_ENVS: the PyBullet arguments),
by the on-device synthetic env (engine.DeviceEnvLoop.PRESETS) and by bench.py's --robot.

'kuka' and 'xarm6' are the reference's demo presets verbatim (rl_framework.py:547-555, :571-580; test-time variation
ranges :642-649, :669-678). 'xarm6_robot' and 'panda' are the two robots BASELINE.json configs[3] / [4] name by URDF
(`xarm/xarm6_robot.urdf`, `franka_panda/panda.urdf` of pybullet_data): the reference has no preset for them, so joint
indices follow the URDFs as pybullet's own examples use them (xarm: fixed world joint 0, revolute joints 1-6, link6 = 6;
panda: revolute joints 0-6, fixed joint 7, hand 8, fingers 9-10, grasp target 11) and target / obstacle are a reachable
pair in front of the arm. PyBullet is absent from this image: those two have only ever been driven through the
kinematic test double (tests/fake_pybullet.py) and the synthetic env.
"""
from __future__ import annotations

from typing import Dict, List

ROBOT_PRESETS: Dict[str, dict] = {
    'kuka': dict(
        manipulator_file='kuka_iiwa/kuka_with_gripper2.sdf', endeffector_index=13,
        fixed_joints=[6, 7, 8, 9, 10, 11, 12, 13], involved_joints=[0, 1, 2, 3, 4, 5],
        target_position=[0.4, 0.85, 0.71], obstacle_position=[0.45, 0.55, 0.55],
        initial_joint_positions=[0.9, 0.45, 0, 0, 0, 0],
        training_variation=[0, 0, 0, 0, 0, 0], testing_variation=[0, 0, .5, .5, .5, .5], visualize_testing=False),
    'xarm6': dict(
        manipulator_file='xarm/xarm6_with_gripper.urdf', endeffector_index=12,
        fixed_joints=[0, 7, 8, 9, 10, 11, 12, 13], involved_joints=[1, 2, 3, 4, 5, 6],
        target_position=[0.3, 0.47, 0.61], obstacle_position=[0.25, 0.27, 0.5],
        initial_joint_positions=[0., 1., 0., -2.3, 0., 0., 0.],
        training_variation=[0, 0, 0, 0.3, 1, 1, 1], testing_variation=[0, 0, 0, 0.3, 1, 1, 1], visualize_testing=True),
    'xarm6_robot': dict(     # BASELINE configs[3]: the bare arm, same workspace as the reference's xarm6 demo
        manipulator_file='xarm/xarm6_robot.urdf', endeffector_index=6,
        fixed_joints=[0], involved_joints=[1, 2, 3, 4, 5, 6],
        target_position=[0.3, 0.47, 0.61], obstacle_position=[0.25, 0.27, 0.5],
        initial_joint_positions=[0., 1., 0., -2.3, 0., 0., 0.],
        training_variation=[0, 0, 0, 0.3, 1, 1, 1], testing_variation=[0, 0, 0, 0.3, 1, 1, 1], visualize_testing=True),
    'panda': dict(           # BASELINE configs[4]: 7 involved joints -> 7 x 7 L / P tiles, state size 23
        manipulator_file='franka_panda/panda.urdf', endeffector_index=11,
        fixed_joints=[7, 8, 9, 10, 11], involved_joints=[0, 1, 2, 3, 4, 5, 6],
        target_position=[0.45, 0.3, 0.6], obstacle_position=[0.35, 0.2, 0.45],
        initial_joint_positions=[0.0, -0.6, 0.0, -2.0, 0.0, 1.6, 0.8],
        training_variation=[0, 0, 0, 0.3, 0.5, 0.5, 0.5], testing_variation=[0, 0, 0, 0.3, 0.5, 0.5, 0.5],
        visualize_testing=True),
}

_PYBULLET_KEYS = ('manipulator_file', 'endeffector_index', 'fixed_joints', 'involved_joints', 'target_position',
                  'obstacle_position', 'initial_joint_positions')


def pybullet_arguments(robot: str) -> dict:
    """Keyword arguments of ManipulatorFramework.initialize_environment for `robot` (manipulator_file still relative to
    pybullet_data.getDataPath())."""
    p = ROBOT_PRESETS[robot]
    return {k: (list(p[k]) if isinstance(p[k], list) else p[k]) for k in _PYBULLET_KEYS}


def action_size(robot: str) -> int:
    return len(ROBOT_PRESETS[robot]['involved_joints'])


def synthetic_initial_joints(robot: str) -> List[float]:
    """Initial value of the A joint-position slots of the state vector. Environment.reset applies value k to joint index k
    (environment.py:284-293) and get_state reports joints 0 .. A-1 (environment.py:442-444, whatever `involved_joints`
    says — for the xarm presets that is the fixed world joint plus joints 1-5), so the reference's observation starts
    at initial_joint_positions[:A]; the synthetic stand-in starts its chain there too."""
    p = ROBOT_PRESETS[robot]
    init = [float(x) for x in p['initial_joint_positions']]
    A = len(p['involved_joints'])
    return (init + [0.0] * A)[:A]


def device_env_preset(robot: str) -> List[float]:
    """[initial joint positions (8, zero padded) | target xyz | obstacle xyz] as csrc/synth_env.hip takes it."""
    p = ROBOT_PRESETS[robot]
    q = synthetic_initial_joints(robot)
    return (q + [0.0] * 8)[:8] + [float(x) for x in p['target_position']] + [float(x) for x in p['obstacle_position']]
