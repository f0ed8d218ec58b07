"""PyBullet environment with the reference's interface (environment/environment.py:19-517). Imported only when
`pybullet` is installed (it is a third-party simulator, absent from the build image and outside the accelerated hot
path); everything here is host glue around pybullet calls, kept so that a user of the reference can switch packages.

Behaviour restated from the reference: parameter validation messages (:60-187); GUI/DIRECT connection, gravity,
URDF/SDF loading, sphere obstacle (sphere_small.urdf x2.5) and cube target (:205-262); reset = position-control to
the (optionally randomised, Python global RNG) initial joints + 50 simulation ticks (:264-309); step = velocity
control on the involved joints, position hold on the fixed ones, one tick (:453-485); reward +250 / -1000 / -(d-0.05)
(:345-371); terminal on obstacle contact or target reached (:311-343); state = [q, qdot, ee, target, obstacle] read
from joints range(len(involved_joints)) — the reference's indexing (:442-444) is kept.
One deliberate economy: the obstacle distances, which the reference queries twice per step (reward and terminal
test), are computed once and shared.
"""
from __future__ import annotations

import random
from typing import List, Optional, Tuple

import numpy as np
import pybullet as p
import pybullet_data

from ..utils.collision_detector import CollisionDetector, CollisionObject
from ..utils.exceptions import InvalidEnvironmentParameter, InvalidManipulatorFile
from ..utils.logger import get_global_logger

logger = get_global_logger()

_NUMBER = (int, float)


def _is_list_of(value, types) -> bool:
    return isinstance(value, list) and all(isinstance(v, types) for v in value)


class EnvironmentConfiguration:
    """Validated bundle of the Environment parameters; raises InvalidEnvironmentParameter with the reference's
    messages (environment.py:60-187)."""

    # (attribute, human name, item types or None for scalars, scalar types, optional)
    _LISTS = (("fixed_joints", "Fixed Joints", int, "an integer", False),
              ("involved_joints", "Involved Joints", int, "an integer", False),
              ("target_position", "Target Position", _NUMBER, "a float", False),
              ("obstacle_position", "Obstacle Position", _NUMBER, "a float", False),
              ("initial_joint_positions", "Initial Joint Positions", _NUMBER, "a float", True),
              ("initial_positions_variation_range", "Initial Positions Variation Range", _NUMBER, "a float", True))

    def __init__(self, endeffector_index: int, fixed_joints: List[int], involved_joints: List[int],
                 target_position: List[float], obstacle_position: List[float],
                 initial_joint_positions: Optional[List[float]] = None,
                 initial_positions_variation_range: Optional[List[float]] = None, max_force: float = 200.,
                 visualize: bool = True):
        if not isinstance(endeffector_index, int):
            raise InvalidEnvironmentParameter('End Effector index received is not an integer')
        self.endeffector_index = endeffector_index
        given = dict(fixed_joints=fixed_joints, involved_joints=involved_joints, target_position=target_position,
                     obstacle_position=obstacle_position, initial_joint_positions=initial_joint_positions,
                     initial_positions_variation_range=initial_positions_variation_range)
        for attr, label, item_types, item_word, optional in self._LISTS:
            value = given[attr]
            if value is None and optional:
                setattr(self, attr, None)
                continue
            if not isinstance(value, list):
                raise InvalidEnvironmentParameter(f'{label} received is not a list')
            if not _is_list_of(value, item_types):
                raise InvalidEnvironmentParameter(f'An item inside the {label} list is not {item_word}')
            setattr(self, attr, value)
        if not isinstance(max_force, _NUMBER):
            raise InvalidEnvironmentParameter('Maximum Force value received is not a float')
        self.max_force = max_force
        if not isinstance(visualize, bool):
            raise InvalidEnvironmentParameter('Visualize value received is not a boolean')
        self.visualize = visualize


class Environment:

    TARGET_THRESHOLD = 0.05
    OBSTACLE_THRESHOLD = 0.0

    def __init__(self, manipulator_file: str, environment_config: EnvironmentConfiguration):
        cfg = environment_config
        self.manipulator_file = manipulator_file
        self.visualize = cfg.visualize
        self.physics_client = p.connect(p.GUI if cfg.visualize else p.DIRECT)
        p.setGravity(0, 0, -9.81)
        p.setRealTimeSimulation(0)
        p.setAdditionalSearchPath(pybullet_data.getDataPath())
        self.target_pos, self.obstacle_pos = cfg.target_position, cfg.obstacle_position
        self.max_force = cfg.max_force
        self.initial_joint_positions = cfg.initial_joint_positions
        self.initial_positions_variation_range = cfg.initial_positions_variation_range
        self.endeffector_index = cfg.endeffector_index
        self.fixed_joints, self.involved_joints = cfg.fixed_joints, cfg.involved_joints

        if not isinstance(manipulator_file, str):
            raise InvalidManipulatorFile('The filename provided is not a string')
        try:
            if manipulator_file.endswith('.urdf'):
                self.manipulator_uid = p.loadURDF(manipulator_file)
            elif manipulator_file.endswith('.sdf'):
                self.manipulator_uid = p.loadSDF(manipulator_file)[0]
            else:
                raise InvalidManipulatorFile('The file extension is neither .sdf nor .urdf')
        except p.error as err:
            logger.critical(err)
            raise InvalidManipulatorFile
        self.num_joints = p.getNumJoints(self.manipulator_uid)
        self.print_table([(j,) + self._joint_summary(j) for j in range(self.num_joints)])
        self.obstacle = p.loadURDF('sphere_small.urdf', basePosition=self.obstacle_pos, useFixedBase=1, globalScaling=2.5)
        self.target = p.loadURDF('cube_small.urdf', basePosition=self.target_pos, useFixedBase=1, globalScaling=1)
        self._observation_space = np.zeros((9 + 2 * len(self.involved_joints),))
        self._action_space = np.zeros((len(self.involved_joints),))

    def _joint_summary(self, joint_index: int):
        info = p.getJointInfo(self.manipulator_uid, joint_index)
        return info[1].decode("utf-8"), info[9], info[8], info[13]     # name, upper, lower, axis

    # ---- episode control -------------------------------------------------------------------------------------------
    def _initial_joint_targets(self) -> List[float]:
        base, var = self.initial_joint_positions, self.initial_positions_variation_range
        if not base and not var:
            return [0 for _ in range(self.num_joints)]
        if base and var:
            return [random.uniform(b - v, b + v) for b, v in zip(base, var)]
        if base:
            return base
        return [random.uniform(-v, v) for v in var]

    def reset(self, verbose: bool = True) -> np.ndarray:
        if verbose:
            logger.info('Resetting Environment...')
        p.resetBasePositionAndOrientation(self.manipulator_uid, [0., 0., 0.], [0., 0., 0., 1.])
        for joint_index, position in enumerate(self._initial_joint_targets()):
            p.setJointMotorControl2(self.manipulator_uid, joint_index, controlMode=p.POSITION_CONTROL, targetPosition=position)
        for _ in range(50):
            p.stepSimulation(self.physics_client)
        if verbose:
            logger.info('Environment Reset')
        return self.get_state()

    def step(self, action) -> Tuple[np.ndarray, float, int]:
        for joint_index, velocity in zip(self.involved_joints, action):
            p.setJointMotorControl2(self.manipulator_uid, joint_index, p.VELOCITY_CONTROL, targetVelocity=velocity,
                                    force=self.max_force)
        for joint_index in self.fixed_joints:
            p.setJointMotorControl2(self.manipulator_uid, joint_index, p.POSITION_CONTROL, targetPosition=0)
        p.stepSimulation(physicsClientId=self.physics_client)
        obstacle_hit = self.get_manipulator_obstacle_collisions(self.OBSTACLE_THRESHOLD)    # queried once, used twice
        reached, margin = self.get_endeffector_target_collision(self.TARGET_THRESHOLD)
        reward = self._reward(reached, margin, obstacle_hit)
        done = 1 if (obstacle_hit or reached) else 0
        return self.get_state(), reward, done

    # ---- reward / terminal -----------------------------------------------------------------------------------------
    @staticmethod
    def _reward(reached: bool, margin, obstacle_hit: bool, self_collision: bool = False) -> float:
        if reached:
            return 250
        if obstacle_hit or self_collision:
            return -1000
        return -1 * float(np.asarray(margin).reshape(-1)[0])

    def _self_collision(self) -> bool:
        return any((d < 0).any() for d in self.get_manipulator_collisions_with_itself().values())

    def get_reward(self, consider_autocollision: bool = False) -> float:
        reached, margin = self.get_endeffector_target_collision(self.TARGET_THRESHOLD)
        return self._reward(reached, margin, self.get_manipulator_obstacle_collisions(0),
                            consider_autocollision and self._self_collision())

    def is_terminal_state(self, target_threshold: float = 0.05, obstacle_threshold: float = 0.,
                          consider_autocollision: bool = False) -> int:
        if self.get_manipulator_obstacle_collisions(threshold=obstacle_threshold):
            return 1
        if self.get_endeffector_target_collision(threshold=target_threshold)[0]:
            return 1
        return 1 if (consider_autocollision and self._self_collision()) else 0

    def get_manipulator_obstacle_collisions(self, threshold: float) -> bool:
        dist = [CollisionDetector(CollisionObject(self.manipulator_uid, j), [self.obstacle]).compute_distances()[0]
                for j in range(self.num_joints)]
        return bool((np.array(dist) < threshold).any())

    def get_manipulator_collisions_with_itself(self) -> dict:
        joints = list(range(self.num_joints))
        return {f'joint_{j}': CollisionDetector(CollisionObject(self.manipulator_uid, j), [])
                .compute_collisions_in_manipulator(affected_joints=joints, max_distance=10) for j in joints}

    def get_endeffector_target_collision(self, threshold: float):
        dist = CollisionDetector(CollisionObject(self.manipulator_uid, self.endeffector_index), [self.target]).compute_distances()
        return bool((dist < threshold).any()), dist - threshold

    # ---- observation -----------------------------------------------------------------------------------------------
    def get_state(self) -> np.ndarray:
        states = [p.getJointState(self.manipulator_uid, j) for j in range(len(self.involved_joints))]
        ee = list(p.getLinkState(self.manipulator_uid, self.endeffector_index)[0])
        return np.hstack([np.array([s[0] for s in states]), np.array([s[1] for s in states]), np.array(ee),
                          np.array(self.target_pos), np.array(self.obstacle_pos)]).astype(float)

    def close(self) -> None:
        p.disconnect(self.physics_client)

    @staticmethod
    def print_table(data) -> None:
        fmt = '{:<6} {:<35} {:<15} {:<15} {:<15}'
        logger.debug(fmt.format('Index', 'Name', 'Upper Limit', 'Lower Limit', 'Axis'))
        for index, name, upper, lower, axis in data:
            logger.debug(fmt.format(index, name, upper, lower, str(axis)))

    @property
    def observation_space(self) -> np.ndarray:
        return self._observation_space

    @property
    def action_space(self) -> np.ndarray:
        return self._action_space
