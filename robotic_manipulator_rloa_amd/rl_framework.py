"""ManipulatorFramework: the user-facing object of the reference (rl_framework.py:47-699) re-hosted on the
MI355X hot path. Same method names, argument meaning, validation rules and exceptions; no compute happens
here — everything numeric goes through NAFAgent -> libnaf_hip.so.

Differences, all opt-in or forced by the image:
  * initialize_environment() needs PyBullet (absent from this image): it raises InvalidManipulatorFile with an
    explanatory message if `pybullet` cannot be imported; initialize_synthetic_environment() builds the
    kinematic stand-in (environment/synthetic.py) so the rest of the API works everywhere.
  * the demos take `interactive=False` to skip the reference's input() prompts (rl_framework.py:525,535) and
    `environment='synthetic'` to run without PyBullet.
  * p_mode / action_mode / use_graph can be passed to initialize_naf_agent(); defaults = reference semantics.
  * many-env training and evaluation: initialize_naf_agent(..., n_envs=E), run_training(...,
    n_envs=E) and test_trained_model(..., n_envs=E) run E copies of the configured environment — on the device for the
    synthetic stand-in, in E worker processes for anything else (PyBullet) — and produce what the one-env calls
    produce: the {episode: (score, last_frame)} dict, checkpoints/{episode}/weights.p + scores.txt, model.p, the logged
    test summary. Without n_envs every call is the reference's one-env loop, signature and behaviour untouched.
"""
from __future__ import annotations

import functools
import json
import os
import re
from dataclasses import dataclass, fields
from typing import List, Optional, Union

import numpy as np
import torch

from .environment.synthetic import SyntheticEnvironment
from .naf_components.naf_algorithm import NAFAgent
from .presets import ROBOT_PRESETS, pybullet_arguments, synthetic_initial_joints
from .utils.exceptions import (ConfigurationIncomplete, EnvironmentNotInitialized, InvalidHyperParameter,
                               InvalidManipulatorFile, InvalidNAFAgentParameter, NAFAgentNotInitialized)
from .utils.logger import Logger, get_global_logger

logger = get_global_logger()
Logger.set_logger_setup()


@dataclass
class HyperParameters:
    """Defaults of the reference (rl_framework.py:33-44)."""
    buffer_size: int = 100000
    batch_size: int = 128
    gamma: float = 0.99
    tau: float = 0.001
    learning_rate: float = 0.001
    update_freq: int = 1
    num_updates: int = 1


def _positive_int(v) -> bool:
    return isinstance(v, int) and v > 0


# accepted spellings -> (field, validity predicate, error text) — the reference's table (rl_framework.py:179-230),
# including its copy-pasted message for num_update
_HYPERPARAMETER_RULES = (
    (r'^(buffer_size|buffersize|BUFFER_SIZE|BUFFERSIZE)$', 'buffer_size', _positive_int,
     'Buffer Size is not an int or has a value lower than 0'),
    (r'^(batch_size|batchsize|BATCH_SIZE|BATCHSIZE)$', 'batch_size', _positive_int,
     'Batch Size is not an int or has a value lower than 0'),
    (r'^(gamma|GAMMA)$', 'gamma', lambda v: isinstance(v, (int, float)) and 0 < v < 1,
     'Gamma is not a float or its value is out of range (0, 1)'),
    (r'^(tau|TAU)$', 'tau', lambda v: isinstance(v, (int, float)) and 0 <= v <= 1,
     'Tau is not a float or its value is out of range [0, 1]'),
    (r'^(learning_rate|learningrate|LEARNING_RATE|LEARNINGRATE)$', 'learning_rate',
     lambda v: isinstance(v, (int, float)) and v > 0, 'Learning Rate is not a float or has a value lower than 0'),
    (r'^(update_freq|updatefreq|UPDATE_FREQ|UPDATEFREQ)$', 'update_freq', _positive_int,
     'Update Frequency is not an int or has a value lower than 0'),
    (r'^(num_update|numupdate|NUMUPDATE|NUM_UPDATE)$', 'num_updates', _positive_int,
     'Buffer Size is not an int or has a value lower than 0'),
)

# presets of run_demo_training / run_demo_testing (rl_framework.py:547-555, :571-580, :642-649, :669-678) plus the two
# robots BASELINE configs[3] / [4] name: one table, shared with the device env (presets.py)
_DEMO_ENVS = {name: pybullet_arguments(name) for name in ROBOT_PRESETS}


def _build_environment(manipulator_file: str, config_kwargs: dict):
    """Picklable factory of one PyBullet Environment (a worker process of environment.vector_env.HostVectorEnv calls it;
    workers never open a GUI: DIRECT mode, as north_star's "N independent PyBullet DIRECT envs per GPU")."""
    from .environment.environment import Environment, EnvironmentConfiguration
    return Environment(manipulator_file=manipulator_file,
                       environment_config=EnvironmentConfiguration(**dict(config_kwargs, visualize=False)))


def _build_synthetic(n_joints, target, obstacle, init, variation):
    return SyntheticEnvironment(n_joints, target, obstacle, init, variation)


class ManipulatorFramework:

    def __init__(self) -> None:
        self.env = None
        self.naf_agent: Optional[NAFAgent] = None
        self._env_factory = None          # picklable zero-argument factory of a copy of self.env (many-env paths)
        self._n_envs: Optional[int] = None
        self._hyperparameters: Optional[HyperParameters] = None
        self._initialize_hyperparameters()
        logger.info('The Framework has been initialized with the default hyperparameters configuration')

    def _initialize_hyperparameters(self) -> None:
        self._hyperparameters = HyperParameters()

    # ---- logging / info ------------------------------------------------------------------------------------------
    @staticmethod
    def set_log_level(log_level: int) -> None:
        names = {10: 'DEBUG', 20: 'INFO', 30: 'WARNING', 40: 'ERROR', 50: 'CRITICAL'}
        if log_level in names:
            logger.setLevel(log_level)
            logger.info(f'Log Level has been set to {log_level} ({names[log_level]})')
        else:
            logger.error(f'The Log level provided is invalid, so the previous Log Level is maintained ({logger.level}))')
            logger.error('Valid values: 10 (DEBUG), 20 (INFO), 30 (WARNING), 40 (ERROR), 50 (CRITICAL)')

    @staticmethod
    def get_required_hyperparameters() -> None:
        if logger.level > 10:
            logger.error('get_required_hyperparameters() only shows information for DEBUG log level. '
                         'Try running this method after setting the log level to DEBUG by calling '
                         'set_log_level(10) class method')
            return
        logger.debug('Required Hyperparameters:')
        for f in fields(HyperParameters):
            logger.debug('{:<25} default {}'.format(f.name, f.default))

    @staticmethod
    def plot_training_rewards(episode: int, mean_range: int = 50) -> None:
        """Mean reward per block of `mean_range` episodes from checkpoints/{episode}/scores.txt (rl_framework.py:124-157)."""
        try:
            with open(f'checkpoints/{episode}/scores.txt', 'r') as f:
                scores = json.loads(f.read())
        except FileNotFoundError as err:
            logger.error(f'File "scores.txt" located in checkpoints/{episode}/ folder was not found')
            raise err
        rewards = [result[0] for result in scores.values()]
        means = [sum(rewards[i:i + mean_range]) / mean_range for i in range(0, len(rewards) - mean_range + 1, mean_range)]
        import matplotlib.pyplot as plt   # optional dependency, only needed for this plot
        plt.plot(range(len(means)), means)
        plt.show()

    # ---- hyper-parameters ---------------------------------------------------------------------------------------
    def set_hyperparameter(self, hyperparameter: str, value: Union[float, int]) -> None:
        for pattern, field, valid, error in _HYPERPARAMETER_RULES:
            if re.match(pattern, hyperparameter):
                if not valid(value):
                    raise InvalidHyperParameter(error)
                setattr(self._hyperparameters, field, value)
                logger.info(f'Hyperparameter {hyperparameter} has been set to {value}')
                return
        raise InvalidHyperParameter(
            'The hyperparameter name passed as parameter is not valid. Valid hyperparameters are: '
            '["buffer_size", "batch_size", "gamma", "tau", "learning_rate", "update_freq", "num_update"]')

    # ---- pretrained weights -------------------------------------------------------------------------------------
    def _require_env_and_agent(self) -> None:
        if not self.env:
            raise EnvironmentNotInitialized
        if not self.naf_agent:
            raise NAFAgentNotInitialized

    def load_pretrained_parameters_from_weights_file(self, parameters_file_path: str) -> None:
        self._require_env_and_agent()
        self.naf_agent.initialize_pretrained_agent_from_weights_file(parameters_file_path)

    def load_pretrained_parameters_from_episode(self, episode: int) -> None:
        self._require_env_and_agent()
        self.naf_agent.initialize_pretrained_agent_from_episode(episode)

    # ---- configuration dumps ------------------------------------------------------------------------------------
    def get_environment_configuration(self) -> None:
        if not self.env:
            logger.error("Environment is not initialized yet, can't show configuration")
            return
        logger.info('Environment Configuration:')
        for label, attr in (('Manipulator File', 'manipulator_file'), ('End Effector index', 'endeffector_index'),
                            ('List of fixed Joints', 'fixed_joints'), ('List of Joints involved in training', 'involved_joints'),
                            ('Position of the Target', 'target_pos'), ('Position of the Obstacle', 'obstacle_pos'),
                            ('Initial position of joints', 'initial_joint_positions'),
                            ('Initial variation range of joints', 'initial_positions_variation_range'),
                            ('Max Force to be applied on joints', 'max_force'), ('Visualize mode', 'visualize')):
            logger.info('* {:<38} {}'.format(label + ':', getattr(self.env, attr, 'n/a')))
        logger.info(f'* Instance of the Environment:         {self.env}')

    def get_nafagent_configuration(self) -> None:
        if not self.naf_agent:
            logger.error("NAFAgent is not initialized yet, can't show configuration")
            return
        logger.info('NAFAgent Configuration:')
        for label, attr in (('Environment Instance', 'environment'), ('State Size', 'state_size'),
                            ('Action Size', 'action_size'), ('Size of layers of the Neural Network', 'layer_size'),
                            ('Batch Size', 'batch_size'), ('Buffer Size', 'buffer_size'), ('Learning Rate', 'learning_rate'),
                            ('Tau', 'tau'), ('Gamma', 'gamma'), ('Update Frequency', 'update_freq'),
                            ('Number of Updates', 'num_updates'), ('Checkpoint frequency', 'checkpoint_frequency'),
                            ('Device', 'device')):
            logger.info('* {:<40} {}'.format(label + ':', getattr(self.naf_agent, attr)))

    # ---- evaluation -----------------------------------------------------------------------------------------------
    def test_trained_model(self, n_episodes: int, frames: int, n_envs: Optional[int] = None) -> dict:
        """n_episodes test episodes of at most `frames` steps; success iff done with reward == 250
        (rl_framework.py:319-367). Also returns the summary it logs. n_envs=E (or the n_envs given to
        initialize_naf_agent): the episodes are spread over E copies of the environment and every vector step is one
        batched act() — same result rule, same log lines."""
        if not self.naf_agent or not self.env:
            raise ConfigurationIncomplete
        results, num_collisions = [], 0
        E = n_envs if n_envs is not None else self._n_envs
        if E is not None and E > 1:
            triples = self._many_env_results(n_episodes, frames, int(E))
            results = [(ok, frame) for ok, frame, _ in triples]
            num_collisions = sum(1 for ok, _, done in triples if done and not ok)
            for ep in range(len(results)):
                logger.info('Test Episode number {ep} completed\n'.format(ep=ep + 1))
            n_episodes = 0
        for ep in range(n_episodes):
            state = self.env.reset()
            for frame in range(frames):
                action = self.naf_agent.act(state)
                state, reward, done = self.env.step(action)
                if done:
                    results.append((reward == 250, frame))
                    num_collisions += int(reward != 250)
                    break
                if frame == frames - 1:
                    results.append((False, frame))
            logger.info('Test Episode number {ep} completed\n'.format(ep=ep + 1))
        logger.info('RESULTS OF THE TEST:')
        for i, (ok, frame) in enumerate(results):
            logger.info(f'Results of Iteration {i + 1}: COMPLETED: {ok}. FRAMES: {frame}')
        wins = [f for ok, f in results if ok]
        summary = {'successes': len(wins), 'episodes': len(results), 'collisions': num_collisions,
                   'mean_frames_to_success': float(np.mean(wins)) if wins else float('nan')}
        logger.info(f'Number of successful executions: {len(wins)}/{len(results)}  '
                    f'({100.0 * len(wins) / max(1, len(results))}%)')
        logger.info(f'Average number of frames required to complete an episode: {summary["mean_frames_to_success"]}')
        logger.info(f'Number of episodes terminated because of collisions: {num_collisions}')
        return summary

    # ---- environment ------------------------------------------------------------------------------------------------
    def initialize_environment(self, manipulator_file: str, endeffector_index: int, fixed_joints: List[int],
                               involved_joints: List[int], target_position: List[float], obstacle_position: List[float],
                               initial_joint_positions: List[float] = None,
                               initial_positions_variation_range: List[float] = None, max_force: float = 200.,
                               visualize: bool = True) -> None:
        """PyBullet environment (rl_framework.py:369-417). The simulator is third-party and not part of this build."""
        try:
            from .environment.environment import Environment, EnvironmentConfiguration
        except ImportError as e:
            raise InvalidManipulatorFile(
                f'PyBullet is not importable here ({e}); use initialize_synthetic_environment() for the built-in '
                f'kinematic stand-in, or install pybullet to load {manipulator_file}') from e
        config = EnvironmentConfiguration(
            endeffector_index=endeffector_index, fixed_joints=fixed_joints, involved_joints=involved_joints,
            target_position=target_position, obstacle_position=obstacle_position,
            initial_joint_positions=initial_joint_positions,
            initial_positions_variation_range=initial_positions_variation_range, max_force=max_force, visualize=visualize)
        self.env = Environment(manipulator_file=manipulator_file, environment_config=config)
        self._env_factory = functools.partial(_build_environment, manipulator_file, dict(
            endeffector_index=endeffector_index, fixed_joints=fixed_joints, involved_joints=involved_joints,
            target_position=target_position, obstacle_position=obstacle_position,
            initial_joint_positions=initial_joint_positions,
            initial_positions_variation_range=initial_positions_variation_range, max_force=max_force))
        logger.info('Pybullet Environment successfully initialized')

    def initialize_synthetic_environment(self, n_joints: int = 6, target_position: List[float] = None,
                                         obstacle_position: List[float] = None, initial_joint_positions: List[float] = None,
                                         initial_positions_variation_range: List[float] = None,
                                         obstacle_jitter: float = 0.0) -> None:
        """obstacle_jitter (many-env runs on the device only): each env's obstacle sits at obstacle_position + U(-j, j)^3,
        drawn once per (rank, env) — BASELINE configs[3]'s "randomized obstacle_position per env"."""
        self.env = SyntheticEnvironment(n_joints, target_position, obstacle_position, initial_joint_positions,
                                        initial_positions_variation_range)
        self._obstacle_jitter = float(obstacle_jitter)
        self._env_factory = functools.partial(_build_synthetic, n_joints, target_position, obstacle_position,
                                              initial_joint_positions, initial_positions_variation_range)
        logger.info('Synthetic (kinematic stand-in) Environment successfully initialized')

    def delete_environment(self) -> None:
        if not self.env:
            logger.error('No existing instance of Environment found')
            return
        close = getattr(self.env, 'close', None)
        if close:
            close()
        self.env = None
        self._env_factory = None
        logger.info('Environment instance has been successfully removed')

    # ---- agent --------------------------------------------------------------------------------------------------------
    def initialize_naf_agent(self, checkpoint_frequency: int = 500, seed: int = 0, n_envs: Optional[int] = None,
                             **agent_options) -> None:
        """rl_framework.py:431-465: same guards, same NAFAgent keyword arguments (layer_size is 256, :452).
        The device is cuda:0 — this build has no CPU path, so a missing GPU is an error, not a silent fallback.
        n_envs=E: run_training / test_trained_model of this agent use E copies of the environment (see the module text)."""
        if not self.env:
            raise EnvironmentNotInitialized
        if not isinstance(checkpoint_frequency, int) or not isinstance(seed, int):
            raise InvalidNAFAgentParameter('Checkpoint Frequency or Seed received is not an integer')
        if n_envs is not None and (not isinstance(n_envs, int) or n_envs < 1):
            raise InvalidNAFAgentParameter('Number of environments received is not a positive integer')
        self._n_envs = n_envs
        hp = self._hyperparameters
        from .parallel import local_device
        device = local_device()              # cuda:LOCAL_RANK — one process per GPU under torch.distributed.run
        self.naf_agent = NAFAgent(environment=self.env,
                                  state_size=self.env.observation_space.shape[0],
                                  action_size=self.env.action_space.shape[0],
                                  layer_size=256,
                                  batch_size=hp.batch_size, buffer_size=hp.buffer_size, learning_rate=hp.learning_rate,
                                  tau=hp.tau, gamma=hp.gamma, update_freq=hp.update_freq, num_updates=hp.num_updates,
                                  checkpoint_frequency=checkpoint_frequency, device=device, seed=seed, **agent_options)
        logger.info('NAF Agent successfully initialized')

    def delete_naf_agent(self) -> None:
        if not self.naf_agent:
            logger.error('No existing instance of NAFAgent found')
            return
        self.naf_agent = None
        self._n_envs = None
        logger.info('NAFAgent instance has been successfully removed')

    # ---- training ---------------------------------------------------------------------------------------------------
    def run_training(self, episodes: int, frames: Optional[int] = 500, verbose: bool = True, n_envs: Optional[int] = None):
        """rl_framework.py:478-501 -> NAFAgent.run(frames, episodes, verbose): {episode: (score, last_frame)}, checkpoints,
        model.p. n_envs=E (or the n_envs given to initialize_naf_agent): the same outputs from E environments at once —
        episodes numbered in completion order, `frames` the budget of each; counters in naf_agent.last_run_stats."""
        if not self.naf_agent or not self.env:
            raise ConfigurationIncomplete
        E = n_envs if n_envs is not None else self._n_envs
        if E is None or E <= 1:
            return self.naf_agent.run(frames, episodes, verbose)
        if isinstance(self.env, SyntheticEnvironment):
            return self.naf_agent.run_vectorized(episodes=episodes, n_envs=int(E), max_frames=frames, verbose=verbose,
                                                 **self._device_env_arguments())['scores']
        vec = self._host_vector_env(int(E), frames)
        try:
            return self.naf_agent.run_host_vectorized(vec, episodes=episodes, verbose=verbose)['scores']
        finally:
            vec.close()

    def run_vectorized_training(self, vector_steps: int, n_envs: int = 64, max_frames: int = 400) -> dict:
        """Many-env training on the device-resident synthetic arms (BASELINE configs[1..4] shape) for a fixed number of
        vector steps; see NAFAgent.run_vectorized (counters + 'scores')."""
        if not self.naf_agent or not self.env:
            raise ConfigurationIncomplete
        kw = self._device_env_arguments() if isinstance(self.env, SyntheticEnvironment) else {}
        return self.naf_agent.run_vectorized(vector_steps, n_envs=n_envs, max_frames=max_frames, **kw)

    # ---- E copies of the configured environment -------------------------------------------------------------------
    def _device_env_arguments(self) -> dict:
        """The synthetic environment's configuration as csrc/synth_env.hip takes it."""
        env = self.env
        pad8 = lambda v: ([float(x) for x in v] + [0.0] * 8)[:8]          # noqa: E731
        var = env.initial_positions_variation_range
        return {'preset': pad8(env.initial_joint_positions) + [float(x) for x in env.target_pos] +
                [float(x) for x in env.obstacle_pos], 'variation': pad8(var) if var is not None else [0.0] * 8,
                'obstacle_jitter': getattr(self, '_obstacle_jitter', 0.0)}

    def _host_vector_env(self, n_envs: int, frames: int):
        from .environment.vector_env import HostVectorEnv
        if self._env_factory is None:
            raise ConfigurationIncomplete('many-env runs need an environment built by initialize_environment() / '
                                          'initialize_synthetic_environment() (a factory of copies of it)')
        return HostVectorEnv(self._env_factory, n_envs, self.naf_agent.state_size, self.naf_agent.action_size,
                             max_frames=frames, seed=self.naf_agent.seed)

    def _many_env_results(self, n_episodes: int, frames: int, n_envs: int):
        if isinstance(self.env, SyntheticEnvironment):
            return self.naf_agent.evaluate_vectorized(n_episodes, frames, n_envs=n_envs, **self._device_env_arguments())
        vec = self._host_vector_env(n_envs, frames)
        try:
            return self.naf_agent.evaluate_host_vectorized(vec, n_episodes)
        finally:
            vec.close()

    # ---- demos --------------------------------------------------------------------------------------------------------
    def _clear_for_demo(self, interactive: bool) -> bool:
        for what, present, delete in (('Environment', self.env, self.delete_environment),
                                      ('NAFAgent', self.naf_agent, self.delete_naf_agent)):
            if present:
                if interactive and input(f'{what} instance found. Overwrite? [Y/n] ').lower() != 'y':
                    logger.info(f'Demo could not run due to the presence of a user-configured {what} instance')
                    return False
                delete()
        return True

    def _demo_environment(self, robot: str, environment: str, variation, visualize: bool) -> None:
        preset = dict(_DEMO_ENVS[robot])
        if environment == 'synthetic':
            n = len(preset['involved_joints'])
            self.initialize_synthetic_environment(n, preset['target_position'], preset['obstacle_position'],
                                                  synthetic_initial_joints(robot), list(variation)[:n])
        else:
            import pybullet_data
            preset['manipulator_file'] = os.path.join(pybullet_data.getDataPath(), preset['manipulator_file'])
            self.initialize_environment(initial_positions_variation_range=variation, visualize=visualize, **preset)

    def run_demo_training(self, demo_type: str, verbose: bool = False, interactive: bool = True,
                          environment: str = 'pybullet', episodes: int = 20, frames: int = 400) -> None:
        """'kuka_training' / 'xarm6_training': preset env + default agent, 20 episodes x 400 frames
        (rl_framework.py:503-598)."""
        old_level = logger.level
        logger.setLevel(10)
        try:
            robot = demo_type[:-len('_training')] if demo_type.endswith('_training') else None
            if robot not in ROBOT_PRESETS:       # the reference knows kuka_training / xarm6_training
                logger.error('Incorrect demo type!')
                return
            if not self._clear_for_demo(interactive):
                return
            variation = list(ROBOT_PRESETS[robot]['training_variation'])
            self._demo_environment(robot, environment, variation, visualize=True)
            self.initialize_naf_agent()
            self.run_training(episodes, frames, verbose=verbose)
            self.delete_environment()
            self.delete_naf_agent()
        finally:
            logger.setLevel(old_level)

    def run_demo_testing(self, demo_type: str, interactive: bool = True, environment: str = 'pybullet',
                         weights_file: Optional[str] = None, episodes: int = 50, frames: int = 750) -> Optional[dict]:
        """'kuka_testing' / 'xarm6_testing': preset env, pretrained weights, 50 x 750-frame test episodes
        (rl_framework.py:600-699). The reference ships demo weights inside its package; pass their path as
        `weights_file` (reference-format .p files load unchanged)."""
        old_level = logger.level
        logger.setLevel(10)
        try:
            robot = demo_type[:-len('_testing')] if demo_type.endswith('_testing') else None
            if robot not in ROBOT_PRESETS:       # the reference knows kuka_testing / xarm6_testing
                logger.error('Incorrect demo type!')
                return None
            if not self._clear_for_demo(interactive):
                return None
            variation = list(ROBOT_PRESETS[robot]['testing_variation'])
            self._demo_environment(robot, environment, variation, visualize=ROBOT_PRESETS[robot]['visualize_testing'])
            self.initialize_naf_agent()
            if weights_file is not None:
                self.load_pretrained_parameters_from_weights_file(weights_file)
            out = self.test_trained_model(episodes, frames)
            self.delete_environment()
            self.delete_naf_agent()
            return out
        finally:
            logger.setLevel(old_level)
