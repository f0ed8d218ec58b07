"""Exception types of the framework API. Names, hierarchy and default messages are the reference's
(robotic_manipulator_rloa/utils/exceptions.py:6-135) so that callers' `except` clauses and message checks keep
working when they switch packages."""
from __future__ import annotations

from typing import Optional


class FrameworkException(Exception):
    """Base class; `message` is what str() shows after the class name."""
    message = ''

    def __init__(self, message: Optional[str] = None) -> None:
        if message:
            self.message = message
        super().__init__(self.message)

    def __str__(self) -> str:
        return f'{type(self).__name__}: {self.message}'

    def set_message(self, value: str) -> "FrameworkException":
        self.message = value
        return self


def _define(name: str, default_message: str, doc: str):
    return type(name, (FrameworkException,), {'message': default_message, '__doc__': doc, '__module__': __name__})


InvalidManipulatorFile = _define(
    'InvalidManipulatorFile', 'The URDF/SDF file received is not valid',
    'The manipulator file cannot be loaded by the simulator.')
InvalidHyperParameter = _define(
    'InvalidHyperParameter', 'The hyperparameter received is not valid',
    'set_hyperparameter() received an unknown name or an out-of-range value.')
InvalidEnvironmentParameter = _define(
    'InvalidEnvironmentParameter', 'The Environment parameter received is not valid',
    'The Environment was configured with an invalid parameter.')
InvalidNAFAgentParameter = _define(
    'InvalidNAFAgentParameter', 'The NAF Agent parameter received is not valid',
    'The NAFAgent was configured with an invalid parameter.')
EnvironmentNotInitialized = _define(
    'EnvironmentNotInitialized',
    'The Environment is not yet initialized. The environment can be initialized via the '
    'initialize_environment() method',
    'A method that needs the Environment was called before initialize_environment().')
NAFAgentNotInitialized = _define(
    'NAFAgentNotInitialized',
    'The NAF Agent is not yet initialized. The agent can be initialized via the '
    'initialize_naf_agent() method',
    'A method that needs the NAFAgent was called before initialize_naf_agent().')
MissingWeightsFile = _define(
    'MissingWeightsFile', 'The weight file provided does not exist',
    'Pretrained weights were requested from a path that does not exist.')
ConfigurationIncomplete = _define(
    'ConfigurationIncomplete',
    'The configuration for the training is incomplete. Either the Environment, the '
    'NAF Agent or both are not yet initialized. The environment can be initialized via the '
    'initialize_environment() method, and the agent can be initialized via the '
    'initialize_naf_agent() method',
    'run_training()/test_trained_model() was called without both Environment and NAFAgent.')
