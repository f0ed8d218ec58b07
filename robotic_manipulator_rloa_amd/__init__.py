"""MI355X-native NAF training hot path behind the robotic_manipulator_rloa API.

    from robotic_manipulator_rloa_amd import ManipulatorFramework     # reference: robotic_manipulator_rloa/__init__.py:1
"""
from .rl_framework import ManipulatorFramework  # noqa: F401
