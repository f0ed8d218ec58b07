"""MI355X-native NAF training hot path behind the robotic_manipulator_rloa API."""
