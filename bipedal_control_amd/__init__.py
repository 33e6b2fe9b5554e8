"""bipedal_control_amd - MI355X-native batched NMPC engine for the bipedal_control OCS2 SQP hot path.

Python is only the host-side mirror of the reference's operator interface (ctypes over the C ABI declared in
include/bpmpc.h); all arithmetic of the hot path runs in hand-written HIP kernels (bipedal_control_amd/csrc/kernels).
There is no CPU fallback: constructing a solver without the HIP library or without a GPU raises.
"""
from .api import (BatchedDdpMpc, BatchedSqpMpc, BipedalRobotInterface, BpmpcError, GaitSchedule, ModeSchedule, ModeSequenceTemplate,  # noqa: F401
                  TargetTrajectories, WeightedWbc, load_library, loadModeSequenceTemplate, swing_reference, time_discretization_with_events)

__all__ = ["BatchedDdpMpc", "BatchedSqpMpc", "BipedalRobotInterface", "BpmpcError", "GaitSchedule", "ModeSchedule", "ModeSequenceTemplate",
           "TargetTrajectories", "WeightedWbc", "load_library", "loadModeSequenceTemplate", "swing_reference", "time_discretization_with_events"]
