// Shared between the host-only part (capi.cpp) and the device part (solver.hip) of libbpmpc.so.
#pragma once
#include <string>

#include "../../include/bpmpc.h"
#include "robot_model.h"

namespace bpmpc {
void set_last_error(const std::string& message);
const RobotModel& model_of(const bpmpc_model* handle);
}  // namespace bpmpc
