// INFO-file reader (the configuration format of task.info / reference.info / gait.info).
// Mirrors the semantics the reference obtains through boost::property_tree::read_info and OCS2 loadData::*
// (call sites: ocs2_bipedal_robot/src/BipedalRobotInterface.cpp:92-108,239-291,298-315;
//  src/common/ModelSettings.cpp:40-67; src/gait/ModeSequenceTemplate.cpp:50-111).
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace bpmpc {

class InfoNode {
 public:
  std::string data;
  std::vector<std::pair<std::string, std::unique_ptr<InfoNode>>> children;

  // First child with this key at each level of a '.'-separated path (ptree::get semantics); nullptr if absent.
  const InfoNode* find(const std::string& dotted_path) const;
  bool get(const std::string& path, double* out) const;
  bool get(const std::string& path, int* out) const;
  bool get(const std::string& path, std::string* out) const;
};

// Throws std::runtime_error on unreadable files or unbalanced braces.
std::unique_ptr<InfoNode> read_info_file(const std::string& path);

// loadData::loadEigenMatrix: entries "(i,j)", optional "scaling" / "default"; row-major result.
// Throws if no entry at all is present.
std::vector<double> load_matrix(const InfoNode& root, const std::string& name, int rows, int cols);
// loadData::loadStdVector: "[0]", "[1]", ... up to the first missing index.
std::vector<double> load_scalar_list(const InfoNode& root, const std::string& name);
std::vector<std::string> load_string_list(const InfoNode& root, const std::string& name);

}  // namespace bpmpc
