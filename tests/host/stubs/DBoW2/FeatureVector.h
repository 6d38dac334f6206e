// TEST STAND-IN for DBoW2/DBoW2/FeatureVector.h:24-46 (a std::map<NodeId, std::vector<unsigned>> with addFeature).
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
class FeatureVector : public std::map<unsigned, std::vector<unsigned>> {
 public:
  void addFeature(unsigned id, unsigned i_feature) { (*this)[id].push_back(i_feature); }
};
}  // namespace DBoW2
