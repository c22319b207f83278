// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow).  DBoW2 is a third-party dependency that is not under /root/reference;
// the matchers use only the TYPE of DBoW2::FeatureVector: std::map<NodeId, std::vector<unsigned int>> (DBoW2 FeatureVector.h).
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
class BowVector : public std::map<WordId, WordValue> {};
}  // namespace DBoW2
