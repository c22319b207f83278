// oracle/_ref shim: orb_params has a YAML constructor; _ref only uses the explicit one, so a node that always
// returns the caller's default is enough to compile orb_params.cc unmodified
#pragma once
#include <list>
#include <map>
#include <string>
#include <vector>
namespace YAML {
struct Node {
    Node operator[](const std::string&) const { return Node(); }
    template <typename T> T as(const T& fallback) const { return fallback; }
    template <typename T> T as() const { return T(); }
};
}
