// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow).  nlohmann-json is not in the build image.  data/common.cc holds the
// map-database JSON helpers (lines 33-203) in the same translation unit as the grid functions oracle/_ref pins (205-363);
// this value class has the constructors and accessors those helpers name so the FILE compiles unmodified.  The helpers
// themselves are not called by oracle/_ref (the wire format is covered on the Python side, tests/test_io_formats.py).
#pragma once
#include <initializer_list>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
namespace nlohmann {
class json {
public:
    json() = default;
    template <typename T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0> json(T v) : num_((double)v), kind_(1) {}
    json(const char* s) : str_(s), kind_(2) {}
    json(const std::string& s) : str_(s), kind_(2) {}
    json(std::initializer_list<json> l) : arr_(l), kind_(3) {}
    json(const std::vector<json>& v) : arr_(v), kind_(3) {}
    size_t size() const { return arr_.size(); }
    const json& at(size_t i) const { return arr_.at(i); }
    const json& at(int i) const { return arr_.at((size_t)i); }
    const json& at(unsigned i) const { return arr_.at((size_t)i); }
    const json& at(const char* key) const {   // an object is held as a list of {key, value} pairs
        for (const auto& kv : arr_) if (kv.arr_.size() == 2 && kv.arr_[0].kind_ == 2 && kv.arr_[0].str_ == key) return kv.arr_[1];
        throw std::out_of_range(key);
    }
    template <typename T> T get() const { return get_impl((T*)nullptr); }
private:
    template <typename T> T get_impl(T*) const { return (T)num_; }
    template <typename T> std::vector<T> get_impl(std::vector<T>*) const { std::vector<T> v; for (const auto& e : arr_) v.push_back((T)e.num_); return v; }
    double num_ = 0;
    std::string str_;
    std::vector<json> arr_;
    int kind_ = 0;
};
}  // namespace nlohmann
