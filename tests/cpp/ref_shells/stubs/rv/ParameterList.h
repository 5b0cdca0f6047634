/* TEST STUB: the slice of rv::Parameter / rv::ParameterList (src/rv/Parameter.h, ParameterList.h:50-215) that the
 * constructors and setParameters() of the hot-path classes use: named values with implicit conversions, insert,
 * hasParam, operator[], iteration.  A missing key throws, like the reference's checkParam. */
#ifndef REF_SHELLS_STUB_RV_PARAMETERLIST_H_
#define REF_SHELLS_STUB_RV_PARAMETERLIST_H_
#include <cstdint>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
namespace rv {
class Parameter {
 public:
  Parameter() {}
  Parameter(const std::string& n, const std::string& text, double num, bool is_string)
      : name_(n), text_(text), num_(num), is_string_(is_string) {}
  const std::string& name() const { return name_; }
  operator bool() const { return num_ != 0.0; }
  operator int() const { return (int)num_; }
  operator unsigned int() const { return (unsigned int)num_; }
  operator double() const { return num_; }
  operator float() const { return (float)num_; }
  operator std::string() const { return text_; }
  bool isString() const { return is_string_; }
 private:
  std::string name_, text_;
  double num_{0.0};
  bool is_string_{false};
};
inline Parameter IntegerParameter(const std::string& n, int v) { return Parameter(n, std::to_string(v), (double)v, false); }
inline Parameter FloatParameter(const std::string& n, double v) { return Parameter(n, std::to_string(v), v, false); }
inline Parameter BooleanParameter(const std::string& n, bool v) { return Parameter(n, v ? "true" : "false", v ? 1.0 : 0.0, false); }
inline Parameter StringParameter(const std::string& n, const std::string& v) { return Parameter(n, v, 0.0, true); }
class ParameterList {
 public:
  typedef std::map<std::string, Parameter>::const_iterator map_iterator;
  class const_iterator {
   public:
    explicit const_iterator(map_iterator it) : it_(it) {}
    const Parameter& operator*() const { return it_->second; }
    const Parameter* operator->() const { return &it_->second; }
    const_iterator& operator++() { ++it_; return *this; }
    bool operator!=(const const_iterator& o) const { return it_ != o.it_; }
    bool operator==(const const_iterator& o) const { return it_ == o.it_; }
   private:
    map_iterator it_;
  };
  void insert(const Parameter& p) { params_[p.name()] = p; }
  void erase(const std::string& name) { params_.erase(name); }
  bool hasParam(const std::string& name) const { return params_.count(name) != 0; }
  const Parameter& operator[](const std::string& name) const {
    map_iterator it = params_.find(name);
    if (it == params_.end()) throw std::runtime_error("Parameter with name '" + name + "' not found.");
    return it->second;
  }
  template <class T>
  T getValue(const std::string& name) const { return static_cast<T>((*this)[name]); }
  const_iterator begin() const { return const_iterator(params_.begin()); }
  const_iterator end() const { return const_iterator(params_.end()); }
 private:
  std::map<std::string, Parameter> params_;
};
}  // namespace rv
#endif
