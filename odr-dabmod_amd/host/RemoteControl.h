// RemoteControl.h -- the slice of ODR-DabMod's remote-control interface that the
// hot-path stages implement (reference lib/RemoteControl.h:62-131, lib/Json.h:44-62):
// a named object with string-typed parameters that another thread may set at any time.
// The control plane itself (telnet/ZMQ servers) is out of scope; this keeps the
// stage-side contract so the adapters compile against either header set.
#pragma once

#include <cstdint>
#include <list>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

namespace json {
struct value_t {
    std::variant<std::shared_ptr<std::unordered_map<std::string, value_t>>, std::vector<value_t>,
                 std::string, double, int64_t, uint64_t, int32_t, uint32_t, bool, std::nullopt_t>
        v;
};
using map_t = std::unordered_map<std::string, value_t>;
}  // namespace json

class ParameterError : public std::exception {
public:
    explicit ParameterError(std::string message) : m_message(std::move(message)) {}
    const char *what() const noexcept override { return m_message.c_str(); }

private:
    std::string m_message;
};

#define RC_ADD_PARAMETER(p, desc)                                                              \
    {                                                                                          \
        std::vector<std::string> p;                                                            \
        p.push_back(#p);                                                                       \
        p.push_back(desc);                                                                     \
        m_parameters.push_back(p);                                                             \
    }

class RemoteControllable {
public:
    explicit RemoteControllable(const std::string &name) : m_rc_name(name) {}
    RemoteControllable(const RemoteControllable &) = delete;
    RemoteControllable &operator=(const RemoteControllable &) = delete;
    virtual ~RemoteControllable() = default;

    virtual std::string get_rc_name() const { return m_rc_name; }
    virtual std::list<std::vector<std::string>> get_parameter_descriptions() const { return m_parameters; }
    virtual void set_parameter(const std::string &parameter, const std::string &value) = 0;
    virtual const std::string get_parameter(const std::string &parameter) const = 0;
    virtual const json::map_t get_all_values() const = 0;

protected:
    std::string m_rc_name;
    std::list<std::vector<std::string>> m_parameters;
};
