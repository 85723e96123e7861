// TEST SHIM (tests/test_abi_and_host.py::test_reference_apps_compile_against_the_facade): the logging calls the reference's applications make, as no-ops, so that
// /root/reference/app/*.cpp can be syntax-checked against include/ without the spdlog submodule (empty in the reference checkout).  Not a product header.
#pragma once
#include <string>
namespace spdlog
{
template <typename... A> inline void trace(const A&...) {}
template <typename... A> inline void debug(const A&...) {}
template <typename... A> inline void info(const A&...) {}
template <typename... A> inline void warn(const A&...) {}
template <typename... A> inline void error(const A&...) {}
template <typename... A> inline void critical(const A&...) {}
namespace level { enum level_enum { trace, debug, info, warn, err, critical, off }; }
inline void set_level(level::level_enum) {}
}  // namespace spdlog
