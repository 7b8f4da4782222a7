// failure.hpp -- how the library's host code reports an error: a (status code, message) pair thrown up to the C ABI's
// `guarded`, which turns it into the returned status and storm_last_error().  Nothing unwinds across the boundary.
#pragma once
#include <string>

namespace storm {

struct Failure {
  int code;
  std::string msg;
};
[[noreturn]] inline void fail(int code, const std::string& msg) { throw Failure{code, msg}; }

}  // namespace storm
