// rt_error.hpp — per-thread last-error text behind rt_last_error().
#pragma once

#include <string>

#include "rt_amd.h"

namespace rt {

// Records `msg` as the calling thread's last error and returns `st`.
rt_status fail(rt_status st, const std::string &msg);
const char *last_error_cstr();

} // namespace rt
