// rt_error.hpp — per-thread last-error text behind rt_last_error().
#pragma once

#include <stdexcept>
#include <string>

#include "rt_amd.h"

namespace rt {

// A failed device call outside the C ABI's own translation unit; the ABI reports `status`
// (RT_ERR_OUT_OF_MEMORY, RT_ERR_NO_DEVICE or RT_ERR_DEVICE), never "invalid argument".
struct DeviceFailure : std::runtime_error {
    rt_status status;
    DeviceFailure(rt_status st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

// Records `msg` as the calling thread's last error and returns `st`.
rt_status fail(rt_status st, const std::string &msg);
const char *last_error_cstr();

} // namespace rt
