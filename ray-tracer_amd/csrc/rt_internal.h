// rt_internal.h — shared between the C-ABI implementation (csrc/) and the host mirror (host/).
#pragma once
#include "../../include/rt_abi.h"

// printf-style setter for the thread-local text behind rt_last_error().
extern "C" void rt_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
