// abi_common.hpp — what every translation unit of libbrl_hip.so shares on the host side: the thread-local error message behind
// brl_last_error() (owned by brl_env.hip) and the argument / HIP-call checks of the C-ABI entry points (include/brl_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/brl_hip.h"

// records the message brl_last_error() returns on this thread; returns `code`
int brl_fail(int code, const char *fmt, const char *detail);

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return brl_fail(BRL_E_HIP, #expr ": %s", hipGetErrorString(_e)); \
  } while (0)

#define NEED(cond, what)                                                   \
  do {                                                                     \
    if (!(cond)) return brl_fail(BRL_E_ARG, "bad argument: %s", what);     \
  } while (0)
