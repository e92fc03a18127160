// viorb_amd/csrc/viorb_common.h — error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/viorb.h"

namespace viorb {

// thread-local last-error text (viorb_last_error())
char* last_error_buf();
void set_error(const char* fmt, ...);

// Optional per-kernel timing with HIP events recorded on the stream the kernel is launched on
// (viorb_profile_* in include/viorb.h); used by bench.py for the roofline line. Off by default.
struct ProfScope {
    int idx;
    hipStream_t st;
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
};
bool prof_times_everything();          // no kernel selection (viorb_profile_select(NULL)): every launch is timed

// hipFuncAttributeMaxDynamicSharedMemorySize is a process-wide, per-kernel setting: every handle asks for its own size, so the limit is
// only ever raised (a later, smaller handle must not lower it under a long-lived one's launches).
hipError_t raise_dynamic_lds(const void* kernel, size_t bytes);

} // namespace viorb

#define VIORB_HIP_TRY(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            viorb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                       \
            return VIORB_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

#define VIORB_REQUIRE(cond, msg)                      \
    do {                                              \
        if (!(cond)) {                                \
            viorb::set_error("invalid argument: %s", msg); \
            return VIORB_ERR_INVALID_ARG;             \
        }                                             \
    } while (0)
