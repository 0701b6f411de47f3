// rk_internal.h -- shared between the translation units of librappas_place.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <exception>
#include <new>

// Developer / test knobs exist only in builds with -DRK_DEV_KNOBS (rappas_amd/librappas_place_dev.so, the
// variant builds of scripts/).  The product library reads NO environment variable: a JVM hands its whole environment down.
#ifdef RK_DEV_KNOBS
static inline const char *rk_knob(const char *name) { return getenv(name); }
#else
static inline const char *rk_knob(const char *) { return nullptr; }
#endif


namespace rk {
// sets the thread-local message rk_last_error() returns and hands back `code`
int fail_msg(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace rk

// No exception may cross the C ABI (the caller can be a JVM): std::bad_alloc / std::system_error from the host-side containers and
// threads become error codes.
#define RK_GUARD_BEGIN try {
#define RK_GUARD_END(who)                                                                                         \
    } catch (const std::bad_alloc &) {                                                                            \
        return rk::fail_msg(RK_ERR_NOMEM, "%s: out of host memory", who);                                         \
    } catch (const std::exception &e_) {                                                                          \
        return rk::fail_msg(RK_ERR_HIP, "%s: %s", who, e_.what());                                                \
    }

#define RK_HIP_TRY(expr)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return rk::fail_msg(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                hipGetErrorString(e_), __FILE__, __LINE__);                                       \
    } while (0)
