// Shared by the translation units that implement the C ABI (pilot_ot.hip, pilot_ot_multi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pilot_ot.h"

#define PILOT_API extern "C" __attribute__((visibility("default")))

namespace pilot {

// record the calling thread's error message (pilot_ot_last_error) and return `code`
int abi_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// release the calling thread's cached multi-GPU context (called by pilot_ot_shutdown)
void abi_multi_release();

}  // namespace pilot

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return pilot::abi_fail(PILOT_OT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                   __FILE__, __LINE__);                                                    \
    } while (0)
