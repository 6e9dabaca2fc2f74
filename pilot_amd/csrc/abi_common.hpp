// Shared by the translation units that implement the C ABI (pilot_ot.hip, pilot_ot_multi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pilot_ot.h"

#define PILOT_API extern "C" __attribute__((visibility("default")))

namespace pilot {

// record the calling thread's error message (pilot_ot_last_error) and return `code`
int abi_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// value of a test switch (pilot_ot_test_switch), or nullptr: how the GPU tests and the A/B tools force a kernel variant.  The
// library reads NO environment variable that changes what it computes.
const char *test_switch(const char *name);
// release every cached multi-GPU context of the host-buffer entry points (called by pilot_ot_shutdown)
void abi_multi_release();
// a buffer of the calling thread's pool of device temporaries (grown on demand, released by pilot_ot_shutdown); slots 0 .. 11
// belong to pilot_ot.hip, 12 .. 19 to pilot_ot_consumers.hip
hipError_t ws_buffer(int slot, size_t bytes, void **out);
// cell-level cohort, internal face used by the multi-device form (pilot_ot_multi.hip)
int cell_enqueue_rows(pilot_ot_cell_cohort *c, double scale, double reg, int num_iter_max, double stop_thr, int check_period,
                      double f32_floor_ulps, int row_begin, int row_end, int row_step, size_t *n_out);
int cell_collect(pilot_ot_cell_cohort *c, size_t n_out, double *w2 /* nullable */, int *iters, double *err, float *kernel_ms);
void cell_buffers(pilot_ot_cell_cohort *c, double **d_w2, hipStream_t *stream);

}  // namespace pilot

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return pilot::abi_fail(PILOT_OT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                   __FILE__, __LINE__);                                                    \
    } while (0)
