// qd_physics.hip -- driver-side per-step diagnostics (pygcm/physics.py, run_simulation.py:1766-2146).
#include "qd_internal.h"
#include "qd_device.h"

int qd_driver_physics_impl(qd_ctx* c, double dt) {
    (void)dt;
    return qd_fail(c, "qd_driver_physics: not built yet");
}
