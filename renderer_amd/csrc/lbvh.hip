// LBVH build (placeholder until the GPU build lands): scenes render through the brute-force kernel.
#include "internal.h"

namespace inv {
hipError_t build_lbvh(hipStream_t, DeviceScene& sc) {
    sc.has_bvh = false;
    return hipSuccess;
}
hipError_t launch_render_bvh(hipStream_t s, const FrameK& f, const DeviceScene& sc, const Outputs& out) {
    return launch_render_brute(s, f, sc, out);
}
} // namespace inv
