// Library-wide entry points: version and the per-thread error message.
#include "common.h"

namespace m355 {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace m355

extern "C" int m355_version(void) { return 110; }   // 0.1.1

extern "C" int m355_init(int device) {
    int count = 0, prev = 0;
    if (int rc = m355::hip_status(hipGetDeviceCount(&count), "hipGetDeviceCount")) return rc;
    if (device < 0 || device >= count) return m355::fail(-2, "m355_init: device %d of %d", device, count);
    if (int rc = m355::hip_status(hipGetDevice(&prev), "hipGetDevice")) return rc;
    if (prev != device)
        if (int rc = m355::hip_status(hipSetDevice(device), "hipSetDevice")) return rc;
    int rc = m355::moe_device_init(device);
    if (!rc) rc = m355::dense_tile_device_init(device);
    if (!rc) rc = m355::moe_static_device_init(device);
    if (prev != device) {
        const int rc2 = m355::hip_status(hipSetDevice(prev), "hipSetDevice");
        if (!rc) rc = rc2;
    }
    return rc;
}

extern "C" const char* m355_last_error(void) { return m355::error_buffer(); }
