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

extern "C" int m355_version(void) { return 100; }   // 0.1.0

extern "C" const char* m355_last_error(void) { return m355::error_buffer(); }
