// Library-level entry points of include/casapose_hip.h: error reporting and probing.
#include "common.h"

#include <string>

namespace {
thread_local std::string g_last_error;
}

namespace cp {
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace cp

extern "C" const char* cp_last_error(void) { return g_last_error.c_str(); }
extern "C" int cp_version(void) { return 100; }
extern "C" int cp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
