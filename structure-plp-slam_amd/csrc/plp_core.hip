// Library-wide C entry points: status strings, version, device probe.
#include "plp_common.hpp"

namespace plp {
static thread_local std::string g_last_error;

plp_status set_error(plp_status s, const char* msg) {
    g_last_error = msg ? msg : "";
    return s;
}
plp_status set_hip_error(hipError_t e, const char* expr, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", expr, hipGetErrorString(e), file, line);
    g_last_error = buf;
    return PLP_ERR_HIP;
}
}  // namespace plp

extern "C" {
const char* plp_strerror(plp_status s) {
    switch (s) {
        case PLP_OK: return "ok";
        case PLP_ERR_INVALID_ARG: return "invalid argument";
        case PLP_ERR_NO_DEVICE: return "no HIP device";
        case PLP_ERR_HIP: return "HIP runtime error";
        case PLP_ERR_CAPACITY: return "output capacity too small";
        case PLP_ERR_OVERFLOW: return "internal buffer overflow";
        case PLP_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown status";
}
const char* plp_last_error(void) { return plp::g_last_error.c_str(); }
int plp_version(void) { return 1; }
int plp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
}
