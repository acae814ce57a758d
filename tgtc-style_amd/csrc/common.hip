#include "common.h"

namespace tgtc {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace tgtc

extern "C" int tgtc_version(void) { return 100; /* 0.1.0 */ }
// 1 in libtgtc_hip_dev.so (`make dev`: the 32x32x16 "wide" development kernels are compiled in), 0 in the product library
extern "C" int tgtc_dev_kernels(void) {
#ifdef TGTC_DEV_KERNELS
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* tgtc_last_error(void) { return tgtc::err_buf(); }
