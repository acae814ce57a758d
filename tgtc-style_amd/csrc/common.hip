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
extern "C" const char* tgtc_last_error(void) { return tgtc::err_buf(); }
