// ure_common.hip -- error reporting and device query for libultrare_hip.so.
#include "ure_internal.h"

#include <cstring>

namespace ure {

char *err_buf()
{
    static thread_local char buf[1024] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 1024, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ure

extern "C" {

int ure_abi_version(void) { return URE_ABI_VERSION; }

#ifndef URE_SOURCE_HASH
#define URE_SOURCE_HASH "unknown"
#endif
const char *ure_source_hash(void) { return URE_SOURCE_HASH; }

const char *ure_last_error(void) { return ure::err_buf(); }

int ure_device_info(int dev, int *n_cu, int *wave_size, char *arch, int arch_len)
{
    hipDeviceProp_t p;
    URE_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

}  // extern "C"
