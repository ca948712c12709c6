// ure_common.hip -- error reporting and device query for libultrare_hip.so.
#include "ure_internal.h"

#include <cstdlib>
#include <cstring>
#include <thread>

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

// Host threads "all cores" means: the CPUs this process may really use -- its affinity mask, cut down to the container's CPU
// quota (cgroup v2 cpu.max / v1 cfs_quota_us) -- not the machine's CPU count.  On the 1-GPU boxes this was measured on the
// machine has 256 CPUs and the container 16: 256 threads exhaust the quota of a 100 ms period and the whole process is
// throttled for the rest of it.
int host_threads()
{
    static const int n = []() {
        int cpus = (int)std::thread::hardware_concurrency();
        if (cpus < 1) cpus = 1;
        long long quota = -1, period = 0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
            std::fclose(f);
        } else {
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
            if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
        }
        if (quota > 0 && period > 0) cpus = (int)std::min<long long>(cpus, std::max<long long>(1, (quota + period - 1) / period));
        return cpus;
    }();
    return n;
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
