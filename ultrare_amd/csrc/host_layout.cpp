// host_layout.cpp -- host-side ingest of the SISA path: CSV -> triples -> per-shard HBM layout.
//
//   ure_host_read_csv      read.py:37   pd.read_csv(dir, header=None): `uid,iid,rating` rows
//   ure_host_partition     read.py:52-70 shard s = rows (file order) whose user is in group s and
//                                        not deleted; rating / max_rating
//   ure_host_build_layout  (no reference counterpart) the slot arrays, schedule and positions the
//                          step kernel walks (include/ultrare_hip.h, struct ure_shard), built with
//                          linear counting sorts instead of numpy argsorts
//
// The reference spends its time here in pandas / np.in1d / per-sample Dataset objects; with
// training at tens of milliseconds this host work IS the end-to-end cost of an unlearning request
// (new deletion set -> new shards -> new layouts), so it is native and linear.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "ultrare_hip.h"

#ifndef URE_NARROW_MAX
#define URE_NARROW_MAX 32     // as in ure_internal.h: widest row with one float4 per lane
#endif

namespace ure {
int fail(int code, const char *fmt, ...);
char *err_buf();
int host_threads();
}

static int build_units_impl(const int32_t *sched, int32_t n_active, int32_t d, int32_t unit_passes, int32_t *units, int64_t capacity, int64_t *n_units);

namespace {

// strtod-free number parsing for the plain decimals of rating files ("123", "3.5", "4.0");
// anything else (exponents, inf, nan, hex) falls back to strtod for that field.
inline const char *parse_field(const char *p, const char *end, double *out, bool *ok)
{
    const char *s = p;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    uint64_t ip = 0;
    int nd = 0;
    while (p < end && *p >= '0' && *p <= '9') { ip = ip * 10 + (uint64_t)(*p - '0'); ++p; ++nd; }
    double v = (double)ip;
    if (p < end && *p == '.') {
        ++p;
        uint64_t fp = 0;
        int fd = 0;
        while (p < end && *p >= '0' && *p <= '9') { if (fd < 18) { fp = fp * 10 + (uint64_t)(*p - '0'); ++fd; } ++p; ++nd; }
        static const double pw[19] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18};
        // exact for the <= 15 significant digits of rating files: one correctly rounded division
        if (nd <= 15) v = (double)(ip * (uint64_t)pw[fd] + fp) / pw[fd];
        else v = (double)ip + (double)fp / pw[fd];
    }
    const bool plain = nd > 0 && nd <= 15 && (p == end || *p == ',' || *p == '\n' || *p == '\r' || *p == ' ');
    if (!plain) {   // exponent or something unusual: let strtod decide
        char buf[64];
        size_t len = 0;
        const char *q = s;
        while (q < end && *q != ',' && *q != '\n' && *q != '\r' && len < sizeof(buf) - 1) buf[len++] = *q++;
        buf[len] = 0;
        char *e = nullptr;
        v = std::strtod(buf, &e);
        if (e == buf) *ok = false;
        *out = v;
        return q;
    }
    *out = neg ? -v : v;
    return p;
}

struct Chunk {
    std::vector<int32_t> u, i;
    std::vector<double> r;
    bool ok = true;
};

}  // namespace

extern "C" {

int ure_host_read_csv(const char *path, int32_t **uid, int32_t **iid, double **rating, int64_t *n_rows, int n_threads)
{
    if (!path || !uid || !iid || !rating || !n_rows) return ure::fail(-1, "ure_host_read_csv: bad arguments");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return ure::fail(-1, "ure_host_read_csv: cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return ure::fail(-1, "ure_host_read_csv: cannot stat %s", path); }
    const size_t size = (size_t)sb.st_size;
    *uid = *iid = nullptr; *rating = nullptr; *n_rows = 0;
    if (size == 0) { close(fd); return 0; }
    const char *data = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == MAP_FAILED) return ure::fail(-1, "ure_host_read_csv: mmap failed for %s", path);
    int nt = n_threads > 0 ? n_threads : ure::host_threads();
    nt = std::max(1, std::min<int>(nt, (int)(size / (1 << 16)) + 1));
    std::vector<size_t> cut(nt + 1, size);
    cut[0] = 0;
    for (int t = 1; t < nt; ++t) {
        size_t p = size / nt * t;
        while (p < size && data[p - 1] != '\n') ++p;
        cut[t] = p;
    }
    std::vector<Chunk> chunks(nt);
    auto work = [&](int t) {
        Chunk &c = chunks[t];
        const char *p = data + cut[t], *end = data + cut[t + 1];
        const size_t guess = (cut[t + 1] - cut[t]) / 8 + 16;
        c.u.reserve(guess); c.i.reserve(guess); c.r.reserve(guess);
        while (p < end) {
            while (p < end && (*p == '\n' || *p == '\r' || *p == ' ')) ++p;
            if (p >= end) break;
            double a, b, r;
            p = parse_field(p, end, &a, &c.ok);
            if (p >= end || *p != ',') { c.ok = false; break; }
            p = parse_field(p + 1, end, &b, &c.ok);
            if (p >= end || *p != ',') { c.ok = false; break; }
            p = parse_field(p + 1, end, &r, &c.ok);
            while (p < end && *p != '\n') ++p;          // further columns (timestamps) are ignored
            c.u.push_back((int32_t)a); c.i.push_back((int32_t)b); c.r.push_back(r);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    munmap((void *)data, size);
    int64_t total = 0;
    for (auto &c : chunks) { if (!c.ok) return ure::fail(-1, "ure_host_read_csv: malformed row in %s", path); total += (int64_t)c.u.size(); }
    *uid = (int32_t *)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1));
    *iid = (int32_t *)std::malloc(sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1));
    *rating = (double *)std::malloc(sizeof(double) * (size_t)std::max<int64_t>(total, 1));
    if (!*uid || !*iid || !*rating) return ure::fail(-1, "ure_host_read_csv: out of memory");
    int64_t o = 0;
    for (auto &c : chunks) {
        std::memcpy(*uid + o, c.u.data(), c.u.size() * sizeof(int32_t));
        std::memcpy(*iid + o, c.i.data(), c.i.size() * sizeof(int32_t));
        std::memcpy(*rating + o, c.r.data(), c.r.size() * sizeof(double));
        o += (int64_t)c.u.size();
    }
    *n_rows = total;
    return 0;
}

void ure_host_free(void *p) { std::free(p); }

int ure_host_partition(const int32_t *uid, const int32_t *iid, const double *rating, int64_t n, const int32_t *shard_of_user,
                       int32_t n_user, int32_t n_shards, double max_rating, int64_t *counts, int32_t *out_uid, int32_t *out_iid,
                       float *out_rating, double *out_rating64)
{
    if (!uid || !iid || !rating || !shard_of_user || !counts || n < 0 || n_user <= 0 || n_shards <= 0)
        return ure::fail(-1, "ure_host_partition: bad arguments");
    std::fill(counts, counts + n_shards, (int64_t)0);
    for (int64_t j = 0; j < n; ++j) {
        const int32_t u = uid[j];
        if (u < 0 || u >= n_user) return ure::fail(-1, "ure_host_partition: user id %d outside [0, %d)", u, n_user);
        const int32_t s = shard_of_user[u];
        if (s >= n_shards) return ure::fail(-1, "ure_host_partition: shard %d outside [0, %d)", s, n_shards);
        if (s >= 0) ++counts[s];
    }
    if (!out_uid) return 0;                       // counting pass only
    std::vector<int64_t> cur(n_shards, 0);
    for (int s = 1; s < n_shards; ++s) cur[s] = cur[s - 1] + counts[s - 1];
    for (int64_t j = 0; j < n; ++j) {
        const int32_t s = shard_of_user[uid[j]];
        if (s < 0) continue;                      // deleted user / user of no group
        const int64_t o = cur[s]++;
        out_uid[o] = uid[j];
        out_iid[o] = iid[j];
        const double q = rating[j] / max_rating;              // read.py:66 (float64 division)
        if (out_rating) out_rating[o] = (float)q;             // read.py:113,124: one cast to float32
        if (out_rating64) out_rating64[o] = q;
    }
    return 0;
}

}  // extern "C"

// The slot layout of one shard (struct ure_shard) from its triples; ids / ratings in whatever width the caller holds them.
template <typename IdT, typename RT>
static int build_layout_t(const IdT *uid, const IdT *iid, const RT *rating, int64_t n, int32_t n_user, int32_t n_item,
                          int32_t *ent_oid, float *ent_r, int32_t *ent_src, int32_t *sched, int32_t *row_slot,
                          int64_t *n_slots, int32_t *n_active, int32_t *u_pos, int32_t *i_pos, bool packed)
{
    const int64_t n_rows = (int64_t)n_user + n_item;
    std::vector<int64_t> nnz(n_rows, 0);
    for (int64_t j = 0; j < n; ++j) {
        if (uid[j] < 0 || uid[j] >= n_user || iid[j] < 0 || iid[j] >= n_item)
            return ure::fail(-2, "ure_host_build_layout: interaction %lld has an id outside [0,%d) x [0,%d)", (long long)j, n_user, n_item);
        ++nnz[uid[j]];
        ++nnz[n_user + iid[j]];
    }
    // schedule: rows by nnz descending, ties by row id (= a stable sort on -nnz): counting sort on nnz
    int64_t max_nnz = 0;
    for (int64_t r = 0; r < n_rows; ++r) max_nnz = std::max(max_nnz, nnz[r]);
    std::vector<int64_t> first(max_nnz + 2, 0);
    for (int64_t r = 0; r < n_rows; ++r) ++first[max_nnz - nnz[r] + 1];
    for (int64_t v = 1; v <= max_nnz + 1; ++v) first[v] += first[v - 1];
    std::vector<int32_t> order(n_rows);
    for (int64_t r = 0; r < n_rows; ++r) order[first[max_nnz - nnz[r]]++] = (int32_t)r;
    int64_t slots = 0;
    for (int64_t r = 0; r < n_rows; ++r) slots += (nnz[r] + 7) / 8 * 8;
    slots = std::max<int64_t>(slots, 8);
    if (slots >= ((int64_t)1 << 31)) return ure::fail(-3, "ure_host_build_layout: shard too large for 32-bit slot indices (%lld slots)", (long long)slots);
    if (packed) {
        // one region: ent_oid [k] | ent_r [k] | ent_src [k] | sched [rows][4] | row_slot [rows], k = the slot count (known by now)
        ent_r = reinterpret_cast<float *>(ent_oid + slots);
        ent_src = ent_oid + 2 * slots;
        sched = ent_oid + 3 * slots;
        row_slot = sched + 4 * n_rows;
    }
    std::vector<int64_t> row_beg(n_rows);
    int64_t at = 0;
    int32_t na = 0;
    for (int64_t q = 0; q < n_rows; ++q) {
        const int32_t r = order[q];
        const int64_t padded = (nnz[r] + 7) / 8 * 8;
        row_beg[r] = at;
        sched[4 * q + 0] = r;
        sched[4 * q + 1] = (int32_t)at;
        sched[4 * q + 2] = (int32_t)(at + padded);
        sched[4 * q + 3] = (int32_t)nnz[r];
        at += padded;
        na += nnz[r] > 0;
        if (row_slot) row_slot[r] = nnz[r] > 0 ? (int32_t)q : -1;      // active rows come first in the schedule: q < n_active
    }
    *n_slots = slots;
    *n_active = na;
    std::memset(ent_oid, 0, sizeof(int32_t) * (size_t)slots);
    std::memset(ent_r, 0, sizeof(float) * (size_t)slots);
    std::fill(ent_src, ent_src + slots, (int32_t)-1);
    // file order inside every segment: one cursor per row
    std::vector<int64_t> cur(row_beg);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t pu = cur[uid[j]]++, pi = cur[n_user + iid[j]]++;
        const float rj = (float)rating[j];
        ent_oid[pu] = (int32_t)iid[j]; ent_r[pu] = rj; ent_src[pu] = (int32_t)j;
        ent_oid[pi] = (int32_t)uid[j]; ent_r[pi] = rj; ent_src[pi] = (int32_t)j;
        if (u_pos) u_pos[j] = (int32_t)pu;
        if (i_pos) i_pos[j] = (int32_t)pi;
    }
    return 0;
}

extern "C" {

int ure_host_partition64(const int32_t *uid, const int32_t *iid, const double *rating, int64_t n, const int32_t *shard_of_user,
                         int32_t n_user, int32_t n_shards, double max_rating, int64_t *counts, double *out)
{
    if (!uid || !iid || !rating || !shard_of_user || !counts || n < 0 || n_user <= 0 || n_shards <= 0)
        return ure::fail(-1, "ure_host_partition64: bad arguments");
    std::fill(counts, counts + n_shards, (int64_t)0);
    for (int64_t j = 0; j < n; ++j) {
        const int32_t u = uid[j];
        if (u < 0 || u >= n_user) return ure::fail(-1, "ure_host_partition64: user id %d outside [0, %d)", u, n_user);
        const int32_t s = shard_of_user[u];
        if (s >= n_shards) return ure::fail(-1, "ure_host_partition64: shard %d outside [0, %d)", s, n_shards);
        if (s >= 0) ++counts[s];
    }
    if (!out) return 0;                           // counting pass only
    // shard s is the block [3][counts[s]] at 3 * sum(counts[:s]): its uid row, its iid row, its rating row
    std::vector<double *> row_u(n_shards), row_i(n_shards), row_r(n_shards);
    double *at = out;
    for (int s = 0; s < n_shards; ++s) {
        row_u[s] = at; row_i[s] = at + counts[s]; row_r[s] = at + 2 * counts[s];
        at += 3 * counts[s];
    }
    for (int64_t j = 0; j < n; ++j) {
        const int32_t s = shard_of_user[uid[j]];
        if (s < 0) continue;                      // deleted user / user of no group
        *row_u[s]++ = (double)uid[j];
        *row_i[s]++ = (double)iid[j];
        *row_r[s]++ = rating[j] / max_rating;     // read.py:66 (float64 division)
    }
    return 0;
}

int ure_host_build_layout(const int32_t *uid, const int32_t *iid, const float *rating, int64_t n, int32_t n_user, int32_t n_item,
                          int32_t *ent_oid, float *ent_r, int32_t *ent_src, int32_t *sched,
                          int64_t *n_slots, int32_t *n_active, int32_t *u_pos, int32_t *i_pos)
{
    if (!uid || !iid || !rating || !ent_oid || !ent_r || !ent_src || !sched || !n_slots || !n_active ||
        n <= 0 || n_user <= 0 || n_item <= 0)
        return ure::fail(-1, "ure_host_build_layout: bad arguments");
    return build_layout_t(uid, iid, rating, n, n_user, n_item, ent_oid, ent_r, ent_src, sched, nullptr, n_slots, n_active, u_pos, i_pos, false);
}

static int build_layouts_impl(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                              int32_t n_user, int32_t n_item, int32_t *const *region, int64_t *n_slots, int32_t *n_active, int n_threads,
                              int32_t units_d, const int64_t *region_words, int64_t *n_units, int32_t *const *dev_region = nullptr, int device = -1,
                              void *stream = nullptr);

int ure_host_build_layouts(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                           int32_t n_user, int32_t n_item, int32_t *const *region, int64_t *n_slots, int32_t *n_active, int n_threads)
{
    return build_layouts_impl(n_shards, uid, iid, rating, n, n_user, n_item, region, n_slots, n_active, n_threads, 0, nullptr, nullptr);
}

int ure_host_build_layouts_units(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                                 int32_t n_user, int32_t n_item, int32_t *const *region, const int64_t *region_words, int64_t *n_slots,
                                 int32_t *n_active, int32_t units_d, int64_t *n_units, int n_threads)
{
    if (!region_words || !n_units || units_d < 4 || units_d > 256 || (units_d & (units_d - 1)))
        return ure::fail(-1, "ure_host_build_layouts_units: bad arguments");
    return build_layouts_impl(n_shards, uid, iid, rating, n, n_user, n_item, region, n_slots, n_active, n_threads, units_d, region_words, n_units);
}

// ure_host_build_layouts_units on a thread of the library's own, started by the caller's thread itself: a request's layouts are what its
// training waits for longest, and handing the blocking call to a Python worker cost 0.3-0.4 ms before the builder ran at all (the worker has to
// be woken and given the interpreter lock by a calling thread that is busy).  Every argument must stay alive and unchanged until the wait.
namespace {
struct AsyncBuild {
    std::thread th;
    int rc = 0;
    std::string why;
};
std::mutex g_async_lock;
std::map<int64_t, AsyncBuild *> g_async;
int64_t g_async_next = 1;
}  // namespace

int64_t ure_host_build_layouts_units_start(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                                           int32_t n_user, int32_t n_item, int32_t *const *region, const int64_t *region_words, int64_t *n_slots,
                                           int32_t *n_active, int32_t units_d, int64_t *n_units, int n_threads, int32_t *const *dev_region, int32_t device,
                                           void *stream)
{
    if (!region_words || !n_units || units_d < 4 || units_d > 256 || (units_d & (units_d - 1))) {
        ure::fail(-1, "ure_host_build_layouts_units_start: bad arguments");
        return 0;
    }
    AsyncBuild *job = new AsyncBuild();
    try {
        job->th = std::thread([=]() {
            job->rc = build_layouts_impl(n_shards, uid, iid, rating, n, n_user, n_item, region, n_slots, n_active, n_threads, units_d, region_words, n_units,
                                         dev_region, device, stream);
            if (job->rc) job->why = ure::err_buf();
        });
    } catch (...) {
        delete job;
        ure::fail(-1, "ure_host_build_layouts_units_start: no thread");
        return 0;
    }
    std::lock_guard<std::mutex> hold(g_async_lock);
    const int64_t handle = g_async_next++;
    g_async[handle] = job;
    return handle;
}

int ure_host_build_layouts_units_wait(int64_t handle)
{
    AsyncBuild *job = nullptr;
    {
        std::lock_guard<std::mutex> hold(g_async_lock);
        auto it = g_async.find(handle);
        if (it == g_async.end()) return ure::fail(-1, "ure_host_build_layouts_units_wait: unknown handle %lld", (long long)handle);
        job = it->second;
        g_async.erase(it);
    }
    job->th.join();
    const int rc = job->rc;
    if (rc) ure::fail(rc, "%s", job->why.c_str());
    delete job;
    return rc;
}

static int build_layouts_impl(int n_shards, const int64_t *const *uid, const int64_t *const *iid, const double *const *rating, const int64_t *n,
                              int32_t n_user, int32_t n_item, int32_t *const *region, int64_t *n_slots, int32_t *n_active, int n_threads,
                              int32_t units_d, const int64_t *region_words, int64_t *n_units, int32_t *const *dev_region, int device, void *stream)
{
    if (n_shards <= 0 || !uid || !iid || !rating || !n || !region || !n_slots || !n_active || n_user <= 0 || n_item <= 0)
        return ure::fail(-1, "ure_host_build_layouts: bad arguments");
    for (int s = 0; s < n_shards; ++s)
        if (!uid[s] || !iid[s] || !rating[s] || !region[s] || n[s] <= 0) return ure::fail(-1, "ure_host_build_layouts: shard %d: bad arguments", s);
    int nt = n_threads > 0 ? n_threads : ure::host_threads();
    nt = std::max(1, std::min(nt, n_shards));
    std::atomic<int> next{0}, rc{0};
    // (ure::fail keeps its message per thread: a worker's failure is re-reported on the calling thread below)
    std::vector<int> bad(n_shards, 0);
    std::vector<std::string> why(n_shards);
    auto work = [&]() {
        if (dev_region && device >= 0) (void)hipSetDevice(device);
        for (int s = next.fetch_add(1); s < n_shards; s = next.fetch_add(1)) {
            const int r = build_layout_t(uid[s], iid[s], rating[s], n[s], n_user, n_item, region[s], (float *)nullptr, (int32_t *)nullptr,
                                         (int32_t *)nullptr, (int32_t *)nullptr, n_slots + s, n_active + s, (int32_t *)nullptr, (int32_t *)nullptr, true);
            if (r) { bad[s] = r; why[s] = ure::err_buf(); rc.store(r); continue; }
            if (units_d) {
                // the work units of this table width right behind the layout, so that ONE copy takes both to the device:
                // region = layout (3 k + 5 rows words) | pad to 8 words | units [n_units][4].  -1: they did not fit (the caller asks
                // ure_host_build_units later)
                const int64_t rows = (int64_t)n_user + n_item;
                const int64_t at = (3 * n_slots[s] + 5 * rows + 7) / 8 * 8;
                const int64_t cap = (region_words[s] - at) / 4;
                const int32_t *sched = region[s] + 3 * n_slots[s];
                int64_t need = 0;
                n_units[s] = -1;
                if (cap > 0 && build_units_impl(sched, n_active[s], units_d, 1, nullptr, 0, &need) == 0 && need <= cap &&
                    build_units_impl(sched, n_active[s], units_d, 1, region[s] + at, cap, &need) == 0)
                    n_units[s] = need;
                if (dev_region) {
                    // the shard's layout (and units) on its way to the device while the other shards are still being built: as ONE copy behind
                    // the last shard the upload of a 32-shard request at the 25 M shape (1.1 GB) stood between the build and the job, 28 ms
                    const int64_t used = at + (n_units[s] > 0 ? (4 * n_units[s] + 7) / 8 * 8 : 0);
                    const hipError_t e = hipMemcpyAsync(dev_region[s], region[s], (size_t)used * 4, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
                    if (e != hipSuccess) { bad[s] = (int)e; why[s] = hipGetErrorString(e); rc.store((int)e); }
                }
            }
        }
    };
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }
    if (rc.load())
        for (int s = 0; s < n_shards; ++s)
            if (bad[s]) return ure::fail(bad[s], "ure_host_build_layouts: shard %d: %s", s, why[s].c_str());      // -2: an id outside its range, -3: too large
    return 0;
}

// Work units of the step kernel.  A lane group of L lanes scans 8 L slots per pass, so a row is cut
// into pieces of 8 L slots (one pass each); a row with more pieces than a workgroup has lane groups
// (256 / L) is cut into 256 / L longer pieces instead.  Rows are taken heaviest first; when the next
// heavy row does not fit into what is left of the workgroup, the gap is filled with the lightest
// rows, so a row never straddles workgroups and its partial sums meet in LDS.
int ure_host_build_units(const int32_t *sched, int32_t n_active, int32_t d, int32_t unit_passes, int32_t *units, int64_t capacity, int64_t *n_units)
{
    if (!sched || !n_units || n_active < 0 || d < 4 || d > 256 || (d & (d - 1)) || unit_passes < 1 || unit_passes > 4096)
        return ure::fail(-1, "ure_host_build_units: bad arguments");
    return build_units_impl(sched, n_active, d, unit_passes, units, capacity, n_units);
}

}  // extern "C"

static int build_units_impl(const int32_t *sched, int32_t n_active, int32_t d, int32_t unit_passes, int32_t *units, int64_t capacity, int64_t *n_units)
{
    const int lanes = d <= URE_NARROW_MAX ? d / 4 : d / 8;
    const int cap = 8 * lanes * unit_passes, upb = 256 / lanes;        // slots of a unit: unit_passes scan passes of one lane group
    auto pieces = [&](int64_t q, int32_t *len) {
        const int64_t slots = (int64_t)sched[4 * q + 2] - sched[4 * q + 1];
        int64_t l = cap, nu = (slots + cap - 1) / cap;
        if (nu > upb) {
            l = ((slots + upb - 1) / upb + 7) / 8 * 8;
            nu = (slots + l - 1) / l;
        }
        *len = (int32_t)l;
        return (int32_t)nu;
    };
    int64_t out = 0;
    int64_t i = 0, j = (int64_t)n_active - 1;
    while (i <= j) {
        const int64_t block0 = out;
        int left = upb;
        bool multi = false;
        while (left > 0 && i <= j) {
            int32_t len;
            int64_t q = i;
            int32_t nu = pieces(q, &len);
            if (nu > left) {
                q = j;
                nu = pieces(q, &len);
                if (nu > left) break;
                --j;
            } else {
                ++i;
            }
            if (units) {
                if (out + nu > capacity) return ure::fail(-1, "ure_host_build_units: capacity %lld too small", (long long)capacity);
                const int32_t beg = sched[4 * q + 1], end = sched[4 * q + 2];
                const int32_t leader = upb - left;
                for (int32_t u = 0; u < nu; ++u) {
                    int32_t *e = units + 4 * (out + u);
                    e[0] = sched[4 * q + 0];
                    e[1] = beg + u * len;
                    e[2] = std::min(end, beg + (u + 1) * len);
                    e[3] = leader | (nu << 16);
                }
            }
            multi = multi || nu > 1;
            out += nu;
            left -= nu;
        }
        if (units) {
            if (out + left > capacity) return ure::fail(-1, "ure_host_build_units: capacity %lld too small", (long long)capacity);
            for (int k = 0; k < left; ++k) {
                int32_t *e = units + 4 * (out + k);
                e[0] = -1; e[1] = 0; e[2] = 0;
                e[3] = (upb - left + k) | (1 << 16);
            }
        }
        out += left;
        if (units && multi)
            for (int64_t u = block0; u < out; ++u) units[4 * u + 3] |= 1 << 30;
    }
    *n_units = out;
    return 0;
}

extern "C" {

// utils.py:377-396 of the comparison clusterer: labels from the [n][k] distance matrix.  capacity <= 0:
// new_label = dist.argmin(axis=1).  capacity > 0 (balanced k-means, capacity = ceil(n / k)): walk the
// (user, group) pairs in ascending distance (exact ties: ascending flat index) and give a user its first
// group that still has room.  *inertia = np.sum(dist[arange(n), label]) with numpy's float32 pairwise sum.
int ure_host_kmeans_assign(const float *dist_nk, int64_t n, int32_t k, int64_t capacity, int32_t *label, double *inertia)
{
    if (!dist_nk || !label || n <= 0 || k <= 0 || n * (int64_t)k >= ((int64_t)1 << 32))
        return ure::fail(-1, "ure_host_kmeans_assign: bad arguments (n=%lld k=%d)", (long long)n, k);
    if (capacity <= 0) {
        for (int64_t i = 0; i < n; ++i) {
            const float *row = dist_nk + i * k;
            int best = 0;
            bool nan_seen = row[0] != row[0];
            for (int c = 1; c < k && !nan_seen; ++c) {
                if (row[c] != row[c]) { best = c; nan_seen = true; }          // numpy: the first NaN wins
                else if (row[c] < row[best]) best = c;
            }
            label[i] = best;
        }
    } else {
        if (capacity * k < n) return ure::fail(-1, "ure_host_kmeans_assign: capacity %lld x %d groups < %lld users", (long long)capacity, k, (long long)n);
        const int64_t total = n * k;
        std::vector<uint64_t> key((size_t)total);
        for (int64_t t = 0; t < total; ++t) {
            uint32_t b;
            std::memcpy(&b, dist_nk + t, 4);
            b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // order-preserving map of float bits
            key[(size_t)t] = ((uint64_t)b << 32) | (uint64_t)t;
        }
        std::sort(key.begin(), key.end());
        std::vector<int64_t> left((size_t)k, capacity);
        std::vector<char> done((size_t)n, 0);
        std::fill(label, label + n, 0);
        int64_t n_done = 0;
        for (int64_t q = 0; q < total && n_done < n; ++q) {
            const int64_t t = (int64_t)(key[(size_t)q] & 0xFFFFFFFFu);
            const int64_t u = t / k;
            const int c = (int)(t % k);
            if (done[(size_t)u] || left[(size_t)c] <= 0) continue;
            label[u] = c;
            done[(size_t)u] = 1;
            --left[(size_t)c];
            ++n_done;
        }
    }
    if (inertia) {
        // numpy pairwise_sum over float32 (blocks of <= 128 with 8 accumulators, halves above)
        std::vector<float> v((size_t)n);
        for (int64_t i = 0; i < n; ++i) v[(size_t)i] = dist_nk[i * k + label[i]];
        struct Pair {
            static float sum(const float *a, int64_t m)
            {
                if (m < 8) {
                    float r = 0.f;
                    for (int64_t i = 0; i < m; ++i) r += a[i];
                    return r;
                }
                if (m <= 128) {
                    float r[8];
                    for (int j = 0; j < 8; ++j) r[j] = a[j];
                    int64_t i = 8;
                    for (; i < m - (m % 8); i += 8)
                        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
                    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                    for (; i < m; ++i) res += a[i];
                    return res;
                }
                int64_t h = m / 2;
                h -= h % 8;
                return sum(a, h) + sum(a + h, m - h);
            }
        };
        *inertia = (double)Pair::sum(v.data(), n);
    }
    return 0;
}

}  // extern "C"
