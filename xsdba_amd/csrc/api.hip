// api.hip — host side of libxsdba_hip.so: context, group plans, staging, launches, C ABI (include/xsdba_hip.h).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <iterator>
#include <limits>
#include <cstdlib>
#include <vector>

#include "../../include/xsdba_hip.h"
#include "kernels.hpp"
#include "lin2d.hpp"
#include "train_cell3.hpp"
#include "train_cell.hpp"
#include "qdm_cell.hpp"
#include "qdm_lanes.hpp"
#include "train_long.hpp"
#include "train_xlong.hpp"
#include "adjust_tm.hpp"
#include "train_cell_f64.hpp"
#include "train_select.hpp"
#include "preproc.hpp"
#include "order_probe.hpp"
#include "train_bigrows.hpp"

#define API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Host-side description of the grouping: built once per (group_of_time, window) and cached.
struct Plan {
    uint64_t hash = 0;
    int64_t T = 0;
    int G = 0, W = 0;
    int max_row = 0, max_mem = 0;
    std::vector<int> row_ptr, row_idx, mem_ptr, mem_idx;
    std::vector<int> g_host;  // the group of every time step (compared on a hash hit)
    unsigned char *d_gidx8 = nullptr;  // the group of every time step in a byte (255: none) when there are at most 254 groups
    int *d_gidx = nullptr, *d_row_ptr = nullptr, *d_row_idx = nullptr, *d_mem_ptr = nullptr, *d_mem_idx = nullptr;
    // fast train path (train_cell.hpp): rows as sorted sets, toggle lists between consecutive groups
    bool simple = false;  // no time index occurs twice in one group row
    int *d_row_sorted = nullptr, *d_tog_ptr = nullptr, *d_tog_idx = nullptr;
    int *d_tog64 = nullptr;  // [G][64] padded toggle table when every list has <= 64 entries, else null
    int *d_tog64p = nullptr;  // the same table for k_train_cell3: no -1 entries (see get_plan)
    int *d_tog128 = nullptr;  // [G][128] padded toggle table of the long series (k_train_long) when every list has <= 128 entries
    // k_train_xlong (series beyond 22 528 steps) keeps rank[] in global memory, ordered by (base group, occurrence) instead of by
    // time — the steps a group transition toggles are whole base groups, i.e. contiguous there: its lists hold those positions
    int *d_tog256 = nullptr;  // [G][256] padded toggle table (positions) when every list has <= 256 entries
    int *d_xl_pos = nullptr, *d_xl_time = nullptr;  // [T] position of a time step, time step of a position
    int *d_xl_rows = nullptr, *d_xl_tog = nullptr;  // row_sorted / tog_idx in positions
    int *d_mem_t32 = nullptr;  // [16][G] member slots -> time indexes, two 16-bit indexes per word, when every group has <= 32 members (k_qdm_lanes)
    int *d_mem_t64 = nullptr;  // [32][G] the same for groups of 33 .. 64 members (the regression of k_group_trend on 60-year series)
    bool has_ungrouped = false;  // some time step belongs to no group
    // k_qdm_node_tables (qdm_cell.hpp): one table per distinct length of the ranking rows ([0]: the groups' members, [1]: the
    // windowed rows), made on first use
    struct GroupTabs {
        bool made = false;
        int K = 0, max_n = 0;  // classes (0: no tables), the longest row among them
        size_t bytes = 0;      // of all tables
        std::vector<int> cls_N, cls_off;  // (host: they travel to k_qdm_cell by value)
        int *d_cls_N = nullptr, *d_cls_off = nullptr;
    } gt[2];
};

struct ProfEntry {
    const char *name;
    int64_t count = 0;
    double ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace

struct xsdba_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<std::unique_ptr<Plan>> plans;
    DevBuf scratch[16];
    std::vector<void *> user_allocs;
    std::vector<size_t> user_sizes;                     // bytes of every user_allocs entry
    std::vector<std::pair<size_t, void *>> pool;       // buffers given back and kept for the next request of the same size
    size_t pool_bytes = 0;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t user_ev[16] = {};  // event_record / stream_wait_event slots, created on first use
    int lds_limit = 160 * 1024;
    int n_cu = 256;
    // the per-cell sorts' stability rests on lane-ordered returning LDS atomics (order_probe.hpp): probed at creation; false
    // sends their work to the generic kernels
    bool lane_ordered_atomics = true;
    size_t pool_cap = 0;   // bytes the pool of freed buffers may hold (set at creation from THIS device's memory)
    void *comm = nullptr;  // ncclComm_t once xsdba_hip_comm_init has run
    int comm_rank = -1, comm_size = 0;
};

namespace {

enum Slot { S_IN0 = 0, S_IN1, S_TAB0, S_TAB1, S_TAB2, S_OUT0, S_OUT1, S_CM0, S_CM1, S_CM2, S_Q, S_MISC, S_SCR, S_FLAGS, S_PARK, S_NSLOTS };

void pool_flush(xsdba_hip_ctx *ctx);

// every device allocation of the library: out of memory while freed user buffers sit in the context's pool (xsdba_hip_malloc /
// _free below) gives those back and tries once more
int dev_malloc(xsdba_hip_ctx *ctx, void **p, size_t bytes)
{
    if (hipMalloc(p, bytes) == hipSuccess) return 0;
    (void)hipGetLastError();
    pool_flush(ctx);
    HIP_TRY(hipMalloc(p, bytes));
    return 0;
}

int ensure(xsdba_hip_ctx *ctx, int slot, size_t bytes, void **out)
{
    DevBuf &b = ctx->scratch[slot];
    if (b.cap < bytes) {
        if (b.p) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
        size_t cap = bytes + bytes / 8 + 256;
        int rc = dev_malloc(ctx, &b.p, cap);
        if (rc) return rc;
        b.cap = cap;
    }
    *out = b.p;
    return 0;
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// returns a device view of `p` (nbytes); host data are copied into scratch slot `slot`
int stage_in(xsdba_hip_ctx *ctx, int slot, const void *p, size_t nbytes, const void **dev)
{
    if (is_device_ptr(p)) { *dev = p; return 0; }
    void *d;
    int rc = ensure(ctx, slot, nbytes, &d);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d, p, nbytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return 0;
}

struct OutStage {
    void *user = nullptr, *dev = nullptr;
    size_t nbytes = 0;
    bool copy_back = false;
};

int stage_out(xsdba_hip_ctx *ctx, int slot, void *p, size_t nbytes, OutStage *o)
{
    o->user = p; o->nbytes = nbytes;
    if (!p) { o->dev = nullptr; o->copy_back = false; return 0; }
    if (is_device_ptr(p)) { o->dev = p; o->copy_back = false; return 0; }
    int rc = ensure(ctx, slot, nbytes, &o->dev);
    if (rc) return rc;
    o->copy_back = true;
    return 0;
}

int finish_out(xsdba_hip_ctx *ctx, OutStage *o)
{
    if (o->copy_back) HIP_TRY(hipMemcpyAsync(o->user, o->dev, o->nbytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// ---- profiling ------------------------------------------------------------------------------
ProfEntry &prof_entry(xsdba_hip_ctx *ctx, const char *name)
{
    for (auto &e : ctx->prof) if (e.name == name || !strcmp(e.name, name)) return e;
    ctx->prof.push_back(ProfEntry{name});
    return ctx->prof.back();
}

hipEvent_t get_event(xsdba_hip_ctx *ctx)
{
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct LaunchScope {
    xsdba_hip_ctx *ctx;
    const char *name;
    hipEvent_t a = nullptr, b = nullptr;
    LaunchScope(xsdba_hip_ctx *c, const char *n) : ctx(c), name(n)
    {
        if (ctx->profiling) { a = get_event(ctx); b = get_event(ctx); (void)hipEventRecord(a, ctx->stream); }
    }
    ~LaunchScope()
    {
        if (ctx->profiling) { (void)hipEventRecord(b, ctx->stream); prof_entry(ctx, name).pending.emplace_back(a, b); }
    }
};

int check_launch(const char *name)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "launch of %s failed: %s", name, hipGetErrorString(e));
    return 0;
}

// ---- plans ------------------------------------------------------------------------------------
uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int upload_ints(xsdba_hip_ctx *ctx, const std::vector<int> &v, int **d)
{
    size_t n = std::max<size_t>(v.size(), 1);
    int rc = dev_malloc(ctx, (void **)d, n * sizeof(int));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

void free_plan(Plan *p)
{
    (void)hipFree(p->d_gidx); (void)hipFree(p->d_row_ptr); (void)hipFree(p->d_row_idx);
    (void)hipFree(p->d_mem_ptr); (void)hipFree(p->d_mem_idx);
    (void)hipFree(p->d_row_sorted); (void)hipFree(p->d_tog_ptr); (void)hipFree(p->d_tog_idx);
    if (p->d_tog64) (void)hipFree(p->d_tog64);
    if (p->d_tog64p) (void)hipFree(p->d_tog64p);
    if (p->d_tog256) (void)hipFree(p->d_tog256);
    if (p->d_xl_pos) (void)hipFree(p->d_xl_pos);
    if (p->d_xl_time) (void)hipFree(p->d_xl_time);
    if (p->d_xl_rows) (void)hipFree(p->d_xl_rows);
    if (p->d_xl_tog) (void)hipFree(p->d_xl_tog);
    if (p->d_tog128) (void)hipFree(p->d_tog128);
    if (p->d_mem_t32) (void)hipFree(p->d_mem_t32);
    if (p->d_mem_t64) (void)hipFree(p->d_mem_t64);
    if (p->d_gidx8) (void)hipFree(p->d_gidx8);
    for (auto &t : p->gt) {
        if (t.d_cls_N) (void)hipFree(t.d_cls_N);
        if (t.d_cls_off) (void)hipFree(t.d_cls_off);
    }
}

// The classes of k_qdm_node_tables for the plan's ranking rows: their distinct lengths (at most 16, tables of at most 1 MB
// together — the offsets travel as 16-bit counts of 16 bytes), the most frequent first.
int make_group_tabs(xsdba_hip_ctx *ctx, Plan *pl, bool rows)
{
    Plan::GroupTabs &t = pl->gt[rows ? 1 : 0];
    if (t.made) return 0;
    const std::vector<int> &ptr = rows ? pl->row_ptr : pl->mem_ptr;
    std::vector<std::pair<int, int>> cnt;  // (groups of this length, length)
    for (int g = 0; g < pl->G; g++) {
        const int n = ptr[g + 1] - ptr[g];
        if (n <= 0) continue;
        size_t k = 0;
        while (k < cnt.size() && cnt[k].second != n) k++;
        if (k == cnt.size()) cnt.push_back({0, n});
        cnt[k].first++;
    }
    std::stable_sort(cnt.begin(), cnt.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first > b.first; });
    size_t bytes = 0;
    for (const auto &c : cnt) {
        const size_t len = ((size_t)2 * c.second + 2 + 15) & ~(size_t)15;
        if (t.cls_N.size() >= 16 || bytes + len > (size_t)65535 * 16) break;
        t.cls_N.push_back(c.second);
        t.cls_off.push_back((int)bytes);
        bytes += len;
        t.max_n = std::max(t.max_n, c.second);
    }
    int rc;
    if (!t.cls_N.empty()) {
        if ((rc = upload_ints(ctx, t.cls_N, &t.d_cls_N)) || (rc = upload_ints(ctx, t.cls_off, &t.d_cls_off))) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    t.K = (int)t.cls_N.size();
    t.bytes = bytes;
    t.made = true;
    return 0;
}

// base.py:232-272 — rolling(center=True, window).construct puts x[t' + j - window/2] in slot j of centre t';
// groupby then collects the centres of each group.  row_idx keeps multiplicities (coarse group + window).
int get_plan(xsdba_hip_ctx *ctx, const int32_t *group_of_time, int64_t T, int G, int W, Plan **out)
{
    if (T <= 0 || T > (1ll << 30) || G <= 0 || W <= 0) return fail(XSDBA_E_ARG, "bad n_time/n_groups/window (%lld, %d, %d)", (long long)T, G, W);
    std::vector<int> g((size_t)T, 0);
    if (group_of_time) {
        if (is_device_ptr(group_of_time)) {
            HIP_TRY(hipMemcpyAsync(g.data(), group_of_time, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else memcpy(g.data(), group_of_time, (size_t)T * sizeof(int));
    } else if (G != 1) return fail(XSDBA_E_ARG, "group_of_time is required when n_groups > 1");
    uint64_t h = fnv(g.data(), (size_t)T * sizeof(int));
    int64_t meta[3] = {T, G, W};
    h = fnv(meta, sizeof meta, h);
    for (size_t i = 0; i < ctx->plans.size(); i++) {
        Plan *p = ctx->plans[i].get();
        // (the hash only narrows the search: the group indexes themselves decide)
        if (p->hash == h && p->T == T && p->G == G && p->W == W && !memcmp(p->g_host.data(), g.data(), (size_t)T * sizeof(int))) {
            if (i + 1 != ctx->plans.size()) std::rotate(ctx->plans.begin() + i, ctx->plans.begin() + i + 1, ctx->plans.end());  // most recent last
            *out = ctx->plans.back().get();
            return 0;
        }
    }
    // least recently used plans make room (a plan holds ~2 * T * W ints on the device); nothing is in flight on an evicted
    // plan once the stream has drained
    constexpr size_t MAX_PLANS = 16;
    if (ctx->plans.size() >= MAX_PLANS) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        free_plan(ctx->plans.front().get());
        ctx->plans.erase(ctx->plans.begin());
    }

    // (a plan that fails half-way gives its device arrays back)
    struct PlanDeleter { void operator()(Plan *p) const { free_plan(p); delete p; } };
    std::unique_ptr<Plan, PlanDeleter> pl(new Plan());
    pl->hash = h; pl->T = T; pl->G = G; pl->W = W;
    pl->g_host = g;
    pl->row_ptr.assign((size_t)G + 1, 0);
    pl->mem_ptr.assign((size_t)G + 1, 0);
    const int hw = W / 2;
    for (int64_t t = 0; t < T; t++) {
        int gg = g[(size_t)t];
        if (gg < 0) { pl->has_ungrouped = true; continue; }
        if (gg >= G) return fail(XSDBA_E_ARG, "group_of_time[%lld] = %d out of range [0,%d)", (long long)t, gg, G);
        int64_t lo = std::max<int64_t>(0, t - hw), hi = std::min<int64_t>(T - 1, t - hw + W - 1);
        if ((int64_t)pl->row_ptr[gg + 1] + (hi - lo + 1) > (1ll << 30)) return fail(XSDBA_E_ROW_TOO_LONG, "group row too long");
        pl->row_ptr[gg + 1] += (int)(hi - lo + 1);
        pl->mem_ptr[gg + 1] += 1;
    }
    for (int i = 0; i < G; i++) {
        pl->max_row = std::max(pl->max_row, pl->row_ptr[i + 1]);
        pl->max_mem = std::max(pl->max_mem, pl->mem_ptr[i + 1]);
        if ((int64_t)pl->row_ptr[i] + pl->row_ptr[i + 1] > (1ll << 31) - 2) return fail(XSDBA_E_ROW_TOO_LONG, "windowed rows exceed 2^31 entries");
        pl->row_ptr[i + 1] += pl->row_ptr[i];
        pl->mem_ptr[i + 1] += pl->mem_ptr[i];
    }
    pl->row_idx.resize((size_t)pl->row_ptr[G]);
    pl->mem_idx.resize((size_t)pl->mem_ptr[G]);
    std::vector<int> rp(pl->row_ptr.begin(), pl->row_ptr.end() - 1), mp(pl->mem_ptr.begin(), pl->mem_ptr.end() - 1);
    for (int64_t t = 0; t < T; t++) {
        int gg = g[(size_t)t];
        if (gg < 0) continue;
        pl->mem_idx[(size_t)mp[gg]++] = (int)t;
        for (int j = 0; j < W; j++) {
            int64_t tt = t + j - hw;
            if (tt >= 0 && tt < T) pl->row_idx[(size_t)rp[gg]++] = (int)tt;
        }
    }
    // rows as sorted sets + symmetric differences of consecutive rows (the XOR lists of k_train_cell)
    std::vector<int> row_sorted(pl->row_idx), tog_ptr((size_t)G, 0), tog_idx;
    pl->simple = true;
    for (int i = 0; i < G; i++) {
        auto b = row_sorted.begin() + pl->row_ptr[i], e = row_sorted.begin() + pl->row_ptr[i + 1];
        std::sort(b, e);
        if (std::adjacent_find(b, e) != e) pl->simple = false;
    }
    if (pl->simple) {
        for (int i = 0; i + 1 < G; i++) {
            std::set_symmetric_difference(row_sorted.begin() + pl->row_ptr[i], row_sorted.begin() + pl->row_ptr[i + 1],
                                          row_sorted.begin() + pl->row_ptr[i + 1], row_sorted.begin() + pl->row_ptr[i + 2],
                                          std::back_inserter(tog_idx));
            tog_ptr[(size_t)i + 1] = (int)tog_idx.size();
        }
    }
    int rc;
    if (pl->simple && G > 1) {
        int maxlen = 0;
        for (int i = 0; i + 1 < G; i++) maxlen = std::max(maxlen, tog_ptr[(size_t)i + 1] - tog_ptr[(size_t)i]);
        if (maxlen <= 64) {
            std::vector<int> t64((size_t)G * 64, -1);
            for (int i = 0; i + 1 < G; i++)
                std::copy(tog_idx.begin() + tog_ptr[(size_t)i], tog_idx.begin() + tog_ptr[(size_t)i + 1], t64.begin() + (size_t)(i + 1) * 64);
            if ((rc = upload_ints(ctx, t64, &pl->d_tog64))) return rc;
            // k_train_cell3 flips whatever a lane reads, without a test: the lanes behind a list get PAIRS of the list's own
            // entries (flipped twice = not at all, and spread over the vector instead of all hitting one word), an odd lane out and
            // the rows without a list get time step T — a padding step of the series (the kernel requires T < TPAD), whose
            // "rank" is a bit of a scratch word
            std::vector<int> t64p((size_t)G * 64, (int)T);
            for (int i = 0; i + 1 < G; i++) {
                const int b = tog_ptr[(size_t)i], n = tog_ptr[(size_t)i + 1] - b;
                int *row = t64p.data() + (size_t)(i + 1) * 64;
                std::copy(tog_idx.begin() + b, tog_idx.begin() + b + n, row);
                for (int j = n, e = 0; j + 1 < 64 && e < n; j += 2, e++) row[j] = row[j + 1] = tog_idx[(size_t)b + e];
            }
            if ((rc = upload_ints(ctx, t64p, &pl->d_tog64p))) return rc;
        }
        if (maxlen <= 128 && T > 11 * 1024 && T <= xs::LongCfg<11>::TPAD) {
            // k_train_long's series: 31 to 61 years of daily steps toggle up to ~124 steps per day-of-year group; the padded
            // table is one independent load per 64 entries, a step ahead, where the CSR lists cost a pointer chase per step
            std::vector<int> t128((size_t)G * 128, -1);
            for (int i = 0; i + 1 < G; i++)
                std::copy(tog_idx.begin() + tog_ptr[(size_t)i], tog_idx.begin() + tog_ptr[(size_t)i + 1], t128.begin() + (size_t)(i + 1) * 128);
            if ((rc = upload_ints(ctx, t128, &pl->d_tog128))) return rc;
        }
        if (T > xs::LongCfg<11>::TPAD && T <= 65535) {
            // positions: the members of group 0 in time order, then group 1's, ..., then the steps of no group
            std::vector<int> pos((size_t)T, -1), tim((size_t)T, 0);
            for (size_t k = 0; k < pl->mem_idx.size(); k++) { pos[(size_t)pl->mem_idx[k]] = (int)k; tim[k] = pl->mem_idx[k]; }
            int next = (int)pl->mem_idx.size();
            for (int64_t t = 0; t < T; t++)
                if (pos[(size_t)t] < 0) { pos[(size_t)t] = next; tim[(size_t)next] = (int)t; next++; }
            std::vector<int> rows_p(row_sorted.size()), tog_p(tog_idx.size());
            for (size_t k = 0; k < row_sorted.size(); k++) rows_p[k] = pos[(size_t)row_sorted[k]];
            for (size_t k = 0; k < tog_idx.size(); k++) tog_p[k] = pos[(size_t)tog_idx[k]];
            if ((rc = upload_ints(ctx, pos, &pl->d_xl_pos)) || (rc = upload_ints(ctx, tim, &pl->d_xl_time)) ||
                (rc = upload_ints(ctx, rows_p, &pl->d_xl_rows)) || (rc = upload_ints(ctx, tog_p, &pl->d_xl_tog))) return rc;
            if (maxlen <= 256) {
                std::vector<int> t256((size_t)G * 256, -1);
                for (int i = 0; i + 1 < G; i++)
                    std::copy(tog_p.begin() + tog_ptr[(size_t)i], tog_p.begin() + tog_ptr[(size_t)i + 1], t256.begin() + (size_t)(i + 1) * 256);
                if ((rc = upload_ints(ctx, t256, &pl->d_tog256))) return rc;
            }
        }
    }
    if (G > 1 && pl->max_mem > 32 && pl->max_mem <= 64 && T < 65535) {
        std::vector<int> mt((size_t)32 * G, -1);
        for (int i = 0; i < G; i++)
            for (int y = 0; y < pl->mem_ptr[i + 1] - pl->mem_ptr[i]; y++) {
                const uint32_t t = (uint32_t)pl->mem_idx[(size_t)pl->mem_ptr[i] + y];
                uint32_t w = (uint32_t)mt[(size_t)(y / 2) * G + i];
                w = (y & 1) ? ((w & 0x0000FFFFu) | (t << 16)) : ((w & 0xFFFF0000u) | t);
                mt[(size_t)(y / 2) * G + i] = (int)w;
            }
        if ((rc = upload_ints(ctx, mt, &pl->d_mem_t64))) return rc;
    }
    if (G > 1 && pl->max_mem <= 32 && T < 65535) {
        // member slots of the short exact groups for k_qdm_lanes, slot-pair-major: word [y / 2][g] holds the time indexes of
        // members y and y + 1 of group g, 16 bits each (0xFFFF: no such member)
        std::vector<int> mt((size_t)16 * G, -1);
        for (int i = 0; i < G; i++)
            for (int y = 0; y < pl->mem_ptr[i + 1] - pl->mem_ptr[i]; y++) {
                const uint32_t t = (uint32_t)pl->mem_idx[(size_t)pl->mem_ptr[i] + y];
                uint32_t w = (uint32_t)mt[(size_t)(y / 2) * G + i];
                w = (y & 1) ? ((w & 0x0000FFFFu) | (t << 16)) : ((w & 0xFFFF0000u) | t);
                mt[(size_t)(y / 2) * G + i] = (int)w;
            }
        if ((rc = upload_ints(ctx, mt, &pl->d_mem_t32))) return rc;
    }
    if ((rc = upload_ints(ctx, row_sorted, &pl->d_row_sorted))) return rc;
    if ((rc = upload_ints(ctx, tog_ptr, &pl->d_tog_ptr))) return rc;
    if ((rc = upload_ints(ctx, tog_idx, &pl->d_tog_idx))) return rc;
    if ((rc = upload_ints(ctx, g, &pl->d_gidx))) return rc;
    std::vector<int> g8;  // (declared here: alive until the synchronisation below)
    if (G <= 254) {
        // the group of every time step in a byte (255: none): a gather through it stays in a CU's vector cache
        g8.assign(((size_t)T + 3) / 4, -1);
        unsigned char *b = reinterpret_cast<unsigned char *>(g8.data());
        for (int64_t t = 0; t < T; t++) b[t] = g[t] >= 0 ? (unsigned char)g[t] : (unsigned char)255;
        int *d8 = nullptr;
        if ((rc = upload_ints(ctx, g8, &d8))) return rc;
        pl->d_gidx8 = reinterpret_cast<unsigned char *>(d8);
    }
    if ((rc = upload_ints(ctx, pl->row_ptr, &pl->d_row_ptr))) return rc;
    if ((rc = upload_ints(ctx, pl->row_idx, &pl->d_row_idx))) return rc;
    if ((rc = upload_ints(ctx, pl->mem_ptr, &pl->d_mem_ptr))) return rc;
    if ((rc = upload_ints(ctx, pl->mem_idx, &pl->d_mem_idx))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // host vectors must outlive the copies
    *out = pl.get();
    ctx->plans.push_back(std::unique_ptr<Plan>(pl.release()));
    return 0;
}

// ---- series staging: any (host|device, strides) -> cell-major contiguous device view ---------------
template <typename K> int set_lds(K kernel, size_t bytes, xsdba_hip_ctx *ctx, const char *what);

template <typename E>
int launch_transpose(xsdba_hip_ctx *ctx, const E *src, int64_t rows, int64_t cols, int64_t srs, int64_t scs, E *dst,
                     int64_t drs, int64_t dcs, bool to_cell_major = false)
{
    if (scs == 1 && srs == cols && drs == 1 && dcs == rows && rows >= 64 && cols >= 64) {
        // both sides dense: the large-tile kernel; the tile is long along the side whose rows are not line-aligned (a cell's
        // series), which is `rows` for an input (time-major -> cell-major) and `cols` for an output
        LaunchScope ls(ctx, "k_transpose_dense");
        if (to_cell_major) {
            constexpr int TR = 256, TC = 64, NT = 512;
            auto kern = xs::k_transpose_dense<E, TR, TC, NT>;
            const size_t lds = (size_t)TR * (TC + 1) * sizeof(E);
            dim3 grid((unsigned)((cols + TC - 1) / TC), (unsigned)((rows + TR - 1) / TR));
            if (grid.y > 65535u) return fail(XSDBA_E_ARG, "series too long for the layout kernel");
            int rc = set_lds(kern, lds, ctx, "k_transpose_dense");
            if (rc) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, src, rows, cols, dst);
        } else {
            constexpr int TR = 128, TC = 128, NT = 1024;
            auto kern = xs::k_transpose_dense<E, TR, TC, NT>;
            const size_t lds = (size_t)TR * (TC + 1) * sizeof(E);
            dim3 grid((unsigned)((cols + TC - 1) / TC), (unsigned)((rows + TR - 1) / TR));
            if (grid.y > 65535u) return fail(XSDBA_E_ARG, "too many cells for the layout kernel");
            int rc = set_lds(kern, lds, ctx, "k_transpose_dense");
            if (rc) return rc;
            hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, src, rows, cols, dst);
        }
        return check_launch("k_transpose_dense");
    }
    dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    if (grid.y > 65535u) return fail(XSDBA_E_ARG, "series too long for the layout kernel");
    LaunchScope ls(ctx, "k_transpose_tile");
    hipLaunchKernelGGL(xs::k_transpose_tile<E>, grid, dim3(256), 0, ctx->stream, src, rows, cols, srs, scs, dst, drs, dcs);
    return check_launch("k_transpose_tile");
}

struct SeriesIn {
    const void *dev = nullptr;  // device pointer the kernels read
    int64_t st = 1, sc = 0;
};

template <typename R>
int series_in(xsdba_hip_ctx *ctx, int slot_raw, int slot_cm, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc,
              SeriesIn *out)
{
    if (!x) return fail(XSDBA_E_ARG, "null series pointer");
    if (st <= 0 || sc <= 0) return fail(XSDBA_E_ARG, "strides must be positive");
    // a HOST series is staged as exactly T * C dense elements: anything but the two dense layouts would be read past the copy
    if (!is_device_ptr(x) && !((st == 1 && sc == T) || (sc == 1 && st == C) || (C == 1 && st == 1) || (T == 1 && sc == 1)))
        return fail(XSDBA_E_ARG, "a host series must be dense: (stride_t, stride_c) = (1, n_time) or (n_cell, 1), got (%lld, %lld)",
                    (long long)st, (long long)sc);
    const void *d;
    size_t nbytes = (size_t)T * (size_t)C * sizeof(R);
    int rc = stage_in(ctx, slot_raw, x, nbytes, &d);
    if (rc) return rc;
    if (st == 1) { out->dev = d; out->st = 1; out->sc = sc; return 0; }
    void *cm;
    if ((rc = ensure(ctx, slot_cm, nbytes, &cm))) return rc;
    // rows = time (r), cols = cell (cc): src[t*st + c*sc] -> dst[c*T + t]
    if ((rc = launch_transpose<R>(ctx, (const R *)d, T, C, st, sc, (R *)cm, 1, T, true))) return rc;
    out->dev = cm; out->st = 1; out->sc = T;
    return 0;
}

// output series: kernels always write cell-major; if the caller wants other strides (or a host buffer)
// the result goes through scratch and a layout kernel / copy.
struct SeriesOut {
    void *user = nullptr;
    void *kernel_dst = nullptr;  // cell-major [C][T] device buffer the kernel writes
    void *dev_final = nullptr;   // device buffer in the caller's layout
    int64_t ost = 1, osc = 0;
    size_t nbytes = 0;
    bool need_layout = false, copy_back = false;
};

template <typename E>
int series_out_begin(xsdba_hip_ctx *ctx, int slot_final, int slot_cm, void *user, int64_t T, int64_t C, int64_t ost,
                     int64_t osc, SeriesOut *o)
{
    o->user = user; o->ost = ost; o->osc = osc; o->nbytes = (size_t)T * (size_t)C * sizeof(E);
    if (!user) return 0;
    if (ost <= 0 || osc <= 0) return fail(XSDBA_E_ARG, "output strides must be positive");
    int rc;
    if (is_device_ptr(user)) o->dev_final = user;
    else { if ((rc = ensure(ctx, slot_final, o->nbytes, &o->dev_final))) return rc; o->copy_back = true; }
    if (ost == 1 && osc == T) { o->kernel_dst = o->dev_final; o->need_layout = false; }
    else { if ((rc = ensure(ctx, slot_cm, o->nbytes, &o->kernel_dst))) return rc; o->need_layout = true; }
    return 0;
}

template <typename E> int series_out_end(xsdba_hip_ctx *ctx, SeriesOut *o, int64_t T, int64_t C)
{
    if (!o->user) return 0;
    int rc;
    if (o->need_layout)  // rows = cell, cols = time: src[c*T + t] -> dst[c*osc + t*ost]
        if ((rc = launch_transpose<E>(ctx, (const E *)o->kernel_dst, C, T, T, 1, (E *)o->dev_final, o->osc, o->ost, false))) return rc;
    if (o->copy_back) HIP_TRY(hipMemcpyAsync(o->user, o->dev_final, o->nbytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int sync_if(xsdba_hip_ctx *ctx, bool need)
{
    if (need) HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

template <typename K> int set_lds(K kernel, size_t bytes, xsdba_hip_ctx *ctx, const char *what)
{
    if (bytes > (size_t)ctx->lds_limit) return fail(XSDBA_E_ROW_TOO_LONG, "%s needs %zu B of LDS (> %d)", what, bytes, ctx->lds_limit);
    if (bytes > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

int check_kind(char kind, bool allow_zero)
{
    if (kind == '+' || kind == '*' || (allow_zero && kind == 0)) return 0;
    return fail(XSDBA_E_ARG, "kind must be '+' or '*'");
}

// ---- train ------------------------------------------------------------------------------------------
template <typename R>
int train_impl(xsdba_hip_ctx *ctx, int mode, const void *x0, const void *x1, int64_t T, int64_t C, int64_t st,
               int64_t sc, const int32_t *gidx, int G, int W, const void *q, int Q, char kind, void *af, void *hq,
               void *scaling)
{
    using U = typename xs::KeyOf<R>::U;
    if (C <= 0 || Q <= 0 || !q || !hq || ((mode == 1 || mode == 2) && (!x1 || !af)) || (mode == 2 && !scaling)) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((int64_t)C * G > (1ll << 31) - 1) return fail(XSDBA_E_ARG, "n_cell * n_groups exceeds 2^31");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn s0, s1;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x0, T, C, st, sc, &s0))) return rc;
    if ((mode == 1 || mode == 2) && (rc = series_in<R>(ctx, S_IN1, S_CM1, x1, T, C, st, sc, &s1))) return rc;
    const void *dq;
    if ((rc = stage_in(ctx, S_Q, q, (size_t)(mode == 3 ? C : Q) * sizeof(R), &dq))) return rc;
    size_t tab = (mode == 3 ? (size_t)C : (size_t)C * G * Q) * sizeof(R);
    OutStage oaf, ohq, osc;
    if ((rc = stage_out(ctx, S_TAB0, af, tab, &oaf))) return rc;
    if ((rc = stage_out(ctx, S_TAB1, hq, tab, &ohq))) return rc;
    if ((rc = stage_out(ctx, S_TAB2, scaling, (size_t)C * G * sizeof(R), &osc))) return rc;

    // generic (cell, group)-row kernel; `cells` (device list of n_list cell ids) restricts it to a subset
    // rows that do not fit the LDS sort (hourly series under group="time": 262 800 values): the same pipeline on rows kept in
    // global memory, a batch of rows at a time (train_bigrows.hpp)
    auto launch_rows_big = [&](int P) -> int {
        const int64_t nrows = mode == 3 ? C : (int64_t)C * G;
        const size_t row_bytes = (size_t)P * sizeof(U);
        const int tiles = P / xs::BIG_TILE;
        int64_t batch = std::max<int64_t>(1, (int64_t)(((size_t)1 << 30) / row_bytes));
        batch = std::min<int64_t>({batch, nrows, (int64_t)((1ll << 30) / tiles)});
        const size_t off_nv = ((size_t)batch * row_bytes + 255) & ~(size_t)255;
        const size_t off_mu = (off_nv + (size_t)batch * sizeof(int) + 255) & ~(size_t)255;
        const size_t off_rq = (off_mu + 2 * (size_t)batch * sizeof(R) + 255) & ~(size_t)255;
        void *scr;
        int rc2;
        if ((rc2 = ensure(ctx, S_SCR, off_rq + (size_t)batch * Q * sizeof(R), &scr))) return rc2;
        U *keys = (U *)scr;
        int *nvd = (int *)((unsigned char *)scr + off_nv);
        R *mu_ref = (R *)((unsigned char *)scr + off_mu), *mu_hist = mu_ref + batch;
        R *refq = (R *)((unsigned char *)scr + off_rq);
        const size_t lds_tile = (size_t)xs::BIG_TILE * sizeof(U);
        if ((rc2 = set_lds(xs::k_big_local<U>, lds_tile, ctx, "a tile of the row sort"))) return rc2;
        const int npass = (mode == 0 || mode == 3) ? 1 : 2;
        for (int64_t row0 = 0; row0 < nrows; row0 += batch) {
            const int64_t nb = std::min<int64_t>(batch, nrows - row0);
            for (int pass = 0; pass < npass; pass++) {
                const SeriesIn &sx = pass == 0 ? s0 : s1;
                LaunchScope ls(ctx, "k_train_rows_big");
                hipLaunchKernelGGL(xs::k_big_fill<R>, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, (const R *)sx.dev, sx.st, sx.sc, pl->d_row_ptr,
                                   pl->d_row_idx, mode == 3 ? 1 : G, row0, P, (int)kind, mode == 2 ? 1 : 0, keys, nvd, pass == 0 ? mu_ref : mu_hist);
                hipLaunchKernelGGL(xs::k_big_local<U>, dim3((unsigned)(nb * tiles)), dim3(1024), lds_tile, ctx->stream, keys, P, 0);
                const int64_t npairs = nb * (int64_t)(P / 2);
                for (int k = 2 * xs::BIG_TILE; k <= P && k > 0; k <<= 1) {
                    for (int j = k >> 1; j >= xs::BIG_TILE; j >>= 1)
                        hipLaunchKernelGGL(xs::k_big_step<U>, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, ctx->stream, keys, P, k, j, npairs);
                    hipLaunchKernelGGL(xs::k_big_local<U>, dim3((unsigned)(nb * tiles)), dim3(1024), lds_tile, ctx->stream, keys, P, k);
                }
                hipLaunchKernelGGL(xs::k_big_quant<R>, dim3((unsigned)nb), dim3(64), 0, ctx->stream, (const U *)keys, (const int *)nvd, P,
                                   mode == 3 ? 1 : G, row0, (const R *)dq, Q, (int)kind, mode, pass, (R *)oaf.dev, (R *)ohq.dev, (R *)osc.dev, refq,
                                   (const R *)mu_ref, (const R *)mu_hist);
                if ((rc2 = check_launch("k_train_rows_big"))) return rc2;
            }
        }
        return 0;
    };
    auto launch_rows = [&](const int *cells, const int *n_cells_dev, int64_t n_blocks_or_cells) -> int {
        int P = 2;
        while (P < pl->max_row) P <<= 1;
        int NT = P >= 8192 ? 1024 : (P >= 2048 ? 512 : (P >= 512 ? 256 : (P >= 128 ? 128 : 64)));
        size_t lds = sizeof(double) * (2 * (size_t)(NT / 64) + 2) + (size_t)P * sizeof(U) + (size_t)Q * sizeof(R);
        if (lds > (size_t)ctx->lds_limit && P >= 2 * xs::BIG_TILE) {
            if (cells) return fail(XSDBA_E_UNSUPPORTED, "internal: a per-cell kernel flagged cells of a grouping whose rows exceed the LDS");
            return launch_rows_big(P);
        }
        int rc2;
        if ((rc2 = set_lds(xs::k_train_rows<R>, lds, ctx, "the row sort"))) return rc2;
        LaunchScope ls(ctx, "k_train_rows");
        const int64_t grid = cells ? n_blocks_or_cells : n_blocks_or_cells * G;
        hipLaunchKernelGGL(xs::k_train_rows<R>, dim3((unsigned)grid), dim3(NT), lds, ctx->stream,
                           (const R *)s0.dev, (const R *)s1.dev, s0.st, s0.sc, pl->d_row_ptr, pl->d_row_idx, G,
                           (const R *)dq, Q, (int)kind, mode, (R *)oaf.dev, (R *)ohq.dev, (R *)osc.dev, P, cells, n_cells_dev);
        return check_launch("k_train_rows");
    };
    bool fast = false;
    if constexpr (std::is_same<R, float>::value) {
        // un-windowed groups (group="time" / "time.month" / "time.season" ...): histogram multi-select, train_select.hpp.
        // XSDBA_HIP_TRAIN=rows pins the generic kernel, =sort the per-cell sort kernels (A/B runs, tests).
        const char *force = getenv("XSDBA_HIP_TRAIN");
        const bool sel = (mode == 0 || mode == 1 || mode == 2) && pl->W == 1 && pl->simple && G <= xs::SEL_MAXG &&
                         (int64_t)G * Q <= 4096 && T >= 64 && T <= 22 * 1024 &&
                         !(force && (!strcmp(force, "rows") || !strcmp(force, "sort")));
        if (sel) {
            void *flags;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            // histogram bins over all groups: 2^14 while two blocks per CU still fit in the LDS next to the scattered keys (the
            // kernel lives on its 16 waves per CU), else 2^13
            auto go = [&](auto cfg, auto lt) -> int {
                constexpr int EPT2 = decltype(cfg)::value, LT = decltype(lt)::value;
                constexpr int TP = 2 * EPT2 * xs::SEL_NT;
                const size_t lds = xs::sel_layout(TP, G, Q, LT).bytes;
                const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(EPT2 <= 11 ? 2 : 1, (size_t)ctx->lds_limit / lds));
                const unsigned nblk = (unsigned)std::min<int64_t>(C, (int64_t)per_cu * ctx->n_cu);  // blocks walk the cells grid-stride
                auto launch = [&](auto kernel) -> int {
                    int rc2;
                    if ((rc2 = set_lds(kernel, lds, ctx, "the multi-select tables"))) return rc2;
                    LaunchScope ls(ctx, "k_train_sel");
                    hipLaunchKernelGGL(kernel, dim3(nblk), dim3(xs::SEL_NT), lds, ctx->stream, (const float *)s0.dev,
                                       (const float *)s1.dev, T, s0.sc, pl->d_gidx, G, (const float *)dq, Q, (int)kind, mode,
                                       (float *)oaf.dev, (float *)ohq.dev, (float *)osc.dev, flag_count, flag_list, C);
                    return check_launch("k_train_sel");
                };
                return mode == 2 ? launch(xs::k_train_sel<EPT2, LT, true>) : launch(xs::k_train_sel<EPT2, LT, false>);
            };
            using std::integral_constant;
            auto pick = [&](auto cfg) -> int {
                constexpr int TP = 2 * decltype(cfg)::value * xs::SEL_NT;
                const bool fits14 = xs::sel_layout(TP, G, Q, 14).bytes <= (size_t)ctx->lds_limit / 2 || decltype(cfg)::value > 11;
                return fits14 ? go(cfg, integral_constant<int, 14>{}) : go(cfg, integral_constant<int, 13>{});
            };
            rc = T <= 2048 ? pick(integral_constant<int, 2>{}) : T <= 4096 ? pick(integral_constant<int, 4>{})
               : T <= 8192 ? pick(integral_constant<int, 8>{}) : T <= 11264 ? pick(integral_constant<int, 11>{})
               : T <= 16384 ? pick(integral_constant<int, 16>{}) : pick(integral_constant<int, 22>{});
            if (rc) return rc;
            // cells the select could not finish (long runs of unequal keys in one bin, candidate overflow): generic kernel
            if ((rc = launch_rows(flag_list, flag_count, std::min<int64_t>((int64_t)C * G, 2048)))) return rc;
            if ((rc = finish_out(ctx, &oaf)) || (rc = finish_out(ctx, &ohq)) || (rc = finish_out(ctx, &osc))) return rc;
            return sync_if(ctx, oaf.copy_back || ohq.copy_back || osc.copy_back || !is_device_ptr(x0));
        }
        fast = mode != 3 && pl->simple && T <= 11 * 1024 && T >= 64 && !(force && !strcmp(force, "rows")) && ctx->lane_ordered_atomics;
        if (fast) {
            void *flags;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            const int ept = T <= 4 * 1024 ? 4 : (T <= 8 * 1024 ? 8 : 11);
            // timing aids that skip work (and so return wrong tables) exist in experiment builds only (-DXSDBA_DIAG): the release
            // library never reads the variable
#ifdef XSDBA_DIAG
            const int dbg = getenv("XSDBA_HIP_DEBUG_PHASE") ? atoi(getenv("XSDBA_HIP_DEBUG_PHASE")) : 0;
#else
            const int dbg = 0;
#endif
            auto go = [&](auto cfg, auto diag, auto dqm, auto t64) -> int {
                constexpr int EPT = decltype(cfg)::value;
                constexpr bool DIAG = decltype(diag)::value, DQM = decltype(dqm)::value, TOG64 = decltype(t64)::value;
                size_t lds = xs::CellCfg<EPT>::lds_bytes;
                int rc2;
                if ((rc2 = set_lds(xs::k_train_cell<EPT, DIAG, DQM, TOG64>, lds, ctx, "the per-cell sort"))) return rc2;
                const int gm = getenv("XSDBA_HIP_GRID_MULT") ? atoi(getenv("XSDBA_HIP_GRID_MULT")) : 4;  // tuning aid
                const unsigned nblk = (unsigned)std::min<int64_t>(C, (int64_t)std::max(gm, 1) * ctx->n_cu);  // blocks walk the cells grid-stride
                LaunchScope ls(ctx, "k_train_cell");
                hipLaunchKernelGGL((xs::k_train_cell<EPT, DIAG, DQM, TOG64>), dim3(nblk), dim3(1024), lds, ctx->stream,
                                   (const float *)s0.dev, (const float *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_row_sorted,
                                   pl->d_tog_ptr, pl->d_tog_idx, G, (const float *)dq, Q, (int)kind, mode, (float *)oaf.dev,
                                   (float *)ohq.dev, (float *)osc.dev, flag_count, flag_list, dbg, C, pl->d_tog64);
                return check_launch("k_train_cell");
            };
            using std::integral_constant;
            using std::false_type;
            using std::true_type;
            // the mode switches are template parameters of the kernel: pick the instantiation
            auto pick_t64 = [&](auto cfg, auto dqm) -> int {
                return pl->d_tog64 ? go(cfg, false_type{}, dqm, true_type{}) : go(cfg, false_type{}, dqm, false_type{});
            };
            auto pick_dqm = [&](auto cfg) -> int { return mode == 2 ? pick_t64(cfg, true_type{}) : pick_t64(cfg, false_type{}); };
            // three group streams per wave (train_cell3.hpp): plain / EQM quantiles of up to 21 nodes on windowed day-of-year-like
            // groups (padded toggle table, at least three groups per wave).  XSDBA_HIP_TRAIN3=0 pins the two-stream kernel.
            auto go3 = [&](auto cfg, auto p10, auto dqm, auto nr) -> int {
                constexpr int EPT = decltype(cfg)::value, NR = decltype(nr)::value;
                constexpr bool P10 = decltype(p10)::value, DQM = decltype(dqm)::value;
                const size_t lds = xs::CellCfg3<EPT>::lds_bytes + (DQM ? ((size_t)G + 1) * sizeof(double) : 0);  // + the group sums of dqm_train
                int rc2;
                if ((rc2 = set_lds(xs::k_train_cell3<EPT, P10, DQM, 0, NR>, lds, ctx, "the per-cell sort"))) return rc2;
                const int gm = getenv("XSDBA_HIP_GRID_MULT") ? atoi(getenv("XSDBA_HIP_GRID_MULT")) : 4;
                const unsigned nblk = (unsigned)std::min<int64_t>(C, (int64_t)std::max(gm, 1) * ctx->n_cu);
                LaunchScope ls(ctx, "k_train_cell3");
                hipLaunchKernelGGL((xs::k_train_cell3<EPT, P10, DQM, 0, NR>), dim3(nblk), dim3(1024), lds, ctx->stream, (const float *)s0.dev,
                                   (const float *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_row_sorted, G, (const float *)dq, Q, (int)kind, mode,
                                   (float *)oaf.dev, (float *)ohq.dev, (float *)osc.dev, flag_count, flag_list, C, pl->d_tog64p,
                                   pl->d_mem_ptr, pl->d_mem_idx, (8 * (int64_t)G <= T) ? pl->d_mem_t32 : (const int *)nullptr, pl->W);
                return check_launch("k_train_cell3");
            };
            const char *t3 = getenv("XSDBA_HIP_TRAIN3");
            const size_t lds3 = (ept == 4 ? xs::CellCfg3<4>::lds_bytes : ept == 8 ? xs::CellCfg3<8>::lds_bytes : xs::CellCfg3<11>::lds_bytes) +
                                (mode == 2 ? ((size_t)G + 1) * sizeof(double) : 0);
            // (T < TPAD: the kernel uses time step T, a padding step of the series in LDS, as the "no toggle" entry of its table)
            // (22 to 63 nodes: two or three rounds of the select per group step, instantiated for the 10-bit packed counts only)
            const bool three = !dbg && pl->d_tog64p && (Q <= 21 || (Q <= 63 && pl->max_row <= 1023)) && G >= 48 && T < (int64_t)ept * 1024 &&
                               lds3 <= (size_t)ctx->lds_limit && !(t3 && !strcmp(t3, "0"));
            auto pick3d = [&](auto cfg, auto p10, auto nr) -> int { return mode == 2 ? go3(cfg, p10, true_type{}, nr) : go3(cfg, p10, false_type{}, nr); };
            auto pick3 = [&](auto cfg) -> int {
                if (Q > 42) return pick3d(cfg, true_type{}, integral_constant<int, 3>{});
                if (Q > 21) return pick3d(cfg, true_type{}, integral_constant<int, 2>{});
                return pl->max_row <= 1023 ? pick3d(cfg, true_type{}, integral_constant<int, 1>{}) : pick3d(cfg, false_type{}, integral_constant<int, 1>{});
            };
            if (three)
                rc = ept == 4 ? pick3(integral_constant<int, 4>{}) : ept == 8 ? pick3(integral_constant<int, 8>{}) : pick3(integral_constant<int, 11>{});
            else if (dbg && ept == 11 && mode != 2 && pl->d_tog64)  // timing aids (XSDBA_HIP_DEBUG_PHASE): diagnostic build
                rc = go(integral_constant<int, 11>{}, true_type{}, false_type{}, true_type{});
            else rc = ept == 4 ? pick_dqm(integral_constant<int, 4>{})
                 : ept == 8 ? pick_dqm(integral_constant<int, 8>{})
                            : pick_dqm(integral_constant<int, 11>{});
            if (rc) return rc;
            // cells whose shortcut assumptions failed (rare) are redone by the generic kernel: a small fixed grid
            // walks the flagged rows, reading the count on the device (no host round trip, the call stays async)
            if ((rc = launch_rows(flag_list, flag_count, std::min<int64_t>((int64_t)C * G, 2048)))) return rc;
        }
    }
    if constexpr (std::is_same<R, float>::value) {
        // 11 265 .. 22 528 time steps (e.g. 60 years daily): split sort + merge, train_long.hpp
        const char *force = getenv("XSDBA_HIP_TRAIN");
        const bool fast_long = !fast && mode != 3 && pl->simple && T > 11 * 1024 && T <= xs::LongCfg<11>::TPAD && G > 1 &&
                               !(force && !strcmp(force, "rows")) && ctx->lane_ordered_atomics;
        if (fast_long) {
            void *flags, *scratch;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            const unsigned nblk = (unsigned)std::min<int64_t>(C, 2 * ctx->n_cu);
            if ((rc = ensure(ctx, S_SCR, xs::LongCfg<11>::scratch_bytes * nblk, &scratch))) return rc;
            const bool short_halves = T <= xs::LongCfg<8>::TPAD;  // up to 16 384 steps: halves of at most 8 192
            auto go2 = [&](auto dqm, auto t64, auto ept) -> int {
                constexpr bool DQM = decltype(dqm)::value, TOG64 = decltype(t64)::value;
                constexpr int EPT = decltype(ept)::value;
                const size_t lds = xs::LongCfg<EPT>::lds_bytes;
                int rc2;
                if ((rc2 = set_lds(xs::k_train_long<DQM, TOG64, EPT>, lds, ctx, "the per-cell sort"))) return rc2;
                LaunchScope ls(ctx, "k_train_long");
                hipLaunchKernelGGL((xs::k_train_long<DQM, TOG64, EPT>), dim3(nblk), dim3(1024), lds, ctx->stream, (const float *)s0.dev,
                                   (const float *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_row_sorted, pl->d_tog_ptr, pl->d_tog_idx, G,
                                   (const float *)dq, Q, (int)kind, mode, (float *)oaf.dev, (float *)ohq.dev, (float *)osc.dev,
                                   flag_count, flag_list, C, pl->d_tog128, (unsigned char *)scratch,
#ifdef XSDBA_DIAG
                                   getenv("XSDBA_HIP_DEBUG_PHASE") ? atoi(getenv("XSDBA_HIP_DEBUG_PHASE")) : 0);
#else
                                   0);
#endif
                return check_launch("k_train_long");
            };
            auto go = [&](auto dqm, auto t64) -> int {
                return short_halves ? go2(dqm, t64, std::integral_constant<int, 8>{}) : go2(dqm, t64, std::integral_constant<int, 11>{});
            };
            using std::false_type;
            using std::true_type;
            rc = mode == 2 ? (pl->d_tog128 ? go(true_type{}, true_type{}) : go(true_type{}, false_type{}))
                           : (pl->d_tog128 ? go(false_type{}, true_type{}) : go(false_type{}, false_type{}));
            if (rc) return rc;
            if ((rc = launch_rows(flag_list, flag_count, std::min<int64_t>((int64_t)C * G, 2048)))) return rc;
            fast = true;
        }
        // 22 529 .. 65 535 time steps: chunk sorts merged in global scratch, train_xlong.hpp
        const bool fast_xlong = !fast && mode != 3 && pl->simple && pl->d_xl_pos && T > xs::LongCfg<11>::TPAD && T <= 65535 && G > 1 &&
                                !(force && !strcmp(force, "rows")) && ctx->lane_ordered_atomics;
        if (fast_xlong) {
            void *flags, *scratch;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            const unsigned nblk = (unsigned)std::min<int64_t>(C, ctx->n_cu);  // one block per CU (155 KB of LDS), cells walked grid-stride
            auto go = [&](auto dqm, auto togp, auto wpl, auto ce) -> int {
                constexpr bool DQM = decltype(dqm)::value, TOGP = decltype(togp)::value;
                constexpr int WPL = decltype(wpl)::value, CE = decltype(ce)::value;
                using X = xs::XLongGeo<WPL>;
                int rc2;
                if ((rc2 = ensure(ctx, S_SCR, X::scratch_bytes * nblk, &scratch))) return rc2;
                if ((rc2 = set_lds(xs::k_train_xlong<DQM, TOGP, WPL, CE>, X::lds_bytes, ctx, "the per-cell sort"))) return rc2;
                LaunchScope ls(ctx, "k_train_xlong");
                hipLaunchKernelGGL((xs::k_train_xlong<DQM, TOGP, WPL, CE>), dim3(nblk), dim3(1024), X::lds_bytes, ctx->stream, (const float *)s0.dev,
                                   (const float *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_xl_rows, pl->d_tog_ptr, pl->d_xl_tog, G,
                                   (const float *)dq, Q, (int)kind, mode, (float *)oaf.dev, (float *)ohq.dev, (float *)osc.dev,
                                   flag_count, flag_list, C, pl->d_tog256, pl->d_xl_pos, pl->d_xl_time, (unsigned char *)scratch,
#ifdef XSDBA_DIAG
                                   getenv("XSDBA_HIP_DEBUG_PHASE") ? atoi(getenv("XSDBA_HIP_DEBUG_PHASE")) : 0);
#else
                                   0);
#endif
                return check_launch("k_train_xlong");
            };
            using std::false_type;
            using std::true_type;
            using std::integral_constant;
            auto pick_w = [&](auto dqm, auto togp) -> int {
                // (up to 32 768 steps: two chunks of at most 16 384, sorted with sixteen keys per thread instead of twenty-two)
                if (T <= 2 * xs::LongCfg<8>::TPAD) return go(dqm, togp, integral_constant<int, 22>{}, integral_constant<int, 8>{});
                return T <= xs::XLongGeo<22>::TPAD ? go(dqm, togp, integral_constant<int, 22>{}, integral_constant<int, 11>{})
                                                   : go(dqm, togp, integral_constant<int, 32>{}, integral_constant<int, 11>{});
            };
            rc = mode == 2 ? (pl->d_tog256 ? pick_w(true_type{}, true_type{}) : pick_w(true_type{}, false_type{}))
                           : (pl->d_tog256 ? pick_w(false_type{}, true_type{}) : pick_w(false_type{}, false_type{}));
            if (rc) return rc;
            if ((rc = launch_rows(flag_list, flag_count, std::min<int64_t>((int64_t)C * G, 2048)))) return rc;
            fast = true;
        }
    }
    if constexpr (std::is_same<R, double>::value) {
        // float64 series: 31-bit range keys + tie repair, train_cell_f64.hpp
        const char *force = getenv("XSDBA_HIP_TRAIN");
        const bool fast64 = mode != 3 && pl->simple && T <= 11 * 1024 && T >= 64 && !(force && !strcmp(force, "rows")) && ctx->lane_ordered_atomics;
        if (fast64) {
            void *flags;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            const int ept = T <= 4 * 1024 ? 4 : (T <= 8 * 1024 ? 8 : 11);
            auto go = [&](auto cfg, auto dqm, auto t64) -> int {
                constexpr int EPT = decltype(cfg)::value;
                constexpr bool DQM = decltype(dqm)::value, TOG64 = decltype(t64)::value;
                const size_t lds = xs::CellCfg<EPT>::lds_bytes;
                int rc2;
                if ((rc2 = set_lds(xs::k_train_cell_f64<EPT, DQM, TOG64>, lds, ctx, "the per-cell sort"))) return rc2;
                const unsigned nblk = (unsigned)std::min<int64_t>(C, 4 * ctx->n_cu);
                LaunchScope ls(ctx, "k_train_cell_f64");
                hipLaunchKernelGGL((xs::k_train_cell_f64<EPT, DQM, TOG64>), dim3(nblk), dim3(1024), lds, ctx->stream,
                                   (const double *)s0.dev, (const double *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_row_sorted,
                                   pl->d_tog_ptr, pl->d_tog_idx, G, (const double *)dq, Q, (int)kind, mode, (double *)oaf.dev,
                                   (double *)ohq.dev, (double *)osc.dev, flag_count, flag_list, C, pl->d_tog64);
                return check_launch("k_train_cell_f64");
            };
            using std::false_type;
            using std::integral_constant;
            using std::true_type;
            auto pick_t64 = [&](auto cfg, auto dqm) -> int { return pl->d_tog64 ? go(cfg, dqm, true_type{}) : go(cfg, dqm, false_type{}); };
            auto pick_dqm = [&](auto cfg) -> int { return mode == 2 ? pick_t64(cfg, true_type{}) : pick_t64(cfg, false_type{}); };
            // three group streams per wave, as for float32 (train_cell3.hpp)
            auto go3 = [&](auto cfg, auto p10, auto dqm) -> int {
                constexpr int EPT = decltype(cfg)::value;
                constexpr bool P10 = decltype(p10)::value, DQM = decltype(dqm)::value;
                const size_t lds = xs::CellCfg3<EPT>::lds_bytes;
                int rc2;
                if ((rc2 = set_lds(xs::k_train_cell3_f64<EPT, P10, DQM>, lds, ctx, "the per-cell sort"))) return rc2;
                const unsigned nblk = (unsigned)std::min<int64_t>(C, 4 * ctx->n_cu);
                LaunchScope ls(ctx, "k_train_cell3_f64");
                hipLaunchKernelGGL((xs::k_train_cell3_f64<EPT, P10, DQM>), dim3(nblk), dim3(1024), lds, ctx->stream, (const double *)s0.dev,
                                   (const double *)s1.dev, T, s0.sc, pl->d_row_ptr, pl->d_row_sorted, G, (const double *)dq, Q, (int)kind,
                                   mode, (double *)oaf.dev, (double *)ohq.dev, (double *)osc.dev, flag_count, flag_list, C, pl->d_tog64);
                return check_launch("k_train_cell3_f64");
            };
            auto pick3d = [&](auto cfg, auto p10) -> int { return mode == 2 ? go3(cfg, p10, true_type{}) : go3(cfg, p10, false_type{}); };
            auto pick3 = [&](auto cfg) -> int { return pl->max_row <= 1023 ? pick3d(cfg, true_type{}) : pick3d(cfg, false_type{}); };
            const char *t3 = getenv("XSDBA_HIP_TRAIN3");
            const bool three = pl->d_tog64 && Q <= 21 && G >= 48 && !(t3 && !strcmp(t3, "0"));
            if (three)
                rc = ept == 4 ? pick3(integral_constant<int, 4>{}) : ept == 8 ? pick3(integral_constant<int, 8>{}) : pick3(integral_constant<int, 11>{});
            else
            rc = ept == 4 ? pick_dqm(integral_constant<int, 4>{}) : ept == 8 ? pick_dqm(integral_constant<int, 8>{})
                                                                              : pick_dqm(integral_constant<int, 11>{});
            if (rc) return rc;
            if ((rc = launch_rows(flag_list, flag_count, std::min<int64_t>((int64_t)C * G, 2048)))) return rc;
            fast = true;
        }
        // (float64 series beyond 11 264 steps take the generic row kernel below: k_train_long / k_train_xlong are float32 kernels —
        //  a copy of the float32 branch's xlong launch stood here since round 3 and handed them doubles: found by
        //  scripts/diag/fuzz_vs_oracle.py in round 5, tests/test_xlong_series_gpu.py now runs float64 too)
    }
    if (!fast && (rc = launch_rows(nullptr, nullptr, C))) return rc;
    if ((rc = finish_out(ctx, &oaf)) || (rc = finish_out(ctx, &ohq)) || (rc = finish_out(ctx, &osc))) return rc;
    return sync_if(ctx, oaf.copy_back || ohq.copy_back || osc.copy_back || !is_device_ptr(x0));
}

// ---- adjust -----------------------------------------------------------------------------------------
// threads per block of the adjust kernels (XSDBA_HIP_ADJ_THREADS overrides, for tuning runs)
inline unsigned adj_threads()
{
    const char *e = getenv("XSDBA_HIP_ADJ_THREADS");
    int v = e ? atoi(e) : 512;  // 512: measured best on MI355X (5.8 ms vs 10.6 at 256 / 6.5 at 1024, 64800 cells doy tables)
    return (unsigned)(v >= 64 && v <= 1024 ? (v / 64) * 64 : 512);
}

inline unsigned qdm_threads()
{
    const char *e = getenv("XSDBA_HIP_QDM_THREADS");
    int v = e ? atoi(e) : 1024;
    return (unsigned)(v >= 64 && v <= 1024 ? (v / 64) * 64 : 1024);
}
inline unsigned trend_threads()
{
    const char *e = getenv("XSDBA_HIP_TREND_THREADS");
    int v = e ? atoi(e) : 512;
    return (unsigned)(v >= 64 && v <= 1024 ? (v / 64) * 64 : 512);
}

template <typename R>
int launch_qm_adjust(xsdba_hip_ctx *ctx, const SeriesIn &s, int64_t T, int64_t C, const R *af, const R *hq, int G, int Q,
                     const int *d_gidx, int interp, int extrap, char kind, R *scen_cm, int *node, const R *scaling,
                     const R *trend_cm, int tkind = 0, int skip0 = 0, xs::TrendCoef tc = xs::TrendCoef())
{
    if (!tkind) tkind = (int)kind;
    const size_t tab0 = (xs::tab_lds_bytes<R>(G, Q) + 15) & ~(size_t)15;
    const size_t tab = tab0 + (tc.coef ? (size_t)3 * G * sizeof(double) + (((size_t)G * sizeof(float) + 15) & ~(size_t)15) : 0);  // + the regression coefficients and the scaling factors of the cell
    const char *name = scaling ? "k_qm_adjust(dqm)" : "k_qm_adjust";
    if (s.st != 1) return fail(XSDBA_E_ARG, "internal: the adjust kernels expect time-contiguous rows");
    // mode switches are template parameters of the kernel: pick the instantiation
    auto go = [&](auto grouped, auto dqm) -> int {
        constexpr bool GROUPED = decltype(grouped)::value, DQM = decltype(dqm)::value;
        int rc;
        if (tab <= (size_t)ctx->lds_limit) {
            if ((rc = set_lds(xs::k_qm_adjust<R, true, GROUPED, DQM>, tab, ctx, "the factor table"))) return rc;
            LaunchScope ls(ctx, name);
            // big tables leave room for two blocks per CU only: 12 waves each hide the table-staging and gather latencies best
            // (measured on the doy table, 68.6 KB: 512 threads 3.51 ms, 640 3.81, 768 2.92, 896 4.32, 1024 3.99; small tables
            // keep 512: many blocks per CU already)
            // (small tables — un-windowed groups — run best with 256 threads: 1.95 vs 2.16 ms at 512 on the monthly table)
            const unsigned nthr = getenv("XSDBA_HIP_ADJ_THREADS") ? adj_threads() : (tab > 54 * 1024 ? 768u : (tab <= 16 * 1024 ? 256u : adj_threads()));
            hipLaunchKernelGGL((xs::k_qm_adjust<R, true, GROUPED, DQM>), dim3((unsigned)C), dim3(nthr), tab, ctx->stream,
                               (const R *)s.dev, T, C, s.sc, af, hq, G, Q, d_gidx, interp, extrap, (int)kind, scen_cm, T, node,
                               scaling, trend_cm, (unsigned char *)nullptr, (size_t)0, tkind, skip0, tc);
        } else {  // table larger than the LDS: per-block slice of a global scratch buffer, cells walked grid-stride
            const unsigned nblk = (unsigned)std::min<int64_t>(C, 1024);
            void *scratch;
            if (tc.coef) return fail(XSDBA_E_ARG, "internal: coefficient trends need the table in LDS");
            if ((rc = ensure(ctx, S_SCR, tab * nblk, &scratch))) return rc;
            LaunchScope ls(ctx, name);
            hipLaunchKernelGGL((xs::k_qm_adjust<R, false, GROUPED, DQM>), dim3(nblk), dim3(256), 0, ctx->stream, (const R *)s.dev, T,
                               C, s.sc, af, hq, G, Q, d_gidx, interp, extrap, (int)kind, scen_cm, T, node, scaling, trend_cm,
                               (unsigned char *)scratch, tab, tkind, skip0);
        }
        return check_launch("k_qm_adjust");
    };
    using std::false_type;
    using std::true_type;
    if (G > 1) return scaling ? go(true_type{}, true_type{}) : go(true_type{}, false_type{});
    return scaling ? go(false_type{}, true_type{}) : go(false_type{}, false_type{});
}

template <typename R>
int qm_adjust_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                   const void *hq, int G, int Q, const int32_t *gidx, int interp, int extrap, char kind, void *scen,
                   int64_t ost, int64_t osc, int32_t *node)
{
    if (C <= 0 || Q <= 0 || !af || !hq || !scen) return fail(XSDBA_E_ARG, "null/empty argument");
    if (G > 1 && interp != XSDBA_INTERP_NEAREST)
        return fail(XSDBA_E_UNSUPPORTED, "grouped interpolation other than 'nearest' is not implemented (Qhull triangulation in the reference)");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, 1, &pl))) return rc;
    size_t tab = (size_t)C * G * Q * sizeof(R);
    if constexpr (std::is_same<R, float>::value) {
        // (time, cell) series in and out, both on the device: mapped in place, a lane per cell (adjust_tm.hpp) — no layout pass
        const bool tm = C > 1 && st == C && sc == 1 && ost == C && osc == 1 && is_device_ptr(sim) && is_device_ptr(scen) && G > 1 && !node &&
                        !pl->has_ungrouped && Q <= 64 && xs::tm_lds_bytes(Q) <= (size_t)ctx->lds_limit && !getenv("XSDBA_HIP_NO_TM_ADJUST");
        if (tm) {
            const void *daf, *dhq;
            if ((rc = stage_in(ctx, S_TAB0, af, tab, &daf)) || (rc = stage_in(ctx, S_TAB1, hq, tab, &dhq))) return rc;
            const int gpw = std::max(1, std::min(8, G / 16));
            const int n_runs = (G + gpw - 1) / gpw;
            const int64_t ntile = ((C + 63) / 64 + xs::TM_WAVES - 1) / xs::TM_WAVES;  // quadruples of tiles: a block each
            const size_t lds = xs::tm_lds_bytes(Q);
            auto launch = [&](auto kernel) -> int {
                int rc2;
                if ((rc2 = set_lds(kernel, lds, ctx, "the lanes' table rows"))) return rc2;
                LaunchScope ls(ctx, "k_qm_adjust_tm");
                hipLaunchKernelGGL(kernel, dim3((unsigned)(ntile * n_runs)), dim3(xs::TM_WAVES * 64), lds, ctx->stream, (const float *)sim, T, C,
                                   (const float *)daf, (const float *)dhq, G, Q, pl->d_mem_ptr, pl->d_mem_idx, gpw, n_runs, extrap, (int)kind,
                                   (float *)scen);
                return check_launch("k_qm_adjust_tm");
            };
            // (load instructions per table and group: ceil(64 / (64 / (Q / 4))), see the kernel)
            rc = Q <= 20 ? launch(xs::k_qm_adjust_tm<6>) : (Q <= 32 ? launch(xs::k_qm_adjust_tm<8>) : launch(xs::k_qm_adjust_tm<16>));
            if (rc) return rc;
            return sync_if(ctx, false);
        }
    }
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    const void *daf, *dhq;
    if ((rc = stage_in(ctx, S_TAB0, af, tab, &daf)) || (rc = stage_in(ctx, S_TAB1, hq, tab, &dhq))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    OutStage on;
    if ((rc = stage_out(ctx, S_MISC, node, (size_t)C * T * sizeof(int), &on))) return rc;
    if ((rc = launch_qm_adjust<R>(ctx, s, T, C, (const R *)daf, (const R *)dhq, G, Q, pl->d_gidx, interp, extrap, kind,
                                  (R *)o.kernel_dst, (int *)on.dev, nullptr, nullptr))) return rc;
    if ((rc = series_out_end<R>(ctx, &o, T, C)) || (rc = finish_out(ctx, &on))) return rc;
    return sync_if(ctx, o.copy_back || on.copy_back || !is_device_ptr(sim));
}

// grouped interp="linear": Delaunay triangulation of every cell's table nodes + piecewise-linear evaluation (lin2d.hpp)
template <typename R>
int qm_adjust_linear2d_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                            const void *hq, int G, int Q, const double *gcoord, const double *newg, int extrap, char kind,
                            const void *trend, const double *lookup, int hq_is_row, void *scen, int64_t ost, int64_t osc, int cubic = 0)
{
    if (C <= 0 || T <= 0 || Q <= 0 || G < 2 || !af || !hq || !scen || !gcoord || !newg) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((int64_t)(G + 2) * Q > 32000) return fail(XSDBA_E_UNSUPPORTED, "more than 32000 table nodes per cell (%d groups x %d quantiles)", G, Q);
    int rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    const size_t tab = (size_t)C * G * Q * sizeof(R);
    const void *daf, *dhq, *dgc, *dng, *dtr = nullptr, *dlk = nullptr;
    if ((rc = stage_in(ctx, S_TAB0, af, tab, &daf)) || (rc = stage_in(ctx, S_TAB1, hq, hq_is_row ? (size_t)Q * sizeof(R) : tab, &dhq)) ||
        (rc = stage_in(ctx, S_Q, gcoord, sizeof(double) * (size_t)(G + 2), &dgc)) ||
        (rc = stage_in(ctx, S_TAB2, newg, sizeof(double) * (size_t)T, &dng))) return rc;
    if (trend && (rc = stage_in(ctx, S_OUT1, trend, (size_t)T * C * sizeof(R), &dtr))) return rc;
    if (lookup && (rc = stage_in(ctx, S_CM2, lookup, (size_t)T * C * sizeof(double), &dlk))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    const xs::Lin2DLayout L = xs::lin2d_layout(G, Q, cubic != 0);
    // cells in chunks so that the triangulations of a chunk take at most ~4 GiB of scratch
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(C, (int64_t)((4ull << 30) / L.total)));
    // strip lists (lin2d.hpp) keep the nodes as float32 pairs: float32 tables on row coordinates that float32 holds exactly
    int strips = std::is_same<R, float>::value && !is_device_ptr(gcoord) ? 1 : 0;  // (row coordinates on the device: no host check, the walk)
    for (int r = 0; r < G + 2 && strips; r++) strips = (double)(float)gcoord[r] == gcoord[r];
    if (getenv("XSDBA_HIP_LIN2D_WALK")) strips = 0;  // (A/B runs: the walk of rounds 1-3)
    void *scratch;
    if ((rc = ensure(ctx, S_SCR, L.total * (size_t)chunk, &scratch))) return rc;
    const size_t lds_staged = xs::lin2d_lds_staged(L, cubic != 0);
    const int use_lds = lds_staged <= 64 * 1024 ? 1 : 0;
    void *jrow;  // the row bracket of every time step, shared by all cells
    if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * (size_t)T, &jrow))) return rc;
    {
        LaunchScope ls(ctx, "k_lin2d_rows");
        hipLaunchKernelGGL(xs::k_lin2d_rows, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)dng,
                           (const double *)dgc, G + 2, T, (int *)jrow);
    }
    if ((rc = check_launch("k_lin2d_rows"))) return rc;
    if (use_lds && (rc = set_lds(xs::k_lin2d_adjust<R, true>, lds_staged, ctx, "the triangulated table"))) return rc;
    for (int64_t c0 = 0; c0 < C; c0 += chunk) {
        const int64_t nc = std::min<int64_t>(chunk, C - c0);
        // the triangulations: a workgroup per cell (row construction + parallel flips, k_lin2d_build_rows); XSDBA_HIP_LIN2D_BUILD=sweep
        // pins the one-thread-per-cell sweep of rounds 1-3 (A/B runs)
        const char *bsel = getenv("XSDBA_HIP_LIN2D_BUILD");
        if (bsel && !strcmp(bsel, "sweep")) {
            LaunchScope ls(ctx, "k_lin2d_build");
            hipLaunchKernelGGL(xs::k_lin2d_build<R>, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, ctx->stream,
                               (const R *)dhq + (hq_is_row ? 0 : (size_t)c0 * G * Q), (const R *)daf + (size_t)c0 * G * Q, (const double *)dgc, G, Q, nc,
                               (unsigned char *)scratch, L.total, hq_is_row ? 1 : 0, cubic);
        } else {
            {
                // small tables are triangulated in LDS (64 B per node + the row tables): a wave per cell, as many cells per CU as fit
                const size_t lds_small = xs::lin2d_rows_lds(G, Q, true);
                const bool in_lds = lds_small <= (size_t)ctx->lds_limit;  // (up to ~2 400 nodes; day-of-year tables work in global memory)
                if (in_lds && (rc = set_lds(xs::k_lin2d_build_rows<R, true>, lds_small, ctx, "the table's triangulation"))) return rc;
                LaunchScope ls(ctx, "k_lin2d_build_rows");
                const char *bt = getenv("XSDBA_HIP_LIN2D_THREADS");  // (tuning aid)
                const unsigned nthr = bt ? (unsigned)std::min(1024, std::max(64, atoi(bt) / 64 * 64)) : (in_lds && (int64_t)(G + 2) * Q <= 640 ? 64u : 256u);
                const int per_cu = in_lds ? (int)std::max<size_t>(1, std::min<size_t>(32, (size_t)ctx->lds_limit / lds_small)) : 8;
                const unsigned nblk = (unsigned)std::min<int64_t>(nc, (int64_t)per_cu * ctx->n_cu);
                auto launch = [&](auto k, size_t lds) {
                    hipLaunchKernelGGL(k, dim3(nblk), dim3(nthr), lds, ctx->stream,
                                       (const R *)dhq + (hq_is_row ? 0 : (size_t)c0 * G * Q), (const R *)daf + (size_t)c0 * G * Q, (const double *)dgc, G, Q,
                                       nc, (unsigned char *)scratch, L.total, hq_is_row ? 1 : 0, cubic, strips);
                };
                if (in_lds) launch(xs::k_lin2d_build_rows<R, true>, lds_small);
                else launch(xs::k_lin2d_build_rows<R, false>, xs::lin2d_rows_lds(G, Q, false));
            }
            if ((rc = check_launch("k_lin2d_build_rows"))) return rc;
            if (cubic) {
                LaunchScope ls(ctx, "k_lin2d_grad");
                // (node records once the launch is wide enough to be bound by the L2: about a wave per SIMD)
                hipLaunchKernelGGL(xs::k_lin2d_grad, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, ctx->stream, G, Q, nc, (unsigned char *)scratch, L.total,
                                   nc >= (int64_t)128 * ctx->n_cu ? 1 : 0);
            }
        }
        if ((rc = check_launch("k_lin2d_build"))) return rc;
        {
            LaunchScope ls(ctx, "k_lin2d_adjust");
            const unsigned nblk = (unsigned)std::min<int64_t>(nc, 8 * ctx->n_cu);
            auto launch = [&](auto k, size_t lds) {
                hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, ctx->stream, (const R *)s.dev + c0 * s.sc, T, nc, s.sc,
                                   (const double *)dng, (const int *)jrow, G, Q, (const unsigned char *)scratch, L.total, extrap,
                                   (int)kind, dtr ? (const R *)dtr + (size_t)c0 * T : (const R *)nullptr,
                                   (R *)o.kernel_dst + (size_t)c0 * T, T, dlk ? (const double *)dlk + (size_t)c0 * T : (const double *)nullptr, cubic);
            };
            if (use_lds) launch(xs::k_lin2d_adjust<R, true>, lds_staged);
            else launch(xs::k_lin2d_adjust<R, false>, 0);
        }
        if ((rc = check_launch("k_lin2d_adjust"))) return rc;
    }
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, true);
}

template <typename R>
int launch_trend(xsdba_hip_ctx *ctx, const SeriesIn &s, int64_t T, int64_t C, const R *scaling, Plan *pl, char kind,
                 const double *d_tnum, int degree, R *trend_cm, int preserve_mean = 0, int min_periods = 0,
                 const double *h_tnum = nullptr, double *coef_out = nullptr, xs::TrendCoef *tc_out = nullptr)
{
    // a time axis that is affine in the step index (daily data in ns since an epoch: exact integers in float64) is computed
    // in the kernel instead of gathered: checked here on the host copy, value for value, with the kernel's own expression
    int tn_affine = 0;
    double tn0 = 0.0, tnd = 0.0;
    if (h_tnum && T >= 2) {
        tn0 = h_tnum[0]; tnd = h_tnum[1] - h_tnum[0];
        tn_affine = 1;
        for (int64_t t = 0; t < T; t++)
            if (!(tn0 + tnd * (double)t == h_tnum[t])) { tn_affine = 0; break; }
    }
    if (tc_out) { tc_out->coef = coef_out; tc_out->tnum = d_tnum; tc_out->affine = tn_affine; tc_out->t0 = tn0; tc_out->dt = tnd; }
    const size_t ncoef = (degree >= 2 || preserve_mean) ? (size_t)degree + 4 : 3;
    // the in-register window means hold TREND_CH values per thread: long series take 1024 threads to stay on that path
    const unsigned nthr = (!getenv("XSDBA_HIP_TREND_THREADS") && T > xs::TREND_CH * 512 && T <= xs::TREND_CH * 1024) ? 1024u : trend_threads();
    const size_t scratch_d = 2 * (size_t)pl->G <= (size_t)(nthr >> 5) ? (size_t)xs::TREND_SCRATCH_D : 0;  // (k_group_trend: half-waves sharing a group)
    const size_t lds1 = sizeof(double) * (ncoef * (size_t)pl->G + scratch_d + (size_t)pl->W + 1) + (size_t)T * sizeof(R);
    const size_t lds2 = lds1 + (size_t)T * sizeof(R);
    const bool m_lds = getenv("XSDBA_HIP_TREND_MLDS") != nullptr;  // window means in LDS: measured slower (occupancy)
    int rc;
    // regression layout: a lane per group when there are many short groups (day-of-year), else a half-wave per group
    const int lane_per_group = (pl->G >= 128 && pl->max_mem <= 256) ? 1 : 0;
    if (lds1 > (size_t)ctx->lds_limit) {
        // the series does not fit the LDS (hourly data; 4 x slower per step than the LDS kernel where both apply: 17.5 against
        // 4.4 ms per 32 400 cells x 21 900 days): the kernel reads it from global memory (k_group_trend<XG>), the window
        // means go through the trend buffer
        if (!trend_cm) return fail(XSDBA_E_ARG, "internal: the long-series trend kernel needs the trend buffer");
        const bool mt64 = !pl->d_mem_t32 && pl->d_mem_t64;
        const size_t ldsx = lds1 - (size_t)T * sizeof(R) + (size_t)pl->G * sizeof(R) + 16;
        auto launch = [&](auto kernel) -> int {
            int rc2;
            if ((rc2 = set_lds(kernel, ldsx, ctx, "the detrending coefficients"))) return rc2;
            LaunchScope ls(ctx, "k_group_trend");
            hipLaunchKernelGGL(kernel, dim3((unsigned)C), dim3(1024), ldsx, ctx->stream,
                               (const R *)s.dev, T, s.st, s.sc, scaling, pl->G, (int)kind, pl->d_gidx, pl->W, pl->d_mem_ptr,
                               pl->d_mem_idx, d_tnum, degree, trend_cm, lane_per_group, preserve_mean, min_periods, tn_affine, tn0, tnd,
                               mt64 ? pl->d_mem_t64 : pl->d_mem_t32, coef_out);
            return 0;
        };
        if ((rc = mt64 ? launch(xs::k_group_trend<R, false, true, true>) : launch(xs::k_group_trend<R, false, false, true>))) return rc;
    } else if (m_lds && lds2 <= (size_t)ctx->lds_limit) {
        if ((rc = set_lds(xs::k_group_trend<R, true>, lds2, ctx, "the detrending series"))) return rc;
        LaunchScope ls(ctx, "k_group_trend");
        hipLaunchKernelGGL((xs::k_group_trend<R, true>), dim3((unsigned)C), dim3(nthr), lds2, ctx->stream,
                           (const R *)s.dev, T, s.st, s.sc, scaling, pl->G, (int)kind, pl->d_gidx, pl->W, pl->d_mem_ptr,
                           pl->d_mem_idx, d_tnum, degree, trend_cm, lane_per_group, preserve_mean, min_periods, tn_affine, tn0, tnd, pl->d_mem_t32, coef_out);
    } else {
        const bool mt64 = !pl->d_mem_t32 && pl->d_mem_t64;
        auto launch = [&](auto kernel) -> int {
            int rc2;
            if ((rc2 = set_lds(kernel, lds1, ctx, "the detrending series"))) return rc2;
            LaunchScope ls(ctx, "k_group_trend");
            hipLaunchKernelGGL(kernel, dim3((unsigned)C), dim3(nthr), lds1, ctx->stream,
                               (const R *)s.dev, T, s.st, s.sc, scaling, pl->G, (int)kind, pl->d_gidx, pl->W, pl->d_mem_ptr,
                               pl->d_mem_idx, d_tnum, degree, trend_cm, lane_per_group, preserve_mean, min_periods, tn_affine, tn0, tnd,
                               mt64 ? pl->d_mem_t64 : pl->d_mem_t32, coef_out);
            return 0;
        };
        if ((rc = mt64 ? launch(xs::k_group_trend<R, false, true>) : launch(xs::k_group_trend<R, false, false>))) return rc;
    }
    return check_launch("k_group_trend");
}

template <typename R>
int dqm_adjust_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                    const void *hq, const void *scaling, int G, int Q, const int32_t *gidx, int W, const double *tnum,
                    int degree, int interp, int extrap, char kind, void *scen, void *trend, int64_t ost, int64_t osc)
{
    // `trend` is an input: the fit was done by the caller (optionally with a kind of its own / mult_skip_zeros)
    const bool trend_given = degree <= XSDBA_TREND_GIVEN && degree >= XSDBA_TREND_GIVEN_MUL_SKIP_ZEROS;
    const int tkind = degree == XSDBA_TREND_GIVEN_ADD ? '+' : (degree == XSDBA_TREND_GIVEN_MUL || degree == XSDBA_TREND_GIVEN_MUL_SKIP_ZEROS) ? '*' : (int)kind;
    const int skip0 = degree == XSDBA_TREND_GIVEN_MUL_SKIP_ZEROS;
    if (C <= 0 || Q <= 0 || !af || !hq || !scaling || !scen || (!tnum && !trend_given)) return fail(XSDBA_E_ARG, "null/empty argument");
    if (!trend_given && (degree < 0 || degree > xs::POLY_MAX_DEG)) return fail(XSDBA_E_DEGREE, "PolyDetrend degree %d not implemented (0..%d)", degree, xs::POLY_MAX_DEG);
    if (trend_given && (!trend || ost != 1 || osc != T)) return fail(XSDBA_E_ARG, "a given trend must be a contiguous [cell][time] array");
    if (G > 1 && interp != XSDBA_INTERP_NEAREST) return fail(XSDBA_E_UNSUPPORTED, "grouped interpolation other than 'nearest' is not implemented");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    size_t tab = (size_t)C * G * Q * sizeof(R);
    const void *daf, *dhq, *dsc, *dtn;
    if ((rc = stage_in(ctx, S_TAB0, af, tab, &daf)) || (rc = stage_in(ctx, S_TAB1, hq, tab, &dhq)) ||
        (rc = stage_in(ctx, S_TAB2, scaling, (size_t)C * G * sizeof(R), &dsc))) return rc;
    if (!trend_given && (rc = stage_in(ctx, S_Q, tnum, (size_t)T * sizeof(double), &dtn))) return rc;
    SeriesOut o, otr;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    const R *trend_cm;
    xs::TrendCoef tcoef;
    if (trend_given) {
        const void *dtr;
        if ((rc = stage_in(ctx, S_OUT1, trend, (size_t)T * C * sizeof(R), &dtr))) return rc;
        trend_cm = (const R *)dtr;
    } else {
        // the trend is always needed on device in cell-major form
        // nobody asked for the trend array and it is a line per group: the trend kernel leaves its coefficients and the mapping
        // kernel evaluates them per element — no [cell][time] array written and read back
        const size_t tab_c = ((xs::tab_lds_bytes<R>(G, Q) + 15) & ~(size_t)15) + (size_t)3 * G * sizeof(double) + (((size_t)G * sizeof(float) + 15) & ~(size_t)15);
        // (series long enough for the trend kernel to park its window means in the trend buffer keep the array)
        const unsigned tthr = (!getenv("XSDBA_HIP_TREND_THREADS") && T > xs::TREND_CH * 512 && T <= xs::TREND_CH * 1024) ? 1024u : trend_threads();  // as launch_trend
        const bool means_in_regs = (pl->W == 1 || T <= (int64_t)xs::TREND_CH * tthr) &&
                                   (size_t)T * sizeof(R) + 16384 < (size_t)ctx->lds_limit;  // (a series beyond the LDS: the trend kernel needs the buffer)
        if (!trend && degree <= 1 && means_in_regs && tab_c <= (size_t)ctx->lds_limit && !getenv("XSDBA_HIP_TREND_ARRAY")) {
            void *cbuf;
            if ((rc = ensure(ctx, S_CM2, (size_t)C * G * 3 * sizeof(double), &cbuf))) return rc;
            if ((rc = launch_trend<R>(ctx, s, T, C, (const R *)dsc, pl, kind, (const double *)dtn, degree, (R *)nullptr, 0, 0,
                                      is_device_ptr(tnum) ? (const double *)nullptr : tnum, (double *)cbuf, &tcoef))) return rc;
            trend_cm = nullptr;
        } else {
            if (trend) { if ((rc = series_out_begin<R>(ctx, S_OUT1, S_CM2, trend, T, C, ost, osc, &otr))) return rc; }
            else { if ((rc = ensure(ctx, S_CM2, (size_t)T * C * sizeof(R), &otr.kernel_dst))) return rc; }
            if ((rc = launch_trend<R>(ctx, s, T, C, (const R *)dsc, pl, kind, (const double *)dtn, degree, (R *)otr.kernel_dst, 0, 0,
                                      is_device_ptr(tnum) ? (const double *)nullptr : tnum))) return rc;
            trend_cm = (const R *)otr.kernel_dst;
        }
    }
    if ((rc = launch_qm_adjust<R>(ctx, s, T, C, (const R *)daf, (const R *)dhq, G, Q, pl->d_gidx, interp, extrap, kind,
                                  (R *)o.kernel_dst, nullptr, (const R *)dsc, trend_cm, tkind, skip0, tcoef))) return rc;
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    if (!trend_given && trend && (rc = series_out_end<R>(ctx, &otr, T, C))) return rc;
    return sync_if(ctx, o.copy_back || otr.copy_back || !is_device_ptr(sim) || (tnum && !is_device_ptr(tnum)) || (trend_given && !is_device_ptr(trend)));
}

template <typename R>
int trend_impl(xsdba_hip_ctx *ctx, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t *gidx, int G,
               int W, const double *tnum, int degree, int preserve_mean, char kind, void *trend, int64_t ost, int64_t osc)
{
    if (C <= 0 || !trend || !tnum) return fail(XSDBA_E_ARG, "null/empty argument");
    if (degree < 0 || degree > xs::POLY_MAX_DEG) return fail(XSDBA_E_DEGREE, "PolyDetrend degree %d not implemented (0..%d)", degree, xs::POLY_MAX_DEG);
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x, T, C, st, sc, &s))) return rc;
    const void *dtn;
    if ((rc = stage_in(ctx, S_Q, tnum, (size_t)T * sizeof(double), &dtn))) return rc;
    SeriesOut otr;
    if ((rc = series_out_begin<R>(ctx, S_OUT1, S_CM2, trend, T, C, ost, osc, &otr))) return rc;
    if ((rc = launch_trend<R>(ctx, s, T, C, (const R *)nullptr, pl, kind, (const double *)dtn, degree, (R *)otr.kernel_dst, preserve_mean, 0,
                              is_device_ptr(tnum) ? (const double *)nullptr : tnum))) return rc;
    if ((rc = series_out_end<R>(ctx, &otr, T, C))) return rc;
    return sync_if(ctx, true);
}

template <typename R>
int rolling_mean_impl(xsdba_hip_ctx *ctx, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc, int W, int min_periods, void *out,
                      int64_t ost, int64_t osc)
{
    if (C <= 0 || !out || W < 1 || min_periods < 1 || min_periods > W) return fail(XSDBA_E_ARG, "bad window / min_periods");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, nullptr, T, 1, W, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x, T, C, st, sc, &s))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT1, S_CM2, out, T, C, ost, osc, &o))) return rc;
    if ((rc = launch_trend<R>(ctx, s, T, C, (const R *)nullptr, pl, '+', (const double *)nullptr, 0, (R *)o.kernel_dst, 0, min_periods))) return rc;
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, true);
}

template <typename R>
int qdm_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
             const void *quantiles, int G, int Q, const int32_t *gidx, int W, int rank_window, int interp, int extrap,
             char kind, void *scen, double *sim_q, int64_t ost, int64_t osc)
{
    const bool ranks_only = af == nullptr;
    if (ranks_only) { interp = (rank_window & 2) ? -1 : 0; }  // bit 1 of rank_window: plain average ranks instead of percent ranks
    rank_window &= 1;
    if (C <= 0 || (!ranks_only && (Q <= 0 || !quantiles || !scen)) || (ranks_only && !sim_q)) return fail(XSDBA_E_ARG, "null/empty argument");
    if (!ranks_only && G > 1 && interp != XSDBA_INTERP_NEAREST) return fail(XSDBA_E_UNSUPPORTED, "grouped interpolation other than 'nearest' is not implemented");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    const void *daf = nullptr, *dqn = nullptr;
    if (!ranks_only) {
        if ((rc = stage_in(ctx, S_TAB0, af, (size_t)C * G * Q * sizeof(R), &daf)) ||
            (rc = stage_in(ctx, S_Q, quantiles, (size_t)Q * sizeof(R), &dqn))) return rc;
    }
    SeriesOut o, oq;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    if ((rc = series_out_begin<double>(ctx, S_OUT1, S_CM2, sim_q, T, C, ost, osc, &oq))) return rc;
    const size_t tab = ranks_only ? 0 : ((xs::tab_lds_bytes<R>(G, Q) + 15) & ~(size_t)15);            // full layout (global scratch)
    const size_t tab_sh = ranks_only ? 0 : ((xs::tab_lds_bytes<R>(G, Q, true) + 15) & ~(size_t)15);   // shared abscissa row (LDS)
    const bool rw = rank_window && W > 1;
    const int *gp = rw ? pl->d_row_ptr : pl->d_mem_ptr, *gi = rw ? pl->d_row_idx : pl->d_mem_idx;
    // short ranking rows (exact day-of-year groups) are sorted once per cell instead of counted per element
    const int max_rank_row = rw ? pl->max_row : pl->max_mem;
    int sort_w = getenv("XSDBA_HIP_QDM_NOSORT") ? 0 : (max_rank_row <= 32 ? 32 : (max_rank_row <= 64 ? 64 : 0));
    auto base_bytes = [&](int sw, bool xg_) {
        return ((sizeof(double) * 3 * (size_t)G + 15) & ~(size_t)15) + (xg_ ? 0 : (((size_t)T * sizeof(R) + 15) & ~(size_t)15)) +
               (sw ? (size_t)G * (sw + 1) * sizeof(typename xs::KeyOf<R>::U) : 0);
    };
    // a series beyond the LDS (hourly data): the generic kernel reads it from global memory in place (k_qdm_adjust<XG>)
    const bool xg = base_bytes(0, false) > (size_t)ctx->lds_limit;
    if (sort_w && base_bytes(sort_w, xg) > (size_t)ctx->lds_limit) sort_w = 0;  // the sorted rows do not fit next to the series
    const size_t base = base_bytes(sort_w, xg);
    // generic kernel; `cells`: only the listed cells, count read on the device.  Every block owns a slice of global
    // scratch for a full-layout table (cells with NaN factors; all cells when the LDS cannot hold the table).
    auto launch_generic = [&](const int *cells, const int *n_cells_dev, int64_t n_blocks) -> int {
        int rc2;
        const unsigned nblk = (unsigned)std::min<int64_t>(n_blocks, 1024);
        void *scratch = nullptr;
        if (tab && (rc2 = ensure(ctx, S_SCR, tab * nblk, &scratch))) return rc2;
        auto go = [&](auto kernel, size_t lds, const char *what) -> int {
            int rc3;
            if ((rc3 = set_lds(kernel, lds, ctx, what))) return rc3;
            LaunchScope ls(ctx, "k_qdm_adjust");
            hipLaunchKernelGGL(kernel, dim3(nblk), dim3(qdm_threads()), lds, ctx->stream,
                               (const R *)s.dev, T, C, s.st, s.sc, (const R *)daf, (const R *)dqn, G, Q, pl->d_gidx, gp, gi, interp,
                               extrap, (int)kind, (R *)o.kernel_dst, (double *)oq.kernel_dst, (int64_t)1, T,
                               (unsigned char *)scratch, tab, cells, n_cells_dev, sort_w);
            return 0;
        };
        if (base + tab_sh <= (size_t)ctx->lds_limit)
            rc2 = xg ? go(xs::k_qdm_adjust<R, true, true>, base + tab_sh, "the factor table") : go(xs::k_qdm_adjust<R, true, false>, base + tab_sh, "the ranking series + factor table");
        else
            rc2 = xg ? go(xs::k_qdm_adjust<R, false, true>, base, "the group statistics") : go(xs::k_qdm_adjust<R, false, false>, base, "the ranking series");
        if (rc2) return rc2;
        return check_launch("k_qdm_adjust");
    };
    bool fast = false;
    {
        // exact groups of at most 32 members (one per year of a day-of-year grouping): a lane per group, the members sorted in
        // registers (qdm_lanes.hpp).  XSDBA_HIP_QDM=count pins the generic kernel, =cell the sort kernel.
        const char *force = getenv("XSDBA_HIP_QDM");
        const int nchunk = (G + 63) / 64;
        const int nw = sizeof(R) == 4 ? 4 : 2;  // waves per block (float64 rows take twice the LDS); any wave takes any (cell, chunk) item
        const size_t lds = xs::qdm_lanes_lds_bytes(nw, Q, pl->max_mem, sizeof(R));
        if (!ranks_only && !sim_q && !rw && G > 1 && interp == XSDBA_INTERP_NEAREST && pl->d_mem_t32 && s.st == 1 && Q < xs::QL_NAN &&
            T < 65535 && (int64_t)C * nchunk < (1ll << 31) - 1 && lds <= (size_t)ctx->lds_limit && !force) {
            void *flags;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C * nchunk + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            // time steps in no group stay NaN (no lane owns them)
            if (pl->has_ungrouped) {
                hipLaunchKernelGGL(xs::k_fill_nan<R>, dim3((unsigned)(8 * ctx->n_cu)), dim3(256), 0, ctx->stream, (R *)o.kernel_dst, (int64_t)T * C);
                if ((rc = check_launch("k_fill_nan"))) return rc;
            }
            if ((rc = set_lds(xs::k_qdm_lanes<R>, lds, ctx, "the factor rows + the member-order buffer"))) return rc;
            // up to 256 VGPRs: two waves per SIMD, eight per CU
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(8 / nw), (size_t)ctx->lds_limit / lds));
            const unsigned nblk = (unsigned)std::min<int64_t>(C, (int64_t)per_cu * ctx->n_cu);
            {
                LaunchScope ls(ctx, "k_qdm_lanes");
                hipLaunchKernelGGL(xs::k_qdm_lanes<R>, dim3(nblk), dim3(64 * nw), lds, ctx->stream, (const R *)s.dev, T, C, s.sc,
                                   (const R *)daf, (const R *)dqn, G, Q, pl->d_mem_t32, pl->max_mem, extrap, (int)kind,
                                   (R *)o.kernel_dst, T, flag_count, flag_list);
                if ((rc = check_launch("k_qdm_lanes"))) return rc;
            }
            // cells with NaN factors (the own-row shortcut does not hold): generic kernel, device-side count
            if ((rc = launch_generic(flag_list, flag_count, std::min<int64_t>(C, 256)))) return rc;
            if ((rc = series_out_end<R>(ctx, &o, T, C)) || (rc = series_out_end<double>(ctx, &oq, T, C))) return rc;
            return sync_if(ctx, o.copy_back || oq.copy_back || !is_device_ptr(sim));
        }
    }
    {
        // sort-based kernel (qdm_cell.hpp): any grouping at O(T log T) per cell instead of O(T * row length)
        const char *force = getenv("XSDBA_HIP_QDM");  // "count" pins the generic kernel (A/B runs, tests)
        fast = interp != XSDBA_INTERP_CUBIC && pl->simple && T <= 11 * 1024 && T >= 64 && G <= xs::QdmCfg<11>::GMAX && (ranks_only || Q <= xs::QdmCfg<11>::QMAX) &&
               s.st == 1 && !(force && !strcmp(force, "count")) && ctx->lane_ordered_atomics;
        // exact (non-windowed) ranking inside short windowed groups, e.g. 30 members per day of year: counting over
        // 30 values beats the sort (measured 6.8 vs 8.6 ms per 16384 cells); "cell" forces the sort kernel anyway
        if (fast && !(rw || W == 1) && pl->max_mem <= 48 && !(force && !strcmp(force, "cell"))) fast = false;
        if (fast) {
            void *flags;
            if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * ((size_t)C + 1), &flags))) return rc;
            int *flag_count = (int *)flags, *flag_list = flag_count + 1;
            HIP_TRY(hipMemsetAsync(flag_count, 0, sizeof(int), ctx->stream));
            const int ept = T <= 4 * 1024 ? 4 : (T <= 8 * 1024 ? 8 : 11);
            // bit 1: un-windowed groups, at most 64 of them: ranks by a group-keyed stable pass of the sorted order instead of the
            // bit-vector walk (XSDBA_HIP_QDM_WALK=1 keeps the walk: A/B runs, tests)
            const bool walk = getenv("XSDBA_HIP_QDM_WALK") != nullptr;
            const int use_tog = ((rw || W == 1) ? 1 : 0) | ((W == 1 && G > 1 && G <= 64 && !pl->has_ungrouped && !walk) ? 2 : 0);
            // the kernel without the walks: those groups, or a single group holding every time step (the sorted order is the ranking)
            const bool pass = (use_tog & 2) || (G == 1 && !pl->has_ungrouped && !walk);
            // the node tables of pass D for rows without NaNs and without ties at their extremes, once per launch (XSDBA_HIP_QDM_TABLES=0:
            // every cell builds its own as before — A/B runs, tests)
            const unsigned char *gtab = (const unsigned char *)flag_count;  // (a readable address when there are no tables)
            xs::QdmTabs gt;
            const char *tabenv = getenv("XSDBA_HIP_QDM_TABLES");
            // (a single group holding every step: ONE table, of the 1-D nearest flavour — interp="nearest" only)
            const bool one_d = G == 1 && pass && interp == XSDBA_INTERP_NEAREST && pl->row_ptr[1] - pl->row_ptr[0] == (int)T;
            if (!ranks_only && !sim_q && (G > 1 || one_d) && Q <= 63 && !(tabenv && !strcmp(tabenv, "0"))) {
                if ((rc = make_group_tabs(ctx, pl, (use_tog & 1) != 0))) return rc;
                const Plan::GroupTabs &t = pl->gt[(use_tog & 1) ? 1 : 0];
                if (t.K) {
                    void *tb;
                    if ((rc = ensure(ctx, S_TAB1, t.bytes, &tb))) return rc;
                    LaunchScope ls(ctx, "k_qdm_node_tables");
                    hipLaunchKernelGGL(xs::k_qdm_node_tables<R>, dim3((unsigned)std::min(8, (2 * t.max_n + 2 + 255) / 256), (unsigned)t.K), dim3(256), 0,
                                       ctx->stream, (const R *)dqn, Q, t.d_cls_N, t.d_cls_off, (unsigned char *)tb, one_d ? 1 : 0);
                    if ((rc = check_launch("k_qdm_node_tables"))) return rc;
                    gtab = (const unsigned char *)tb;
                    gt.K = t.K; gt.bytes = (int)t.bytes;
                    for (int k = 0; k < t.K; k++) { gt.N[k] = t.cls_N[k]; gt.off[k] = t.cls_off[k] / 16; }
                    if (tabenv && !strcmp(tabenv, "global")) gt.bytes = 1 << 20;  // (A/B runs: the tables stay in global memory)
                }
            }
            auto go = [&](auto cfg, auto pass_c) -> int {
                constexpr int EPT = decltype(cfg)::value;
                constexpr bool PASS = decltype(pass_c)::value;
                const size_t lds = xs::QdmCfg<EPT>::lds_bytes;
                int rc2;
                if ((rc2 = set_lds((xs::k_qdm_cell<EPT, R, PASS>), lds, ctx, "the per-cell sort"))) return rc2;
                const unsigned nblk = (unsigned)std::min<int64_t>(C, 4 * ctx->n_cu);
                LaunchScope ls(ctx, "k_qdm_cell");
                hipLaunchKernelGGL((xs::k_qdm_cell<EPT, R, PASS>), dim3(nblk), dim3(1024), lds, ctx->stream, (const R *)s.dev, T, C, s.sc,
                                   (const R *)daf, (const R *)dqn, G, Q, pl->d_gidx, (use_tog & 1) ? pl->d_row_ptr : pl->d_mem_ptr,
                                   (use_tog & 1) ? pl->d_row_sorted : pl->d_mem_idx, pl->d_mem_ptr, pl->d_mem_idx, pl->d_tog_ptr,
                                   pl->d_tog_idx, pl->d_tog64, use_tog, interp, extrap, (int)kind, (R *)o.kernel_dst,
                                   (double *)oq.kernel_dst, flag_count, flag_list, pl->max_mem, pl->d_mem_t32, gtab, gt, pl->d_gidx8);
                return check_launch("k_qdm_cell");
            };
            using std::integral_constant;
            using std::true_type;
            using std::false_type;
            if (pass) rc = ept == 4 ? go(integral_constant<int, 4>{}, true_type{}) : ept == 8 ? go(integral_constant<int, 8>{}, true_type{}) : go(integral_constant<int, 11>{}, true_type{});
            else rc = ept == 4 ? go(integral_constant<int, 4>{}, false_type{}) : ept == 8 ? go(integral_constant<int, 8>{}, false_type{}) : go(integral_constant<int, 11>{}, false_type{});
            if (rc) return rc;
            // cells the own-row shortcut could not serve (NaN factors) are redone by the generic kernel, device-side count
            if (!ranks_only && (rc = launch_generic(flag_list, flag_count, std::min<int64_t>(C, 256)))) return rc;
        }
    }
    if (!fast && (rc = launch_generic(nullptr, nullptr, C))) return rc;
    if ((rc = series_out_end<R>(ctx, &o, T, C)) || (rc = series_out_end<double>(ctx, &oq, T, C))) return rc;
    return sync_if(ctx, o.copy_back || oq.copy_back || !is_device_ptr(sim));
}

template <typename R>
int tail_mask_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *hq_raw,
                   int G, int Q, const int32_t *gidx, double factor, void *scen, int64_t ost, int64_t osc)
{
    if (C <= 0 || Q <= 0 || !hq_raw || !scen) return fail(XSDBA_E_ARG, "null/empty argument");
    if (T > (1ll << 30) || C > 65535ll * 65535ll) return fail(XSDBA_E_ARG, "series too long / too many cells");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, 1, &pl))) return rc;
    SeriesIn s, so;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    // scen is read-modify-write: its current values come in like an input series, the result leaves like an output
    if ((rc = series_in<R>(ctx, S_IN1, S_CM1, scen, T, C, ost, osc, &so))) return rc;
    const void *dhq;
    if ((rc = stage_in(ctx, S_TAB0, hq_raw, (size_t)C * G * Q * sizeof(R), &dhq))) return rc;
    const bool inplace = is_device_ptr(scen) && ost == 1;  // so.dev is the caller's buffer itself
    R *dst = (R *)const_cast<void *>(so.dev);
    {
        LaunchScope ls(ctx, "k_tail_mask");
        const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 64);
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t nc = std::min<int64_t>(65535, C - c0);
            hipLaunchKernelGGL(xs::k_tail_mask<R>, dim3(gx, (unsigned)nc), dim3(256), 0, ctx->stream,
                               (const R *)s.dev + c0 * s.sc, T, nc, s.sc, (const R *)dhq + (size_t)c0 * G * Q, G, Q, pl->d_gidx,
                               (R)factor, dst + c0 * so.sc, so.sc);
        }
        if ((rc = check_launch("k_tail_mask"))) return rc;
    }
    if (!inplace) {
        // the masked copy lives in library scratch (cell-major, row stride so.sc): send it back in the caller's layout
        if (is_device_ptr(scen)) {
            if ((rc = launch_transpose<R>(ctx, (const R *)dst, C, T, so.sc, 1, (R *)scen, osc, ost))) return rc;
        } else {
            SeriesOut o;
            if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM2, scen, T, C, ost, osc, &o))) return rc;
            if (so.sc == T) HIP_TRY(hipMemcpyAsync(o.kernel_dst, dst, (size_t)T * C * sizeof(R), hipMemcpyDeviceToDevice, ctx->stream));
            else if ((rc = launch_transpose<R>(ctx, (const R *)dst, C, T, so.sc, 1, (R *)o.kernel_dst, T, 1))) return rc;
            if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
        }
    }
    return sync_if(ctx, true);
}

// ---- pre-processing ---------------------------------------------------------------------
template <typename R> R next_up(R v) { return std::nextafter(v, std::numeric_limits<R>::infinity()); }
template <typename R> R next_down(R v) { return std::nextafter(v, -std::numeric_limits<R>::infinity()); }

template <typename R>
int jitter_impl(xsdba_hip_ctx *ctx, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc, double lower, double minimum,
                double upper, double maximum, uint64_t seed, int64_t cell0, void *out, int64_t ost, int64_t osc)
{
    if (C <= 0 || !out) return fail(XSDBA_E_ARG, "null/empty argument");
    if (upper == upper && !(maximum == maximum)) return fail(XSDBA_E_ARG, "If 'upper' is given, so must 'maximum'.");
    if (T > (1ll << 30)) return fail(XSDBA_E_ARG, "series too long");
    int rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x, T, C, st, sc, &s))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, out, T, C, ost, osc, &o))) return rc;
    // processing.py:223-225, 245-251: the open ends are stepped by one ulp in the data dtype
    const double jmin = (double)next_up<R>((R)(minimum == minimum ? minimum : 0.0));
    const double jmax = sizeof(R) < 8 ? (double)next_down<R>((R)maximum) : maximum;
    {
        LaunchScope ls(ctx, "k_jitter");
        const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 64);
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t nc = std::min<int64_t>(65535, C - c0);
            hipLaunchKernelGGL(xs::k_jitter<R>, dim3(gx, (unsigned)nc), dim3(256), 0, ctx->stream, (const R *)s.dev + c0 * s.sc, T,
                               s.sc, lower, jmin, upper, jmax, seed, cell0 + c0, (R *)o.kernel_dst + c0 * T, T);
        }
        if ((rc = check_launch("k_jitter"))) return rc;
    }
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, o.copy_back || !is_device_ptr(x));
}

template <typename R>
int adapt_freq_impl(xsdba_hip_ctx *ctx, const void *sim, const void *ref, int64_t T, int64_t C, int64_t st, int64_t sc,
                    const int32_t *gidx, int G, int W, double thresh, uint64_t seed, int64_t cell0, double *P0_ref,
                    double *P0_hist, void *pth, double *dP0, void *sim_ad, int64_t ost, int64_t osc)
{
    using U = typename xs::KeyOf<R>::U;
    if (C <= 0 || !P0_ref || !P0_hist || !pth || !sim_ad) return fail(XSDBA_E_ARG, "null/empty argument");
    if (!ref && W != 1) return fail(XSDBA_E_ARG, "adapt_freq on the adjust side runs on the groups without window (window must be 1)");
    if ((int64_t)C * G > (1ll << 31) - 1) return fail(XSDBA_E_ARG, "n_cell * n_groups exceeds 2^31");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    if (!pl->simple) return fail(XSDBA_E_UNSUPPORTED, "adapt_freq: a time step occurs twice in one group block (coarse group + window)");
    SeriesIn s, r;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    if (ref && (rc = series_in<R>(ctx, S_IN1, S_CM1, ref, T, C, st, sc, &r))) return rc;
    const size_t ng = (size_t)C * G;
    OutStage o0, o1, o2, o3;  // P0_ref, P0_hist, pth, dP0: outputs when training, inputs when adjusting
    const void *i0 = nullptr, *i1 = nullptr, *i2 = nullptr;
    if (ref) {
        if ((rc = stage_out(ctx, S_TAB0, P0_ref, ng * sizeof(double), &o0)) || (rc = stage_out(ctx, S_TAB1, P0_hist, ng * sizeof(double), &o1)) ||
            (rc = stage_out(ctx, S_TAB2, pth, ng * sizeof(R), &o2))) return rc;
    } else {
        if ((rc = stage_in(ctx, S_TAB0, P0_ref, ng * sizeof(double), &i0)) || (rc = stage_in(ctx, S_TAB1, P0_hist, ng * sizeof(double), &i1)) ||
            (rc = stage_in(ctx, S_TAB2, pth, ng * sizeof(R), &i2))) return rc;
    }
    if ((rc = stage_out(ctx, S_MISC, dP0, ng * sizeof(double), &o3))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM2, sim_ad, T, C, ost, osc, &o))) return rc;
    // sim_ad starts as a copy of sim (cell-major): untouched values and time steps in no group keep it
    if (s.sc == T) HIP_TRY(hipMemcpyAsync(o.kernel_dst, s.dev, (size_t)T * C * sizeof(R), hipMemcpyDeviceToDevice, ctx->stream));
    else if ((rc = launch_transpose<R>(ctx, (const R *)s.dev, C, T, s.sc, 1, (R *)o.kernel_dst, T, 1))) return rc;
    int P = 2;
    while (P < pl->max_row) P <<= 1;
    int NT = P >= 8192 ? 1024 : (P >= 2048 ? 512 : (P >= 512 ? 256 : (P >= 128 ? 128 : 64)));
    if (const char *e = getenv("XSDBA_HIP_AF_THREADS")) { const int v = atoi(e); if (v >= 64 && v <= 1024) NT = (v / 64) * 64; }  // (tuning runs)
    const size_t red_b = sizeof(double) * (2 * (size_t)(NT / 64) + 2);
    // By selection (preproc.hpp adapt_freq_block_sel; no row limit, and no slower than the LDS sorts where those apply), the rows
    // staged in LDS when they fit.  XSDBA_HIP_AF_SELECT=0 (A/B runs, tests): the bitonic sorts in LDS wherever they apply — a
    // ref row within the LDS, at most 65 535 values in a block (the slot of the LDS tags has 16 bits).
    const char *selenv = getenv("XSDBA_HIP_AF_SELECT");
    const bool sorts = selenv && !strcmp(selenv, "0") && red_b + (size_t)P * sizeof(U) <= (size_t)ctx->lds_limit && pl->max_row <= 65535;
    const size_t sel_b = (size_t)(32 + 2048) * sizeof(int);  // counters, results, four histograms (the last two double-buffered)
    int Lcap = -1, Ps = P;
    size_t lds;
    if (sorts) {
        // the (key, tag) pairs of the sub-threshold values: as many as the LDS takes, at most the row
        while (Ps > 2 && red_b + (size_t)Ps * (sizeof(U) + sizeof(uint32_t)) > (size_t)ctx->lds_limit) Ps >>= 1;
        lds = std::max(red_b + std::max((size_t)P * sizeof(U), (size_t)Ps * (sizeof(U) + sizeof(uint32_t))), sel_b);
    } else {
        const size_t per = (ref ? 2 : 1) * sizeof(R) + sizeof(int);  // a staged element: the sim value, the ref value (training), the time step
        Lcap = sel_b + (size_t)pl->max_row * per <= (size_t)ctx->lds_limit ? pl->max_row : 0;
        lds = sel_b + (size_t)Lcap * per;
    }
    if ((rc = sorts ? set_lds(xs::k_adapt_freq<R, true>, lds, ctx, "the adapt_freq rows") : set_lds(xs::k_adapt_freq<R, false>, lds, ctx, "the adapt_freq rows"))) return rc;
    // [0] unused, [1] length of the work list, [2...] the (cell, group) blocks with dP0 > 0
    void *flag;
    if ((rc = ensure(ctx, S_FLAGS, sizeof(int) * (ng + 2), &flag))) return rc;
    HIP_TRY(hipMemsetAsync(flag, 0, 2 * sizeof(int), ctx->stream));
    int *n_work = (int *)flag + 1, *work_list = (int *)flag + 2;
    // cheap screening pass per cell (prefix sums over time) when the four prefix arrays fit the LDS
    const int snt = 512;
    const size_t slds = (((size_t)8 * (size_t)(T + 1) + 15) & ~(size_t)15) + sizeof(int) * 2 * snt;
    // (a thread per group there: with few groups — one group: ONE thread walks the whole series, 77 ms per 8 192 cells against 2.8 for
    //  everything k_adapt_freq does — the blocks count for themselves, in parallel)
    const bool screened = T < 65535 && slds <= (size_t)ctx->lds_limit && G >= 64;
    if (screened) {
        if ((rc = set_lds(xs::k_af_screen<R>, slds, ctx, "the adapt_freq prefix sums"))) return rc;
        LaunchScope ls(ctx, "k_af_screen");
        hipLaunchKernelGGL(xs::k_af_screen<R>, dim3((unsigned)std::min<int64_t>(C, 8 * ctx->n_cu)), dim3(snt), slds, ctx->stream,
                           (const R *)s.dev, ref ? (const R *)r.dev : (const R *)nullptr, T, C, s.sc, ref ? r.sc : (int64_t)0,
                           pl->d_mem_ptr, pl->d_mem_idx, G, W, thresh, ref ? 1 : 0, ref ? (double *)o0.dev : (double *)const_cast<void *>(i0),
                           ref ? (double *)o1.dev : (double *)const_cast<void *>(i1), ref ? (R *)o2.dev : (R *)const_cast<void *>(i2),
                           (double *)o3.dev, work_list, n_work);
        if ((rc = check_launch("k_af_screen"))) return rc;
    }
    {
        LaunchScope ls(ctx, "k_adapt_freq");
        const unsigned grid = screened ? (unsigned)std::min<size_t>(ng, (size_t)32 * ctx->n_cu) : (unsigned)ng;
        auto *kern = sorts ? xs::k_adapt_freq<R, true> : xs::k_adapt_freq<R, false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, ctx->stream, (const R *)s.dev,
                           ref ? (const R *)r.dev : (const R *)nullptr, s.sc, ref ? r.sc : (int64_t)0, pl->d_row_ptr, pl->d_row_idx,
                           pl->d_gidx, G, thresh, seed, cell0, ref ? 1 : 0, ref ? (double *)o0.dev : (double *)const_cast<void *>(i0),
                           ref ? (double *)o1.dev : (double *)const_cast<void *>(i1), ref ? (R *)o2.dev : (R *)const_cast<void *>(i2),
                           (double *)o3.dev, (R *)o.kernel_dst, T, P, Ps, Lcap, screened ? work_list : (const int *)nullptr,
                           screened ? n_work : (const int *)nullptr, (int64_t)ng);
        if ((rc = check_launch("k_adapt_freq"))) return rc;
    }
    if (ref && ((rc = finish_out(ctx, &o0)) || (rc = finish_out(ctx, &o1)) || (rc = finish_out(ctx, &o2)))) return rc;
    if ((rc = finish_out(ctx, &o3))) return rc;
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, true);
}

template <typename R>
int group_means_impl(xsdba_hip_ctx *ctx, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t *gidx, int G,
                     int W, void *out)
{
    if (C <= 0 || !out) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((int64_t)C * G > (1ll << 31) - 1) return fail(XSDBA_E_ARG, "n_cell * n_groups exceeds 2^31");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x, T, C, st, sc, &s))) return rc;
    OutStage o;
    if ((rc = stage_out(ctx, S_TAB0, out, (size_t)C * G * sizeof(R), &o))) return rc;
    const int NT = pl->max_row >= 2048 ? 512 : (pl->max_row >= 256 ? 256 : 64);
    {
        LaunchScope ls(ctx, "k_group_mean");
        hipLaunchKernelGGL(xs::k_group_mean<R>, dim3((unsigned)((size_t)C * G)), dim3(NT), sizeof(double) * (2 * (NT / 64) + 2),
                           ctx->stream, (const R *)s.dev, s.sc, pl->d_row_ptr, pl->d_row_idx, G, (R *)o.dev);
        if ((rc = check_launch("k_group_mean"))) return rc;
    }
    if ((rc = finish_out(ctx, &o))) return rc;
    return sync_if(ctx, o.copy_back || !is_device_ptr(x));
}

template <typename R>
int scaling_adjust_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                        const int32_t *gidx, int G, const int32_t *ghi, const double *frac, char kind, void *scen, int64_t ost,
                        int64_t osc)
{
    if (C <= 0 || !af || !scen) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((ghi == nullptr) != (frac == nullptr)) return fail(XSDBA_E_ARG, "group_hi and frac go together");
    if (T > (1ll << 30)) return fail(XSDBA_E_ARG, "series too long");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, 1, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    const void *daf, *dhi = nullptr, *dfr = nullptr;
    if ((rc = stage_in(ctx, S_TAB0, af, (size_t)C * G * sizeof(R), &daf))) return rc;
    if (frac && ((rc = stage_in(ctx, S_TAB1, ghi, (size_t)T * sizeof(int32_t), &dhi)) || (rc = stage_in(ctx, S_TAB2, frac, (size_t)T * sizeof(double), &dfr)))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    {
        LaunchScope ls(ctx, "k_scale_groups");
        const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 64);
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t nc = std::min<int64_t>(65535, C - c0);
            hipLaunchKernelGGL(xs::k_scale_groups<R>, dim3(gx, (unsigned)nc), dim3(256), 0, ctx->stream, (const R *)s.dev + c0 * s.sc,
                               T, s.sc, (const R *)daf + (size_t)c0 * G, G, pl->d_gidx, (const int *)dhi, (const double *)dfr, (int)kind,
                               (R *)o.kernel_dst + c0 * T, T);
        }
        if ((rc = check_launch("k_scale_groups"))) return rc;
    }
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, o.copy_back || !is_device_ptr(sim));
}

template <typename R>
int series_correction_impl(xsdba_hip_ctx *ctx, const void *x, int64_t T, int64_t C, int64_t st, int64_t sc, const void *f,
                           int64_t fst, int64_t fsc, char kind, int inv, int skip0, void *out, int64_t ost, int64_t osc)
{
    if (C <= 0 || T <= 0 || !x || !f || !out) return fail(XSDBA_E_ARG, "null/empty argument");
    int rc;
    SeriesIn s, sf;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, x, T, C, st, sc, &s))) return rc;
    const bool bcast = fsc == 0;  // one series for all cells
    if ((rc = series_in<R>(ctx, S_IN1, S_CM2, f, T, bcast ? 1 : C, fst, bcast ? T : fsc, &sf))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, out, T, C, ost, osc, &o))) return rc;
    {
        LaunchScope ls(ctx, "k_series_correction");
        const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 64);
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t nc = std::min<int64_t>(65535, C - c0);
            hipLaunchKernelGGL(xs::k_series_correction<R>, dim3(gx, (unsigned)nc), dim3(256), 0, ctx->stream,
                               (const R *)s.dev + c0 * s.sc, (const R *)sf.dev + (bcast ? 0 : c0 * sf.sc), T, s.sc, bcast ? (int64_t)0 : sf.sc,
                               (int)kind, inv, skip0, (R *)o.kernel_dst + c0 * T, T);
        }
        if ((rc = check_launch("k_series_correction"))) return rc;
    }
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, true);
}

template <typename R>
int loci_train_impl(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int64_t T, int64_t C, int64_t st, int64_t sc,
                    const int32_t *gidx, int G, int W, double thresh, void *af, void *hist_thresh)
{
    using U = typename xs::KeyOf<R>::U;
    if (C <= 0 || !af || !hist_thresh) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((int64_t)C * G > (1ll << 31) - 1) return fail(XSDBA_E_ARG, "n_cell * n_groups exceeds 2^31");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, W, &pl))) return rc;
    SeriesIn r, h;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, ref, T, C, st, sc, &r)) || (rc = series_in<R>(ctx, S_IN1, S_CM1, hist, T, C, st, sc, &h))) return rc;
    OutStage oa, ot;
    const size_t ng = (size_t)C * G;
    if ((rc = stage_out(ctx, S_TAB0, af, ng * sizeof(R), &oa)) || (rc = stage_out(ctx, S_TAB1, hist_thresh, ng * sizeof(R), &ot))) return rc;
    int P = 2;
    while (P < pl->max_row) P <<= 1;
    const int NT = P >= 8192 ? 1024 : (P >= 2048 ? 512 : (P >= 512 ? 256 : (P >= 128 ? 128 : 64)));
    const size_t red_b = sizeof(double) * (2 * (size_t)(NT / 64) + 2), lds = red_b + (size_t)P * sizeof(U);
    if (lds > (size_t)ctx->lds_limit && P >= 2 * xs::BIG_TILE) {
        // a hist row beyond the LDS sort (60 years of daily float64 values under group="time"): sorted in global memory by the
        // pipeline of train_bigrows.hpp, a batch of rows at a time
        const size_t row_bytes = (size_t)P * sizeof(U);
        const int tiles = P / xs::BIG_TILE;
        int64_t batch = std::max<int64_t>(1, (int64_t)(((size_t)1 << 30) / row_bytes));
        batch = std::min<int64_t>({batch, (int64_t)ng, (int64_t)((1ll << 30) / tiles)});
        const size_t off_nv = ((size_t)batch * row_bytes + 255) & ~(size_t)255;
        const size_t off_mu = (off_nv + (size_t)batch * sizeof(int) + 255) & ~(size_t)255;
        void *scr;
        if ((rc = ensure(ctx, S_SCR, off_mu + (size_t)batch * sizeof(R), &scr))) return rc;
        U *keys = (U *)scr;
        int *nvd = (int *)((unsigned char *)scr + off_nv);
        R *mu = (R *)((unsigned char *)scr + off_mu);
        const size_t lds_tile = (size_t)xs::BIG_TILE * sizeof(U);
        if ((rc = set_lds(xs::k_big_local<U>, lds_tile, ctx, "a tile of the row sort"))) return rc;
        LaunchScope ls(ctx, "k_loci_thresh(big rows)");
        for (int64_t row0 = 0; row0 < (int64_t)ng; row0 += batch) {
            const int64_t nb = std::min<int64_t>(batch, (int64_t)ng - row0);
            hipLaunchKernelGGL(xs::k_big_fill<R>, dim3((unsigned)nb), dim3(1024), 0, ctx->stream, (const R *)h.dev, h.st, h.sc, pl->d_row_ptr,
                               pl->d_row_idx, G, row0, P, (int)'+', 0, keys, nvd, mu);
            hipLaunchKernelGGL(xs::k_big_local<U>, dim3((unsigned)(nb * tiles)), dim3(1024), lds_tile, ctx->stream, keys, P, 0);
            const int64_t npairs = nb * (int64_t)(P / 2);
            for (int k = 2 * xs::BIG_TILE; k <= P && k > 0; k <<= 1) {
                for (int j = k >> 1; j >= xs::BIG_TILE; j >>= 1)
                    hipLaunchKernelGGL(xs::k_big_step<U>, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, ctx->stream, keys, P, k, j, npairs);
                hipLaunchKernelGGL(xs::k_big_local<U>, dim3((unsigned)(nb * tiles)), dim3(1024), lds_tile, ctx->stream, keys, P, k);
            }
            hipLaunchKernelGGL(xs::k_loci_thresh_big<R>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const R *)r.dev, r.sc, pl->d_row_ptr,
                               pl->d_row_idx, G, row0, P, thresh, (const U *)keys, (const int *)nvd, (R *)ot.dev);
            if ((rc = check_launch("k_loci_thresh(big rows)"))) return rc;
        }
    } else {
        if ((rc = set_lds(xs::k_loci_thresh<R>, lds, ctx, "the LOCI row sort"))) return rc;
        LaunchScope ls(ctx, "k_loci_thresh");
        hipLaunchKernelGGL(xs::k_loci_thresh<R>, dim3((unsigned)ng), dim3(NT), lds, ctx->stream, (const R *)r.dev, (const R *)h.dev,
                           r.sc, pl->d_row_ptr, pl->d_row_idx, G, thresh, (R *)ot.dev);
        if ((rc = check_launch("k_loci_thresh"))) return rc;
    }
    {
        LaunchScope ls(ctx, "k_loci_factor");
        hipLaunchKernelGGL(xs::k_loci_factor<R>, dim3((unsigned)ng), dim3(NT), red_b, ctx->stream, (const R *)r.dev, (const R *)h.dev,
                           r.sc, pl->d_row_ptr, pl->d_row_idx, pl->d_gidx, G, thresh, (const R *)ot.dev, (R *)oa.dev);
        if ((rc = check_launch("k_loci_factor"))) return rc;
    }
    if ((rc = finish_out(ctx, &oa)) || (rc = finish_out(ctx, &ot))) return rc;
    return sync_if(ctx, oa.copy_back || ot.copy_back || !is_device_ptr(ref));
}

template <typename R>
int loci_adjust_impl(xsdba_hip_ctx *ctx, const void *sim, int64_t T, int64_t C, int64_t st, int64_t sc, const void *af,
                     const void *hist_thresh, const int32_t *gidx, int G, const int32_t *ghi, const double *frac, double thresh,
                     void *scen, int64_t ost, int64_t osc)
{
    if (C <= 0 || !af || !hist_thresh || !scen) return fail(XSDBA_E_ARG, "null/empty argument");
    if ((ghi == nullptr) != (frac == nullptr)) return fail(XSDBA_E_ARG, "group_hi and frac go together");
    if (T > (1ll << 30)) return fail(XSDBA_E_ARG, "series too long");
    int rc;
    Plan *pl;
    if ((rc = get_plan(ctx, gidx, T, G, 1, &pl))) return rc;
    SeriesIn s;
    if ((rc = series_in<R>(ctx, S_IN0, S_CM0, sim, T, C, st, sc, &s))) return rc;
    const void *daf, *dth, *dhi = nullptr, *dfr = nullptr;
    if ((rc = stage_in(ctx, S_TAB0, af, (size_t)C * G * sizeof(R), &daf)) || (rc = stage_in(ctx, S_TAB1, hist_thresh, (size_t)C * G * sizeof(R), &dth))) return rc;
    if (frac && ((rc = stage_in(ctx, S_TAB2, ghi, (size_t)T * sizeof(int32_t), &dhi)) || (rc = stage_in(ctx, S_MISC, frac, (size_t)T * sizeof(double), &dfr)))) return rc;
    SeriesOut o;
    if ((rc = series_out_begin<R>(ctx, S_OUT0, S_CM1, scen, T, C, ost, osc, &o))) return rc;
    {
        LaunchScope ls(ctx, "k_loci_adjust");
        const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 64);
        for (int64_t c0 = 0; c0 < C; c0 += 65535) {
            const int64_t nc = std::min<int64_t>(65535, C - c0);
            hipLaunchKernelGGL(xs::k_loci_adjust<R>, dim3(gx, (unsigned)nc), dim3(256), 0, ctx->stream, (const R *)s.dev + c0 * s.sc, T,
                               s.sc, (const R *)daf + (size_t)c0 * G, (const R *)dth + (size_t)c0 * G, G, pl->d_gidx, (const int *)dhi,
                               (const double *)dfr, thresh, (R *)o.kernel_dst + c0 * T, T);
        }
        if ((rc = check_launch("k_loci_adjust"))) return rc;
    }
    if ((rc = series_out_end<R>(ctx, &o, T, C))) return rc;
    return sync_if(ctx, o.copy_back || !is_device_ptr(sim) || frac != nullptr);
}

int check_ctx(xsdba_hip_ctx *ctx)
{
    if (!ctx) return fail(XSDBA_E_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

#define DISPATCH(dtype, call_f32, call_f64)                                        \
    ((dtype) == XSDBA_F32 ? (call_f32) : (dtype) == XSDBA_F64 ? (call_f64) : fail(XSDBA_E_ARG, "dtype must be XSDBA_F32 or XSDBA_F64"))

}  // namespace

// =================================================== C ABI ===================================================
API int xsdba_hip_abi_version(void) { return XSDBA_HIP_ABI_VERSION; }

API int xsdba_hip_last_error(char *buf, size_t buflen)
{
    if (!buf || !buflen) return XSDBA_E_ARG;
    snprintf(buf, buflen, "%s", g_err.c_str());
    return 0;
}

API int xsdba_hip_device_count(int *count)
{
    if (!count) return fail(XSDBA_E_ARG, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return 0;
}

API int xsdba_hip_create(int device, xsdba_hip_ctx **out)
{
    if (!out) return fail(XSDBA_E_ARG, "null out pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(XSDBA_E_NO_DEVICE, "no HIP device visible"); }
    if (device < 0 || device >= n) return fail(XSDBA_E_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(XSDBA_E_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    auto *ctx = new (std::nothrow) xsdba_hip_ctx();
    if (!ctx) return fail(XSDBA_E_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->lds_limit = (int)prop.maxSharedMemoryPerMultiProcessor > 0 ? (int)std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor, 160 * 1024) : 64 * 1024;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail((int)e, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    (void)hipEventCreate(&ctx->t0);
    (void)hipEventCreate(&ctx->t1);
    {   // the pool of freed buffers: XSDBA_HIP_POOL_MB, else a quarter of THIS device's memory (two passes of a 129 600-cell
        // shard keep 26 GB of tables and series alive at once; a fixed 16 GiB sent every pass back to hipMalloc / hipFree)
        const char *e = getenv("XSDBA_HIP_POOL_MB");
        if (e) {
            double mb = atof(e);
            if (!(mb >= 0.0)) mb = 0.0;  // negative or unparsable: no pool
            if (mb > 1.0e9) mb = 1.0e9;
            ctx->pool_cap = (size_t)mb << 20;
        } else {
            size_t fr = 0, tot = 0;
            ctx->pool_cap = hipMemGetInfo(&fr, &tot) == hipSuccess ? tot / 4 : (size_t)4096 << 20;
        }
    }
    {   // XSDBA_HIP_ATOMIC_ORDER = probe (default) | assume (skip the probe) | broken (behave as if it had failed: tests)
        const char *m = getenv("XSDBA_HIP_ATOMIC_ORDER");
        if (m && !strcmp(m, "broken")) ctx->lane_ordered_atomics = false;
        else if (!(m && !strcmp(m, "assume"))) {
            unsigned long long *dbad = nullptr, hbad = ~0ull;
            if (hipMalloc((void **)&dbad, sizeof hbad) == hipSuccess && hipMemsetAsync(dbad, 0, sizeof hbad, ctx->stream) == hipSuccess) {
                hipLaunchKernelGGL(xs::k_atomic_order_probe, dim3(16), dim3(1024), 0, ctx->stream, 8, dbad);
                if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&hbad, dbad, sizeof hbad, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                    hipStreamSynchronize(ctx->stream) == hipSuccess)
                    ctx->lane_ordered_atomics = hbad == 0;
                else ctx->lane_ordered_atomics = false;
            } else ctx->lane_ordered_atomics = false;
            (void)hipGetLastError();
            if (dbad) (void)hipFree(dbad);
        }
    }
    *out = ctx;
    return 0;
}

API int xsdba_hip_lane_ordered_atomics(xsdba_hip_ctx *ctx, int *ordered)
{
    if (!ctx || !ordered) return fail(XSDBA_E_ARG, "null argument");
    *ordered = ctx->lane_ordered_atomics ? 1 : 0;
    return 0;
}

API int xsdba_hip_destroy(xsdba_hip_ctx *ctx)
{
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)xsdba_hip_comm_destroy(ctx);
    for (auto &p : ctx->plans) free_plan(p.get());
    for (auto &b : ctx->scratch) if (b.p) (void)hipFree(b.p);
    for (void *p : ctx->user_allocs) (void)hipFree(p);
    for (auto &e : ctx->pool) (void)hipFree(e.second);
    for (auto &e : ctx->prof) for (auto &pr : e.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    for (auto e : ctx->user_ev) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

API int xsdba_hip_synchronize(xsdba_hip_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

API int xsdba_hip_get_stream(xsdba_hip_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return fail(XSDBA_E_ARG, "null argument");
    *stream = (void *)ctx->stream;
    return 0;
}

// Buffers of 1 MB and more that are given back stay in a per-context pool (XSDBA_HIP_POOL_MB; default: a quarter of the device's
// memory) and serve the
// next request of exactly that size: a train / adjust loop allocates the same tables and series every pass, and hipFree costs a
// stream synchronisation each time.  Everything the library does with a buffer is ordered on the context's stream, so a pooled
// buffer may be handed out again while earlier kernels that used it are still queued.
namespace {
void pool_flush(xsdba_hip_ctx *ctx)
{
    if (ctx->pool.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->pool) (void)hipFree(e.second);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}
}  // namespace

API int xsdba_hip_malloc(xsdba_hip_ctx *ctx, size_t bytes, void **dptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dptr) return fail(XSDBA_E_ARG, "null out pointer");
    const size_t n = bytes ? bytes : 1;
    for (size_t i = ctx->pool.size(); i-- > 0;)
        if (ctx->pool[i].first == n) {
            *dptr = ctx->pool[i].second;
            ctx->pool_bytes -= n;
            ctx->pool.erase(ctx->pool.begin() + (long)i);
            ctx->user_allocs.push_back(*dptr);
            ctx->user_sizes.push_back(n);
            return 0;
        }
    if ((rc = dev_malloc(ctx, dptr, n))) return rc;
    ctx->user_allocs.push_back(*dptr);
    ctx->user_sizes.push_back(n);
    return 0;
}

API int xsdba_hip_free(xsdba_hip_ctx *ctx, void *dptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dptr) return 0;
    auto it = std::find(ctx->user_allocs.begin(), ctx->user_allocs.end(), dptr);
    if (it == ctx->user_allocs.end()) return fail(XSDBA_E_ARG, "pointer was not allocated by xsdba_hip_malloc on this context");
    const size_t idx = (size_t)(it - ctx->user_allocs.begin()), n = ctx->user_sizes[idx];
    ctx->user_allocs.erase(it);
    ctx->user_sizes.erase(ctx->user_sizes.begin() + (long)idx);
    const size_t cap = ctx->pool_cap;
    if (n >= ((size_t)1 << 20) && n <= cap) {
        if (ctx->pool_bytes + n > cap) {  // full: the oldest entries make room (sizes that are no longer asked for age out)
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            while (!ctx->pool.empty() && ctx->pool_bytes + n > cap) {
                (void)hipFree(ctx->pool.front().second);
                ctx->pool_bytes -= ctx->pool.front().first;
                ctx->pool.erase(ctx->pool.begin());
            }
        }
        ctx->pool.emplace_back(n, dptr);
        ctx->pool_bytes += n;
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(dptr));
    return 0;
}

API int xsdba_hip_memcpy_h2d(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

API int xsdba_hip_memcpy_d2h(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));  // also device -> device
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

API int xsdba_hip_host_register(xsdba_hip_ctx *ctx, void *host_ptr, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!host_ptr || !bytes) return fail(XSDBA_E_ARG, "null/empty host buffer");
    // read-write first; a read-only mapping (a numpy memmap opened with mode "r") can only be locked as read-only
    hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipHostRegister(host_ptr, bytes, hipHostRegisterReadOnly);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return fail((int)e, "hipHostRegister failed: %s", hipGetErrorString(e)); }
    return 0;
}

API int xsdba_hip_host_unregister(xsdba_hip_ctx *ctx, void *host_ptr)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!host_ptr) return 0;
    HIP_TRY(hipHostUnregister(host_ptr));
    return 0;
}

API int xsdba_hip_memcpy_async(xsdba_hip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    return 0;
}

API int xsdba_hip_event_record(xsdba_hip_ctx *ctx, int slot)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (slot < 0 || slot >= 16) return fail(XSDBA_E_ARG, "event slot %d out of range [0,16)", slot);
    if (!ctx->user_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&ctx->user_ev[slot], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ctx->user_ev[slot], ctx->stream));
    return 0;
}

// ---- multi-GPU: RCCL through dlopen (no link-time dependency) ---------------------------------------------------------
namespace {
struct Id128 { char b[XSDBA_COMM_ID_BYTES]; };  // ncclUniqueId (passed by value)
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return 0;
    // RCCL must come from the SAME ROCm as the HIP runtime this library runs on: a bare dlopen("librccl.so") may find another
    // copy first (a Python wheel's bundled one, built against its own libamdhip64 — its communicator then sees no device).
    // So: the directory the loaded libamdhip64 lives in first, then the loader's search path.
    void *h = nullptr;
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_GLOBAL);
        }
    }
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(XSDBA_E_ARG, "librccl.so not found (%s): the multi-GPU gather needs RCCL", dlerror());
    auto sym = [&](const char *n) { return dlsym(h, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))sym("ncclCommUserRank");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.CommCount || !g_rccl.CommUserRank ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send || !g_rccl.Recv || !g_rccl.GetErrorString) {
        dlclose(h);
        return fail(XSDBA_E_ARG, "librccl.so lacks a required entry point");
    }
    g_rccl.lib = h;
    return 0;
}

#define RCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        int r_ = (expr);                                                                                        \
        if (r_ != 0) return fail(XSDBA_E_ARG, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));               \
    } while (0)

// inside ncclGroupStart .. ncclGroupEnd: a failed call closes the group (result ignored) before the error is returned, so that
// later RCCL calls of this thread are not queued into a group nobody ends
#define RCCL_TRY_IN_GROUP(expr)                                                                                 \
    do {                                                                                                        \
        int r_ = (expr);                                                                                        \
        if (r_ != 0) {                                                                                          \
            (void)g_rccl.GroupEnd();                                                                            \
            return fail(XSDBA_E_ARG, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));                        \
        }                                                                                                       \
    } while (0)

void shard_block(int64_t n_cell, int n_ranks, int rank, int64_t *first, int64_t *count)
{
    const int64_t base = n_cell / n_ranks, rem = n_cell % n_ranks;
    *first = rank * base + std::min<int64_t>(rank, rem);
    *count = base + (rank < rem ? 1 : 0);
}
}  // namespace

API int xsdba_hip_shard_plan(int64_t n_cell, int n_ranks, int rank, int64_t *first_cell, int64_t *n_local)
{
    if (n_cell < 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !first_cell || !n_local)
        return fail(XSDBA_E_ARG, "bad shard_plan arguments (n_cell %lld, n_ranks %d, rank %d)", (long long)n_cell, n_ranks, rank);
    shard_block(n_cell, n_ranks, rank, first_cell, n_local);
    return 0;
}

API int xsdba_hip_comm_unique_id(void *id)
{
    if (!id) return fail(XSDBA_E_ARG, "null id buffer");
    int rc = rccl_load();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id));
    return 0;
}

API int xsdba_hip_comm_destroy(xsdba_hip_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (ctx->comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_rank = -1;
        ctx->comm_size = 0;
    }
    return 0;
}

API int xsdba_hip_comm_init(xsdba_hip_ctx *ctx, const void *id, int rank, int n_ranks)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(XSDBA_E_ARG, "bad comm_init arguments");
    if ((rc = rccl_load())) return rc;
    if (ctx->comm && (rc = xsdba_hip_comm_destroy(ctx))) return rc;
    Id128 tok;
    memcpy(tok.b, id, XSDBA_COMM_ID_BYTES);
    HIP_TRY(hipSetDevice(ctx->device));
    RCCL_TRY(g_rccl.CommInitRank(&ctx->comm, n_ranks, tok, rank));
    RCCL_TRY(g_rccl.CommUserRank(ctx->comm, &ctx->comm_rank));
    RCCL_TRY(g_rccl.CommCount(ctx->comm, &ctx->comm_size));
    return 0;
}

API int xsdba_hip_comm_info(xsdba_hip_ctx *ctx, int *rank, int *n_ranks)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (rank) *rank = ctx->comm_rank;
    if (n_ranks) *n_ranks = ctx->comm_size;
    return 0;
}

API int xsdba_hip_gather(xsdba_hip_ctx *ctx, const void *send, int64_t n_local, int64_t row_elems, int dtype, void *recv,
                         int64_t n_cell, int root)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ctx->comm) return fail(XSDBA_E_ARG, "xsdba_hip_comm_init has not been called on this context");
    if (root < 0 || root >= ctx->comm_size || row_elems <= 0 || n_cell < 0) return fail(XSDBA_E_ARG, "bad gather arguments");
    const size_t esz = dtype == XSDBA_F32 ? 4 : (dtype == XSDBA_F64 ? 8 : 0);
    if (!esz) return fail(XSDBA_E_ARG, "dtype must be XSDBA_F32 or XSDBA_F64");
    const int nccl_dt = dtype == XSDBA_F32 ? 7 : 8;  // ncclFloat32 / ncclFloat64
    int64_t first, count;
    shard_block(n_cell, ctx->comm_size, ctx->comm_rank, &first, &count);
    if (count != n_local) return fail(XSDBA_E_ARG, "rank %d owns %lld cells of %lld, not %lld", ctx->comm_rank, (long long)count, (long long)n_cell, (long long)n_local);
    if (n_local && (!send || !is_device_ptr(send))) return fail(XSDBA_E_ARG, "send must be a device buffer");
    if (ctx->comm_rank == root && n_cell && (!recv || !is_device_ptr(recv))) return fail(XSDBA_E_ARG, "recv must be a device buffer on the root");
    // one group of point-to-point transfers; every block travels in pieces of at most 2^28 elements (1 GiB of float32):
    // sender and receiver cut their block the same way, so the pieces pair up in order
    const size_t PIECE = (size_t)1 << 28;
    RCCL_TRY(g_rccl.GroupStart());
    if (ctx->comm_rank == root)
        for (int r = 0; r < ctx->comm_size; r++) {
            int64_t f, c;
            shard_block(n_cell, ctx->comm_size, r, &f, &c);
            const size_t total = (size_t)c * row_elems;
            char *base = (char *)recv + (size_t)f * row_elems * esz;
            for (size_t o = 0; o < total; o += PIECE)
                RCCL_TRY_IN_GROUP(g_rccl.Recv(base + o * esz, std::min(PIECE, total - o), nccl_dt, r, ctx->comm, ctx->stream));
        }
    {
        const size_t total = (size_t)n_local * row_elems;
        for (size_t o = 0; o < total; o += PIECE)
            RCCL_TRY_IN_GROUP(g_rccl.Send((const char *)send + o * esz, std::min(PIECE, total - o), nccl_dt, root, ctx->comm, ctx->stream));
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return 0;
}

API int xsdba_hip_event_synchronize(xsdba_hip_ctx *ctx, int slot)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (slot < 0 || slot >= 16) return fail(XSDBA_E_ARG, "event slot %d out of range [0,16)", slot);
    if (!ctx->user_ev[slot]) return 0;  // never recorded: nothing to wait for
    HIP_TRY(hipEventSynchronize(ctx->user_ev[slot]));
    return 0;
}

API int xsdba_hip_stream_wait_event(xsdba_hip_ctx *waiter, xsdba_hip_ctx *src, int slot)
{
    int rc = check_ctx(waiter);
    if (rc) return rc;
    if (!src || slot < 0 || slot >= 16) return fail(XSDBA_E_ARG, "bad source context / event slot");
    if (src->device != waiter->device) return fail(XSDBA_E_ARG, "both contexts must live on the same device");
    if (!src->user_ev[slot]) return 0;  // never recorded: nothing to wait for
    HIP_TRY(hipStreamWaitEvent(waiter->stream, src->user_ev[slot], 0));
    return 0;
}

API int xsdba_hip_memset(xsdba_hip_ctx *ctx, void *dst, int value, size_t bytes)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return 0;
}

API int xsdba_hip_profile_enable(xsdba_hip_ctx *ctx, int enable)
{
    if (!ctx) return fail(XSDBA_E_ARG, "null context");
    ctx->profiling = enable != 0;
    return 0;
}

API int xsdba_hip_profile_read(xsdba_hip_ctx *ctx, const char **names, int64_t *counts, double *total_ms, int cap, int *n,
                               int reset)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!n) return fail(XSDBA_E_ARG, "null argument");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int k = 0;
    for (auto &e : ctx->prof) {
        for (auto &pr : e.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { e.ms += ms; e.count++; }
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        e.pending.clear();
        if (k < cap && names && counts && total_ms) { names[k] = e.name; counts[k] = e.count; total_ms[k] = e.ms; }
        k++;
    }
    *n = std::min(k, cap);
    if (reset) for (auto &e : ctx->prof) { e.count = 0; e.ms = 0; }
    return 0;
}

API int xsdba_hip_timer_start(xsdba_hip_ctx *ctx)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return 0;
}

API int xsdba_hip_timer_stop(xsdba_hip_ctx *ctx, double *elapsed_ms)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(XSDBA_E_ARG, "null argument");
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *elapsed_ms = ms;
    return 0;
}

API int xsdba_hip_quantile_groups(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                  const int32_t *gidx, int32_t G, int32_t W, const void *q, int32_t Q, void *out)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (train_impl<float>(ctx, 0, x, nullptr, T, C, st, sc, gidx, G, W, q, Q, '+', nullptr, out, nullptr)),
                    (train_impl<double>(ctx, 0, x, nullptr, T, C, st, sc, gidx, G, W, q, Q, '+', nullptr, out, nullptr)));
}

API int xsdba_hip_eqm_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st,
                            int64_t sc, const int32_t *gidx, int32_t G, int32_t W, const void *q, int32_t Q, char kind,
                            void *af, void *hist_q)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    return DISPATCH(dtype, (train_impl<float>(ctx, 1, ref, hist, T, C, st, sc, gidx, G, W, q, Q, kind, af, hist_q, nullptr)),
                    (train_impl<double>(ctx, 1, ref, hist, T, C, st, sc, gidx, G, W, q, Q, kind, af, hist_q, nullptr)));
}

API int xsdba_hip_dqm_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st,
                            int64_t sc, const int32_t *gidx, int32_t G, int32_t W, const void *q, int32_t Q, char kind,
                            void *af, void *hist_q, void *scaling)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    return DISPATCH(dtype, (train_impl<float>(ctx, 2, ref, hist, T, C, st, sc, gidx, G, W, q, Q, kind, af, hist_q, scaling)),
                    (train_impl<double>(ctx, 2, ref, hist, T, C, st, sc, gidx, G, W, q, Q, kind, af, hist_q, scaling)));
}

API int xsdba_hip_qm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                            const void *af, const void *hist_q, int32_t G, int32_t Q, const int32_t *gidx, int interp,
                            int extrap, char kind, void *scen, int64_t ost, int64_t osc, int32_t *node_index)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, true))) return rc;
    return DISPATCH(dtype, (qm_adjust_impl<float>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, gidx, interp, extrap, kind, scen, ost, osc, node_index)),
                    (qm_adjust_impl<double>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, gidx, interp, extrap, kind, scen, ost, osc, node_index)));
}

API int xsdba_hip_qm_adjust_linear2d(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                     const void *af, const void *hist_q, int32_t G, int32_t Q, const double *group_coord,
                                     const double *new_group, int extrap, char kind, const void *trend, const double *lookup,
                                     int hist_q_is_row, void *scen, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, trend == nullptr))) return rc;
    return DISPATCH(dtype, (qm_adjust_linear2d_impl<float>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, group_coord, new_group, extrap, kind, trend, lookup, hist_q_is_row, scen, ost, osc)),
                    (qm_adjust_linear2d_impl<double>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, group_coord, new_group, extrap, kind, trend, lookup, hist_q_is_row, scen, ost, osc)));
}

API int xsdba_hip_qm_adjust_cubic2d(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                    const void *af, const void *hist_q, int32_t G, int32_t Q, const double *group_coord,
                                    const double *new_group, int extrap, char kind, const void *trend, const double *lookup,
                                    int hist_q_is_row, void *scen, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, trend == nullptr))) return rc;
    return DISPATCH(dtype, (qm_adjust_linear2d_impl<float>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, group_coord, new_group, extrap, kind, trend, lookup, hist_q_is_row, scen, ost, osc, 1)),
                    (qm_adjust_linear2d_impl<double>(ctx, sim, T, C, st, sc, af, hist_q, G, Q, group_coord, new_group, extrap, kind, trend, lookup, hist_q_is_row, scen, ost, osc, 1)));
}

API int xsdba_hip_dqm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                             const void *af, const void *hist_q, const void *scaling, int32_t G, int32_t Q,
                             const int32_t *gidx, int32_t W, const double *time_num, int32_t degree, int interp, int extrap,
                             char kind, void *scen, void *trend, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    return DISPATCH(dtype, (dqm_adjust_impl<float>(ctx, sim, T, C, st, sc, af, hist_q, scaling, G, Q, gidx, W, time_num, degree, interp, extrap, kind, scen, trend, ost, osc)),
                    (dqm_adjust_impl<double>(ctx, sim, T, C, st, sc, af, hist_q, scaling, G, Q, gidx, W, time_num, degree, interp, extrap, kind, scen, trend, ost, osc)));
}

API int xsdba_hip_qdm_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                             const void *af, const void *quantiles, int32_t G, int32_t Q, const int32_t *gidx, int32_t W,
                             int rank_window, int interp, int extrap, char kind, void *scen, double *sim_q, int64_t ost,
                             int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    if (!af) return fail(XSDBA_E_ARG, "null af");
    return DISPATCH(dtype, (qdm_impl<float>(ctx, sim, T, C, st, sc, af, quantiles, G, Q, gidx, W, rank_window, interp, extrap, kind, scen, sim_q, ost, osc)),
                    (qdm_impl<double>(ctx, sim, T, C, st, sc, af, quantiles, G, Q, gidx, W, rank_window, interp, extrap, kind, scen, sim_q, ost, osc)));
}

API int xsdba_hip_rank_groups(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                              const int32_t *gidx, int32_t G, int32_t W, int rank_window, double *out, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (qdm_impl<float>(ctx, x, T, C, st, sc, nullptr, nullptr, G, 0, gidx, W, rank_window, 0, 0, '+', nullptr, out, ost, osc)),
                    (qdm_impl<double>(ctx, x, T, C, st, sc, nullptr, nullptr, G, 0, gidx, W, rank_window, 0, 0, '+', nullptr, out, ost, osc)));
}

API int xsdba_hip_tail_mask(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                            const void *hist_q_raw, int32_t G, int32_t Q, const int32_t *gidx, double max_tail_factor,
                            void *scen, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (tail_mask_impl<float>(ctx, sim, T, C, st, sc, hist_q_raw, G, Q, gidx, max_tail_factor, scen, ost, osc)),
                    (tail_mask_impl<double>(ctx, sim, T, C, st, sc, hist_q_raw, G, Q, gidx, max_tail_factor, scen, ost, osc)));
}

API int xsdba_hip_jitter(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc, double lower,
                         double minimum, double upper, double maximum, uint64_t seed, int64_t first_cell, void *out, int64_t ost,
                         int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (jitter_impl<float>(ctx, x, T, C, st, sc, lower, minimum, upper, maximum, seed, first_cell, out, ost, osc)),
                    (jitter_impl<double>(ctx, x, T, C, st, sc, lower, minimum, upper, maximum, seed, first_cell, out, ost, osc)));
}

API int xsdba_hip_adapt_freq(xsdba_hip_ctx *ctx, const void *sim, const void *ref, int dtype, int64_t T, int64_t C, int64_t st,
                             int64_t sc, const int32_t *gidx, int32_t G, int32_t W, double thresh, uint64_t seed,
                             int64_t first_cell, double *P0_ref, double *P0_hist, void *pth, double *dP0, void *sim_ad,
                             int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (adapt_freq_impl<float>(ctx, sim, ref, T, C, st, sc, gidx, G, W, thresh, seed, first_cell, P0_ref, P0_hist, pth, dP0, sim_ad, ost, osc)),
                    (adapt_freq_impl<double>(ctx, sim, ref, T, C, st, sc, gidx, G, W, thresh, seed, first_cell, P0_ref, P0_hist, pth, dP0, sim_ad, ost, osc)));
}

API int xsdba_hip_group_means(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                              const int32_t *gidx, int32_t G, int32_t W, void *out)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (group_means_impl<float>(ctx, x, T, C, st, sc, gidx, G, W, out)),
                    (group_means_impl<double>(ctx, x, T, C, st, sc, gidx, G, W, out)));
}

API int xsdba_hip_scaling_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                 const void *af, const int32_t *gidx, int32_t G, const int32_t *ghi, const double *frac, char kind,
                                 void *scen, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (scaling_adjust_impl<float>(ctx, sim, T, C, st, sc, af, gidx, G, ghi, frac, kind, scen, ost, osc)),
                    (scaling_adjust_impl<double>(ctx, sim, T, C, st, sc, af, gidx, G, ghi, frac, kind, scen, ost, osc)));
}

API int xsdba_hip_loci_train(xsdba_hip_ctx *ctx, const void *ref, const void *hist, int dtype, int64_t T, int64_t C, int64_t st,
                             int64_t sc, const int32_t *gidx, int32_t G, int32_t W, double thresh, void *af, void *hist_thresh)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (loci_train_impl<float>(ctx, ref, hist, T, C, st, sc, gidx, G, W, thresh, af, hist_thresh)),
                    (loci_train_impl<double>(ctx, ref, hist, T, C, st, sc, gidx, G, W, thresh, af, hist_thresh)));
}

API int xsdba_hip_loci_adjust(xsdba_hip_ctx *ctx, const void *sim, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                              const void *af, const void *hist_thresh, const int32_t *gidx, int32_t G, const int32_t *ghi,
                              const double *frac, double thresh, void *scen, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (loci_adjust_impl<float>(ctx, sim, T, C, st, sc, af, hist_thresh, gidx, G, ghi, frac, thresh, scen, ost, osc)),
                    (loci_adjust_impl<double>(ctx, sim, T, C, st, sc, af, hist_thresh, gidx, G, ghi, frac, thresh, scen, ost, osc)));
}

API int xsdba_hip_polydetrend_trend(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                    const int32_t *gidx, int32_t G, int32_t W, const double *time_num, int32_t degree,
                                    void *trend, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (trend_impl<float>(ctx, x, T, C, st, sc, gidx, G, W, time_num, degree, 0, '+', trend, ost, osc)),
                    (trend_impl<double>(ctx, x, T, C, st, sc, gidx, G, W, time_num, degree, 0, '+', trend, ost, osc)));
}

API int xsdba_hip_series_correction(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                    const void *f, int64_t fst, int64_t fsc, char kind, int inverted, int skip_zeros, void *out,
                                    int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    return DISPATCH(dtype, (series_correction_impl<float>(ctx, x, T, C, st, sc, f, fst, fsc, kind, inverted ? 1 : 0, skip_zeros ? 1 : 0, out, ost, osc)),
                    (series_correction_impl<double>(ctx, x, T, C, st, sc, f, fst, fsc, kind, inverted ? 1 : 0, skip_zeros ? 1 : 0, out, ost, osc)));
}

API int xsdba_hip_rolling_mean(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                               int32_t window, int32_t min_periods, void *out, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (rolling_mean_impl<float>(ctx, x, T, C, st, sc, window, min_periods, out, ost, osc)),
                    (rolling_mean_impl<double>(ctx, x, T, C, st, sc, window, min_periods, out, ost, osc)));
}

API int xsdba_hip_polydetrend_fit(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                                  const int32_t *gidx, int32_t G, int32_t W, const double *time_num, int32_t degree,
                                  int preserve_mean, char kind, void *trend, int64_t ost, int64_t osc)
{
    int rc = check_ctx(ctx);
    if (rc || (rc = check_kind(kind, false))) return rc;
    return DISPATCH(dtype, (trend_impl<float>(ctx, x, T, C, st, sc, gidx, G, W, time_num, degree, preserve_mean ? 1 : 0, kind, trend, ost, osc)),
                    (trend_impl<double>(ctx, x, T, C, st, sc, gidx, G, W, time_num, degree, preserve_mean ? 1 : 0, kind, trend, ost, osc)));
}

API int xsdba_hip_synth_series(xsdba_hip_ctx *ctx, void *dst, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                               int generator, int array_id, uint64_t seed, int64_t cell_offset, float nan_cell_frac,
                               float nan_iid_frac)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dst || !is_device_ptr(dst) || T <= 0 || C <= 0 || generator < 0 || generator > 1 || array_id < 0 || array_id > 2)
        return fail(XSDBA_E_ARG, "synth_series needs a device destination and generator in {0,1}, array_id in {0,1,2}");
    LaunchScope ls(ctx, "k_synth");
    if (dtype == XSDBA_F32)
        hipLaunchKernelGGL(xs::k_synth<float>, dim3(4096), dim3(256), 0, ctx->stream, (float *)dst, T, C, st, sc, generator,
                           array_id, seed, cell_offset, nan_cell_frac, nan_iid_frac);
    else if (dtype == XSDBA_F64)
        hipLaunchKernelGGL(xs::k_synth<double>, dim3(4096), dim3(256), 0, ctx->stream, (double *)dst, T, C, st, sc, generator,
                           array_id, seed, cell_offset, nan_cell_frac, nan_iid_frac);
    else return fail(XSDBA_E_ARG, "bad dtype");
    return check_launch("k_synth");
}

API int xsdba_hip_vecquantiles(xsdba_hip_ctx *ctx, const void *x, int dtype, int64_t T, int64_t C, int64_t st, int64_t sc,
                               const void *rnk, void *out)
{
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return DISPATCH(dtype, (train_impl<float>(ctx, 3, x, nullptr, T, C, st, sc, nullptr, 1, 1, rnk, 1, '+', nullptr, out, nullptr)),
                    (train_impl<double>(ctx, 3, x, nullptr, T, C, st, sc, nullptr, 1, 1, rnk, 1, '+', nullptr, out, nullptr)));
}
