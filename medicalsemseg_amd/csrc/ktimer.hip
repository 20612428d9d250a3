// Event pairs directly around the launches of the main kernels: the measurement bench.py's `roofline.achieved` rests on.
// Off by default (one relaxed load per launch); never enabled while a stream is capturing (bench.py instruments eager
// steps after its timed region).  The pairs bracket ONE kernel each -- unlike the event pairs hip.py puts around a whole
// entry point, which also cover that entry point's follow-up kernels (statistics finalize, slab reduction).
#include <atomic>
#include <mutex>
#include <string.h>
#include <vector>

#include "common.h"

namespace {
struct KtRec {
    const char* name;
    hipEvent_t e0, e1;
};
std::atomic<int> g_on{0};
std::mutex g_mu;
std::vector<KtRec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool msseg_ktimer_on() { return g_on.load(std::memory_order_relaxed) != 0; }

int msseg_ktimer_begin(const char* name, hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_recs.size() >= 200000) return -1;
    KtRec r{name, get_event(), get_event()};
    if (r.e0 == nullptr || r.e1 == nullptr || hipEventRecord(r.e0, stream) != hipSuccess) return -1;
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void msseg_ktimer_end(int slot, hipStream_t stream) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].e1, stream);
}

extern "C" {

int msseg_ktimer_enable(int on) {
    g_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return MSSEG_OK;
}

int msseg_ktimer_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) {
        g_pool.push_back(r.e0);
        g_pool.push_back(r.e1);
    }
    g_recs.clear();
    return MSSEG_OK;
}

int msseg_ktimer_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_recs.size();
}

int msseg_ktimer_get(int i, char* name, int cap, float* ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (i < 0 || i >= (int)g_recs.size() || name == nullptr || ms == nullptr || cap < 2)
        MSSEG_FAIL(MSSEG_EINVAL, "ktimer_get: bad index %d / buffer", i);
    const KtRec& r = g_recs[i];
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess)
        MSSEG_FAIL(MSSEG_ELAUNCH, "ktimer_get: record %d has no elapsed time", i);
    strncpy(name, r.name, (size_t)cap - 1);
    name[cap - 1] = 0;
    return MSSEG_OK;
}

}  // extern "C"
