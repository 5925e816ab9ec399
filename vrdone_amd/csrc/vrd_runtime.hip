// Error string + event-based per-family profiling for libvrdone_hip.so.
#include "vrd_common.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <utility>
#include <cstring>

namespace vrd {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static bool g_prof_on = false;
static unsigned long long g_prof_mask = ~0ull;      // kernel families that record events (vrd_prof_select)

struct ProfRec {
    int id;
    hipEvent_t e0, e1;
    double flops, bytes;
};
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free_events;
static double g_ms[VRD_K_COUNT], g_flops[VRD_K_COUNT], g_bytes[VRD_K_COUNT];
static int64_t g_launches[VRD_K_COUNT];
static double g_skipped[VRD_K_COUNT];        // launched-but-skipped FLOPs (padding maps), folded in by drain()
double take_big_skipped_flops();             // vrd_gemm_x3_big.hip
double take_f32_skipped_flops();             // vrd_gemm.hip

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int reserve_lds(const void* kernel, size_t bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    static std::vector<std::pair<const void*, int>> done;
    for (const auto& d : done)
        if (d.first == kernel && d.second == dev) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: cannot reserve %zu B of LDS on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return -2;
    }
    done.emplace_back(kernel, dev);
    return 0;
}

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    static std::vector<int> cus;           // by device ordinal, 0 = not asked yet
    if (dev >= (int)cus.size()) cus.resize(dev + 1, 0);
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// The f16 operand-range flag of the CURRENT device: one word of device memory per device ordinal, allocated on first use and
// kept for the life of the process.  Every producer of VRD_PAIR_F16 rows ORs its tag into it when an element's scaled value
// does not fit an f16 (vrd_common.h, RangeTrack).  nullptr when the allocation fails (producers then skip the report).
unsigned* range_flag() {
    // hot path (every producer launch): a lock-free read of the per-device table; the slow path below runs once per device
    constexpr int MAX_DEV = 64;
    static std::atomic<unsigned*> flags[MAX_DEV];
    static std::atomic<bool> failed[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= MAX_DEV) return nullptr;
    if (unsigned* q = flags[dev].load(std::memory_order_acquire)) return q;
    if (failed[dev].load(std::memory_order_relaxed)) return nullptr;          // (an allocation that failed is not retried per launch)
    std::lock_guard<std::mutex> lk(g_mu);
    if (unsigned* q = flags[dev].load(std::memory_order_acquire)) return q;
    // (allocation and a default-stream memset are not legal inside a stream capture: the word must exist before the first one --
    // every recording path of the host mirror calls ops.f16_range_flag(device) first: train_graph's warm-up,
    // MaskVRD.forward_training, the eval recordings)
    unsigned* q = nullptr;
    if (hipMalloc(&q, 64) != hipSuccess) {
        failed[dev].store(true);
        return nullptr;
    }
    if (hipMemset(q, 0, 64) != hipSuccess) {
        (void)hipFree(q);
        failed[dev].store(true);
        return nullptr;
    }
    flags[dev].store(q, std::memory_order_release);
    return q;
}

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(int kernel_id, hipStream_t s, double flops, double bytes)
    : id(kernel_id), stream(s), slot(nullptr) {
    if (!g_prof_on || !((g_prof_mask >> kernel_id) & 1ull)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRec r;
    r.id = kernel_id;
    r.e0 = get_event();
    r.e1 = get_event();
    r.flops = flops;
    r.bytes = bytes;
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    slot = reinterpret_cast<void*>(g_recs.size());   // index + 1
}

ProfScope::~ProfScope() {
    if (!slot) return;
    std::lock_guard<std::mutex> lk(g_mu);
    size_t idx = reinterpret_cast<size_t>(slot) - 1;
    if (idx < g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, stream);
}

// fold finished records into the per-family totals (synchronises their events)
static void drain() {
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            g_ms[r.id] += ms;
            g_flops[r.id] += r.flops;
            g_bytes[r.id] += r.bytes;
            g_launches[r.id] += 1;
        }
        g_free_events.push_back(r.e0);
        g_free_events.push_back(r.e1);
    }
    g_recs.clear();
    g_skipped[VRD_K_GEMM_X3_BIG] += take_big_skipped_flops();     // synchronous copy: every launch above has finished
    g_skipped[VRD_K_GEMM] += take_f32_skipped_flops();
}

}  // namespace vrd

extern "C" {

int vrd_abi_version(void) { return VRD_ABI_VERSION; }
const char* vrd_last_error(void) { return vrd::g_err; }

int vrd_f16_range_flag(void** flag) {
    VRD_CHECK_ARG(flag, "vrd_f16_range_flag: null pointer");
    *flag = vrd::range_flag();
    if (!*flag) {
        vrd::set_error("vrd_f16_range_flag: cannot allocate the flag word");
        return -2;
    }
    return 0;
}

int vrd_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(vrd::g_mu);
    vrd::g_prof_on = on != 0;
    return 0;
}

int vrd_prof_select(unsigned long long family_mask) {
    std::lock_guard<std::mutex> lk(vrd::g_mu);
    vrd::g_prof_mask = family_mask;
    return 0;
}

int vrd_prof_reset(void) {
    std::lock_guard<std::mutex> lk(vrd::g_mu);
    vrd::drain();
    memset(vrd::g_ms, 0, sizeof(vrd::g_ms));
    memset(vrd::g_flops, 0, sizeof(vrd::g_flops));
    memset(vrd::g_bytes, 0, sizeof(vrd::g_bytes));
    memset(vrd::g_launches, 0, sizeof(vrd::g_launches));
    memset(vrd::g_skipped, 0, sizeof(vrd::g_skipped));
    return 0;
}

int vrd_prof_read_skipped(int kernel_id, double* flops_skipped) {
    VRD_CHECK_ARG(kernel_id >= 0 && kernel_id < VRD_K_COUNT && flops_skipped, "vrd_prof_read_skipped: bad arguments");
    std::lock_guard<std::mutex> lk(vrd::g_mu);
    vrd::drain();
    *flops_skipped = vrd::g_skipped[kernel_id];
    return 0;
}

int vrd_prof_read(int kernel_id, double* ms, int64_t* launches, double* flops, double* bytes) {
    VRD_CHECK_ARG(kernel_id >= 0 && kernel_id < VRD_K_COUNT, "vrd_prof_read: bad kernel id %d", kernel_id);
    std::lock_guard<std::mutex> lk(vrd::g_mu);
    vrd::drain();
    if (ms) *ms = vrd::g_ms[kernel_id];
    if (launches) *launches = vrd::g_launches[kernel_id];
    if (flops) *flops = vrd::g_flops[kernel_id];
    if (bytes) *bytes = vrd::g_bytes[kernel_id];
    return 0;
}

}  // extern "C"
