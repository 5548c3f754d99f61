// Process-wide state of libdasp_hip.so that is not a kernel: the sticky device error words (common.hpp, "look-back words and the sticky
// device error"), the look-back plan switch, the occupancy query behind the look-back gate, and three test-support entry points
// (a time-out provoked on purpose, a kernel that keeps the CUs busy, a stream confined to a few CUs).
#include "common.hpp"

#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

namespace dasp {
namespace {
constexpr int MAX_DEV = 64;
unsigned* g_err_host[MAX_DEV];
unsigned* g_err_dev[MAX_DEV];
std::atomic<int> g_lookback{1};

int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEV ? dev : -1;
}
std::mutex g_err_mu;
bool ensure(int dev) {
    if (dev < 0) return false;
    std::lock_guard<std::mutex> lock(g_err_mu);          // (two host threads may make their first call on a device together)
    if (g_err_host[dev]) return true;
    // 64 bytes of host memory the device writes straight into: nothing is copied and nothing is polled on the fast path - a kernel
    // touches it only when it gives a word up, the host reads plain memory. (One allocation per device and process, made on the first
    // call that can need it - dasp_device_error() at library load from the Python side - so never inside a stream capture.)
    void* h = nullptr;
    if (hipHostMalloc(&h, DASP_DEVERR_WORDS * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return false;
    std::memset(h, 0, DASP_DEVERR_WORDS * sizeof(unsigned));
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return false; }
    g_err_dev[dev] = static_cast<unsigned*>(d);
    g_err_host[dev] = static_cast<unsigned*>(h);
    return true;
}

__global__ void stall_kernel(const unsigned long long* never, unsigned* err, float* out) {
    const float v = lookback_poll(never, 0xFFFFFFFFu, err, DASP_DEVERR_TEST);       // the word is zero and stays zero
    if (out) *out = v;
}
__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) acc = acc * 1664525u + 1013904223u;
    if (acc == 0x12345678u && sink) *sink = acc;
}
}  // namespace

unsigned* error_words_device() {
    const int dev = current_device();
    return ensure(dev) ? g_err_dev[dev] : nullptr;
}
int error_pending() {
    const int dev = current_device();
    if (dev < 0 || !g_err_host[dev]) return 0;
    int bits = 0;
    for (int i = 0; i < DASP_DEVERR_TIMEOUT_SLOT; ++i)
        if (reinterpret_cast<volatile unsigned*>(g_err_host[dev])[i]) bits |= 1 << i;
    return bits;
}
int lookback_enabled() { return g_lookback.load(std::memory_order_relaxed); }
bool lookback_has_room(const void* kernel, int threads) {
    // (asked once per kernel and device: the occupancy query is a runtime call on the small-batch path, where host time is the step)
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, bool> seen;
    const int dev = current_device();
    if (dev < 0) return false;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(kernel, dev);
    const auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    int per_cu = 0, cus = 0;
    bool ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) == hipSuccess &&
              hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && (long)per_cu * cus >= 64;
    seen[key] = ok;
    return ok;
}
}  // namespace dasp

using namespace dasp;

extern "C" {

int dasp_device_error(void) {
    error_words_device();           // (allocates the words of the current device on the first call)
    return error_pending();
}
void dasp_device_error_clear(void) {
    const int dev = current_device();
    if (dev >= 0 && g_err_host[dev])
        for (int i = 0; i < DASP_DEVERR_TIMEOUT_SLOT; ++i) reinterpret_cast<volatile unsigned*>(g_err_host[dev])[i] = 0u;
}
int dasp_plan_lookback(int on) {
    if (on >= 0) g_lookback.store(on ? 1 : 0, std::memory_order_relaxed);
    return g_lookback.load(std::memory_order_relaxed);
}

/* ---- test support ---- */
int dasp_test_lookback_timeout(int milliseconds) {          // 0 = the default (2 s)
    unsigned* d = error_words_device();
    const int dev = current_device();
    if (!d || milliseconds < 0) return DASP_ERR_ARG;
    reinterpret_cast<volatile unsigned*>(g_err_host[dev])[DASP_DEVERR_TIMEOUT_SLOT] = (unsigned)milliseconds;
    return DASP_OK;
}
int dasp_test_lookback_stall(const unsigned long long* zero_word, float* out, void* stream) {
    unsigned* d = error_words_device();
    if (!d || !zero_word) return DASP_ERR_ARG;
    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, zero_word, d, out);
    return (int)hipGetLastError();
}
int dasp_test_spin(int workgroups, int threads, double milliseconds, void* stream) {
    if (workgroups <= 0 || threads <= 0 || threads > 1024 || milliseconds < 0) return DASP_ERR_ARG;
    hipLaunchKernelGGL(spin_kernel, dim3(workgroups), dim3(threads), 0, (hipStream_t)stream, (unsigned long long)(milliseconds * 1e5), (unsigned*)nullptr);
    return (int)hipGetLastError();
}
int dasp_test_stream_with_cus(int n_cus, void** stream) {
    if (!stream || n_cus <= 0 || n_cus > 256) return DASP_ERR_ARG;
    unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_cus; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    *stream = (void*)s;
    return (int)e;
}
int dasp_test_stream_destroy(void* stream) { return (int)hipStreamDestroy((hipStream_t)stream); }

}  // extern "C"
