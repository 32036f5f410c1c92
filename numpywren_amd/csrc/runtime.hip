// runtime.hip -- device, memory, stream and event entry points of libnpw_hip.so.
//
// These replace the reference's storage / worker plumbing: S3 object GET/PUT of tiles
// (reference numpywren/matrix.py:497-533) becomes HBM allocations plus pinned-host async
// copies; the pywren worker pool and its asyncio read/compute/write pipeline (reference
// numpywren/job_runner.py:224-370) becomes HIP streams and events driven by the host.
#include "npw_internal.h"

namespace npw {

char* error_buffer() {
    static thread_local char buf[1024] = {0};
    return buf;
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 1024, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace npw

using npw::as_stream;

#include <dlfcn.h>
#include <cstdlib>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace npw {
int stream_cu_count_query(hipStream_t s);
void forget_stream(hipStream_t s);
void forget_side_streams(hipStream_t main);

namespace {
// helper streams per (caller's stream, calling thread): process-wide so that npw_stream_destroy can retire the entries of a
// stream -- a recycled handle may belong to a stream with another CU mask, and its helpers would carry the old one
std::mutex g_side_mutex;
std::unordered_map<hipStream_t, std::map<std::thread::id, SideStream>> g_side_table;
}  // namespace

void forget_side_streams(hipStream_t main) {
    std::map<std::thread::id, SideStream> dead;
    {
        std::lock_guard<std::mutex> lock(g_side_mutex);
        auto it = g_side_table.find(main);
        if (it == g_side_table.end()) return;
        dead.swap(it->second);
        g_side_table.erase(it);
    }
    for (auto& kv : dead) {   // (work still queued on a helper finishes before the driver releases it)
        SideStream& e = kv.second;
        for (hipStream_t st : {e.stream, e.stream2, e.stream3})
            if (st != nullptr) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : {e.fork, e.join, e.fork2, e.join2, e.fork3, e.join3})
            if (ev != nullptr) (void)hipEventDestroy(ev);
    }
    (void)hipGetLastError();
}

int side_stream(hipStream_t main, SideStream** out) {
    SideStream* slot;
    {
        std::lock_guard<std::mutex> lock(g_side_mutex);
        slot = &g_side_table[main][std::this_thread::get_id()];   // (node-based containers: the address is stable)
    }
    SideStream& e = *slot;
    if (e.stream == nullptr) {
        // A caller's stream restricted to some compute units (npw_stream_create_masked: the executor's partitions) gets
        // helpers with the SAME mask: what a call forks off must not spill onto the CUs its stream was kept away from.
        const int cus = device_cu_count();
        const uint32_t words = (uint32_t)((cus + 31) / 32);
        uint32_t mask[16] = {0};
        bool masked = false;
        if (main != nullptr && words <= 16 && hipExtStreamGetCUMask(main, words, mask) == hipSuccess) {
            int bits = 0;
            for (uint32_t w = 0; w < words; ++w) bits += __builtin_popcount(mask[w]);
            masked = bits > 0 && bits < cus;
        } else {
            (void)hipGetLastError();
        }
        auto make = [&](hipStream_t* out) -> hipError_t {
            return masked ? hipExtStreamCreateWithCUMask(out, words, mask) : hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        };
        NPW_HIP_CHECK(make(&e.stream));
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.fork, hipEventDisableTiming));
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.join, hipEventDisableTiming));
        NPW_HIP_CHECK(make(&e.stream2));
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.fork2, hipEventDisableTiming));
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.join2, hipEventDisableTiming));
        // the throughput helper: optionally kept off the leading `reserve` bits of the CU mask (the bits are dealt
        // round-robin over the XCDs, so every XCD keeps reserve / 8 CUs free of its ~100 us GEMM workgroups for the
        // latency-bound launches of the other streams)
        static const int reserve = [] {
            const char* v = getenv("NPW_QR_FAR_RESERVE_CUS");
            return v ? atoi(v) : 0;
        }();
        if (!masked && reserve > 0 && reserve < cus && cus <= 512) {
            uint32_t rest[16] = {0};
            for (int cu = reserve; cu < cus; ++cu) rest[cu / 32] |= 1u << (cu % 32);
            NPW_HIP_CHECK(hipExtStreamCreateWithCUMask(&e.stream3, words, rest));
        } else {
            NPW_HIP_CHECK(make(&e.stream3));
        }
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.fork3, hipEventDisableTiming));
        NPW_HIP_CHECK(hipEventCreateWithFlags(&e.join3, hipEventDisableTiming));
    }
    *out = &e;
    return NPW_OK;
}

namespace {
std::mutex g_cu_cache_mutex;
std::unordered_map<hipStream_t, int> g_cu_cache;   // process-wide, so that npw_stream_destroy can drop an entry
}  // namespace

int stream_cu_count(hipStream_t s) {
    // (cached per stream: the mask of a stream never changes, the query is a driver call; the entry dies with the stream --
    //  a recycled handle may belong to a stream with another mask)
    {
        std::lock_guard<std::mutex> lock(g_cu_cache_mutex);
        auto it = g_cu_cache.find(s);
        if (it != g_cu_cache.end()) return it->second;
    }
    const int result = stream_cu_count_query(s);
    std::lock_guard<std::mutex> lock(g_cu_cache_mutex);
    g_cu_cache[s] = result;
    return result;
}

void forget_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_cu_cache_mutex);
    g_cu_cache.erase(s);
}

namespace {
std::atomic<int> g_live_comms{0};
constexpr int COMM_RESERVE_DEFAULT = 64;
}  // namespace

void comm_live_changed(int delta) { g_live_comms.fetch_add(delta); }

int comm_reserved_cus() {
    // RCCL launches ONE kernel per grouped exchange with one workgroup per channel it uses; measured on this part
    // (profiles/r05_rccl_headroom.md, rocprofv3 kernel trace): ncclDevKernel_Generic_1 with 24 workgroups of 256 threads
    // for one send / receive pair, 64 -- RCCL's channel limit on this part -- for a group of 8 pairs; 37 KiB of LDS and
    // 124 VGPRs each.  One compute unit per channel is set aside: a parked transfer workgroup takes one of the CU's two big
    // slots AND enough LDS that a 133 KiB Cholesky workgroup no longer fits beside it.
    static const int reserve = [] {
        const char* e = getenv("NPW_COMM_RESERVE_CUS");
        return e ? atoi(e) : COMM_RESERVE_DEFAULT;
    }();
    return g_live_comms.load() > 0 ? reserve : 0;
}

int resident_cu_count(hipStream_t s) {
    const int cus = stream_cu_count(s), keep = comm_reserved_cus();
    return cus - keep > 0 ? cus - keep : 1;
}

int device_cu_count() {
    static const int device_cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return device_cus;
}

int stream_cu_count_query(hipStream_t s) {
    const int device_cus = device_cu_count();
    int cus = device_cus;
    uint32_t mask[16] = {0};
    const int words = (device_cus + 31) / 32;
    if (s != nullptr && words <= 16) {
        if (hipExtStreamGetCUMask(s, (uint32_t)words, mask) == hipSuccess) {
            int bits = 0;
            for (int w = 0; w < words; ++w) bits += __builtin_popcount(mask[w]);
            if (bits > 0 && bits < cus) cus = bits;
        } else {
            (void)hipGetLastError();
        }
    }
    return cus;
}

}  // namespace npw

extern "C" {

int npw_version(void) { return 110; }   // 110: npw_dgemm_nt_sub_batched_workspace_bytes (the batched workspace is per problem), npw_dmul, npw_dflip

namespace {
struct Roctx {
    bool tried = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
Roctx g_roctx;
std::mutex g_roctx_mutex;
void load_roctx() {
    std::lock_guard<std::mutex> lock(g_roctx_mutex);
    if (g_roctx.tried) return;
    g_roctx.tried = true;
    for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h == nullptr) continue;
        auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push != nullptr && pop != nullptr) {
            g_roctx.push = push;
            g_roctx.pop = pop;
            return;
        }
    }
}
}  // namespace

int npw_range_push(const char* name) {
    NPW_REQUIRE(name != nullptr, "npw_range_push: NULL name");
    if (!g_roctx.tried) load_roctx();
    return g_roctx.push != nullptr ? g_roctx.push(name) : 0;
}

int npw_range_pop(void) {
    if (!g_roctx.tried) load_roctx();
    return g_roctx.pop != nullptr ? g_roctx.pop() : 0;
}

const char* npw_last_error(void) { return npw::error_buffer(); }

int npw_device_count(int* count) {
    NPW_REQUIRE(count != nullptr, "npw_device_count: NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return npw::set_error(NPW_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = c;
    return NPW_OK;
}

int npw_set_device(int device) {
    NPW_HIP_CHECK(hipSetDevice(device));
    return NPW_OK;
}

int npw_get_device(int* device) {
    NPW_REQUIRE(device != nullptr, "npw_get_device: NULL");
    NPW_HIP_CHECK(hipGetDevice(device));
    return NPW_OK;
}

int npw_device_info(int device, char* name, size_t name_len, size_t* total_mem_bytes,
                    int* compute_units, int* clock_khz) {
    hipDeviceProp_t prop;
    NPW_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    return NPW_OK;
}

int npw_device_pci_bus_id(int device, char* out, size_t out_len) {
    NPW_REQUIRE(out != nullptr && out_len >= 16, "npw_device_pci_bus_id: need a buffer of at least 16 bytes");
    out[0] = 0;
    NPW_HIP_CHECK(hipDeviceGetPCIBusId(out, (int)out_len, device));
    return NPW_OK;
}

int npw_mem_info(size_t* free_bytes, size_t* total_bytes) {
    size_t f = 0, t = 0;
    NPW_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return NPW_OK;
}

int npw_malloc(void** dptr, size_t bytes) {
    NPW_REQUIRE(dptr != nullptr, "npw_malloc: NULL");
    *dptr = nullptr;
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMalloc(dptr, bytes));
    return NPW_OK;
}

int npw_free(void* dptr) {
    if (dptr) NPW_HIP_CHECK(hipFree(dptr));
    return NPW_OK;
}

int npw_host_alloc(void** hptr, size_t bytes) {
    NPW_REQUIRE(hptr != nullptr, "npw_host_alloc: NULL");
    *hptr = nullptr;
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return NPW_OK;
}

int npw_host_free(void* hptr) {
    if (hptr) NPW_HIP_CHECK(hipHostFree(hptr));
    return NPW_OK;
}

int npw_memcpy_h2d_async(void* dst, const void* src, size_t bytes, npw_stream_t stream) {
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return NPW_OK;
}

int npw_memcpy_d2h_async(void* dst, const void* src, size_t bytes, npw_stream_t stream) {
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return NPW_OK;
}

int npw_memcpy_d2d_async(void* dst, const void* src, size_t bytes, npw_stream_t stream) {
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return NPW_OK;
}

int npw_memset_async(void* dst, int byte_value, size_t bytes, npw_stream_t stream) {
    if (bytes == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemsetAsync(dst, byte_value, bytes, as_stream(stream)));
    return NPW_OK;
}

int npw_memcpy2d_h2d_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream) {
    if (row_bytes == 0 || rows == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows,
                                   hipMemcpyHostToDevice, as_stream(stream)));
    return NPW_OK;
}

int npw_memcpy2d_d2h_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream) {
    if (row_bytes == 0 || rows == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows,
                                   hipMemcpyDeviceToHost, as_stream(stream)));
    return NPW_OK;
}

int npw_memcpy2d_d2d_async(void* dst, size_t dpitch, const void* src, size_t spitch,
                           size_t row_bytes, size_t rows, npw_stream_t stream) {
    if (row_bytes == 0 || rows == 0) return NPW_OK;
    NPW_HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows,
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
    return NPW_OK;
}

int npw_stream_create(npw_stream_t* stream, int high_priority) {
    NPW_REQUIRE(stream != nullptr, "npw_stream_create: NULL");
    hipStream_t s;
    if (high_priority) {
        int lo = 0, hi = 0;
        NPW_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        NPW_HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    } else {
        NPW_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    *stream = reinterpret_cast<npw_stream_t>(s);
    return NPW_OK;
}

namespace {
// CU-masked streams are never handed back to the HIP runtime: create / destroy cycles of masked streams hang inside this
// ROCm release about every tenth cycle (with nothing but one GEMM on the stream in between; ADVICE / VERDICT r4: the export
// was a trap for any caller but the backend, which kept its masked streams for life).  npw_stream_destroy PARKS a masked
// stream -- idle, with its helper streams and cached CU count, which carry the same mask -- and npw_stream_create_masked
// hands a parked stream of the same mask out again.  A process uses a handful of distinct masks, so the park stays small.
std::mutex g_masked_mutex;
std::unordered_map<hipStream_t, std::vector<uint32_t>> g_masked;        // live masked streams -> their mask
std::map<std::vector<uint32_t>, std::vector<hipStream_t>> g_parked;      // mask -> idle streams waiting for a new owner
}  // namespace

int npw_stream_create_masked(npw_stream_t* stream, const uint32_t* cu_mask, int words) {
    NPW_REQUIRE(stream != nullptr && cu_mask != nullptr && words > 0, "npw_stream_create_masked: bad arguments");
    std::vector<uint32_t> mask(cu_mask, cu_mask + words);
    hipStream_t s = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_masked_mutex);
        auto it = g_parked.find(mask);
        if (it != g_parked.end() && !it->second.empty()) {
            s = it->second.back();
            it->second.pop_back();
            g_masked[s] = mask;
        }
    }
    if (s == nullptr) {
        NPW_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask));
        std::lock_guard<std::mutex> lock(g_masked_mutex);
        g_masked[s] = std::move(mask);
    }
    *stream = reinterpret_cast<npw_stream_t>(s);
    return NPW_OK;
}

int npw_stream_destroy(npw_stream_t stream) {
    if (stream) {
        hipStream_t s = as_stream(stream);
        {
            std::unique_lock<std::mutex> lock(g_masked_mutex);
            auto it = g_masked.find(s);
            if (it != g_masked.end()) {
                std::vector<uint32_t> mask = std::move(it->second);
                g_masked.erase(it);
                lock.unlock();
                NPW_HIP_CHECK(hipStreamSynchronize(s));   // parked idle; helpers and CU count stay valid (same mask)
                lock.lock();
                g_parked[mask].push_back(s);
                return NPW_OK;
            }
        }
        npw::forget_stream(s);
        npw::forget_side_streams(s);
        NPW_HIP_CHECK(hipStreamDestroy(s));
    }
    return NPW_OK;
}

int npw_stream_cu_count(npw_stream_t stream, int* compute_units, int* resident_units) {
    if (compute_units != nullptr) *compute_units = npw::stream_cu_count(static_cast<hipStream_t>(stream));
    if (resident_units != nullptr) *resident_units = npw::resident_cu_count(static_cast<hipStream_t>(stream));
    return NPW_OK;
}

int npw_stream_synchronize(npw_stream_t stream) {
    NPW_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return NPW_OK;
}

int npw_stream_query(npw_stream_t stream, int* done) {
    NPW_REQUIRE(done != nullptr, "npw_stream_query: NULL");
    hipError_t e = hipStreamQuery(as_stream(stream));
    if (e == hipSuccess) {
        *done = 1;
    } else if (e == hipErrorNotReady) {
        *done = 0;
        (void)hipGetLastError();
    } else {
        return npw::set_error(NPW_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
    }
    return NPW_OK;
}

int npw_device_synchronize(void) {
    NPW_HIP_CHECK(hipDeviceSynchronize());
    return NPW_OK;
}

int npw_event_create(npw_event_t* event, int timing) {
    NPW_REQUIRE(event != nullptr, "npw_event_create: NULL");
    hipEvent_t e;
    NPW_HIP_CHECK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
    *event = reinterpret_cast<npw_event_t>(e);
    return NPW_OK;
}

int npw_event_destroy(npw_event_t event) {
    if (event) NPW_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return NPW_OK;
}

int npw_event_record(npw_event_t event, npw_stream_t stream) {
    NPW_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return NPW_OK;
}

int npw_event_synchronize(npw_event_t event) {
    NPW_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return NPW_OK;
}

int npw_event_query(npw_event_t event, int* done) {
    NPW_REQUIRE(done != nullptr, "npw_event_query: NULL");
    hipError_t e = hipEventQuery(reinterpret_cast<hipEvent_t>(event));
    if (e == hipSuccess) {
        *done = 1;
    } else if (e == hipErrorNotReady) {
        *done = 0;
        (void)hipGetLastError();
    } else {
        return npw::set_error(NPW_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
    }
    return NPW_OK;
}

int npw_stream_wait_event(npw_stream_t stream, npw_event_t event) {
    NPW_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return NPW_OK;
}

int npw_event_elapsed_ms(npw_event_t start, npw_event_t stop, float* ms) {
    NPW_REQUIRE(ms != nullptr, "npw_event_elapsed_ms: NULL");
    NPW_HIP_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start),
                                      reinterpret_cast<hipEvent_t>(stop)));
    return NPW_OK;
}

}  // extern "C"
