// afp_host.hip -- host-side services of libafp_hip.so that are not tied to one stage of the path: the retire list and
// grow-only device buffers, handle synchronisation and timing events, the persistent host thread pool with the ring
// download and the background prefault, pinned host memory, what runtime the process bound.
#include "afp_internal.h"

thread_local std::string g_hip_err;

// Buffers that had to grow leave their old allocation HERE instead of calling hipFree on the spot: hipFree waits for every
// stream of the device (r04: 8 ms in the middle of the pipelined c4 job, behind two queued uploads), hipMalloc does not.
// The retired allocations are released in one go at a moment that is idle anyway -- the end of a batch whose results are
// being fetched (finalize), the end of a table download, afp_destroy -- once they add up to AFP_RETIRE_MAX_MB (default
// 1024), or at once if an allocation fails.  Releasing them is safe at any time (hipFree's own wait makes it so); the list
// only decides WHEN the wait is paid.  Process-wide, per device.
struct Retired { int device; void* p; size_t bytes; };
static std::mutex g_retire_mu;
static std::vector<Retired> g_retired;
static size_t g_retired_bytes = 0;
static size_t retire_limit()
{
    static size_t lim = 0;
    if (!lim) { const char* e = getenv("AFP_RETIRE_MAX_MB"); lim = ((size_t)(e && atol(e) >= 0 ? atol(e) : 1024) << 20) + 1; }
    return lim;
}
void drain_retired(bool force)
{
    std::vector<Retired> take;
    {
        std::lock_guard<std::mutex> g(g_retire_mu);
        if (g_retired.empty() || (!force && g_retired_bytes < retire_limit())) return;
        take.swap(g_retired);
        g_retired_bytes = 0;
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (const Retired& r : take) { if (hipSetDevice(r.device) == hipSuccess) (void)hipFree(r.p); }
    if (have_cur) (void)hipSetDevice(cur);
}
extern "C" int64_t afp_retired_bytes(void) { std::lock_guard<std::mutex> g(g_retire_mu); return (int64_t)g_retired_bytes; }

int ensure(DevBuf& b, size_t bytes, bool rows)
{
    if (bytes <= b.cap && b.p) return AFP_OK;
    if (bytes == 0) bytes = 256;
    const bool regrow = b.p != nullptr;
    if (b.p) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(g_retire_mu);
        g_retired.push_back(Retired{dev, b.p, b.cap});
        g_retired_bytes += b.cap;
        b.p = nullptr; b.cap = 0;
    }
    // a buffer that has to GROW gets headroom: batches of a real ingest differ by a few rows.  First allocations are exact,
    // except buffers sized by a batch's ROW count (`rows`: the next batch of the same shape has a few rows more or less):
    // those start with an eighth to spare.
    size_t want = bytes;
    if (regrow) want += bytes >= ((size_t)1 << 30) ? bytes / 8 : bytes / 4;
    else if (rows) want += bytes / 8;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        // out of memory with allocations parked on the retire list: release them (this is the wait the list postpones), then
        // once more, exact size last
        (void)hipGetLastError();
        drain_retired(true);
        e = hipMalloc(&b.p, want);
        if (e != hipSuccess && want != bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&b.p, want); }
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
        b.p = nullptr;
        return AFP_ERR_NOMEM;
    }
    b.cap = want;
    return AFP_OK;
}

// Wait (on the host) for everything this handle has queued: a staged batch is joined through its completion
// event -- NOT by making the handle's stream wait for it: HIP multiplexes streams onto a few hardware queues,
// and a queue barrier parked on a stream that shares its queue with a stage stream would stall the stages
// of the other handles behind it.
hipError_t sync_handle(afp_handle* h)
{
    if (h->join_pending) {
        hipError_t e = hipEventSynchronize(h->ev_b);
        if (e != hipSuccess) return e;
        h->join_pending = false;
    }
    return hipStreamSynchronize(h->stream);
}

hipEvent_t get_event(afp_handle* h)
{
    if (!h->ev_pool.empty()) { hipEvent_t e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void resolve_timings(afp_handle* h)
{
    for (auto& ep : h->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            h->t_ms[ep.slot] += ms;
            h->t_n[ep.slot] += 1;
        }
        h->ev_pool.push_back(ep.a);
        h->ev_pool.push_back(ep.b);
    }
    h->pending.clear();
}

// out[0] = HIP_VERSION the library was COMPILED against (hipcc of the build), out[1] = hipRuntimeGetVersion() of the runtime the
// process actually bound (PyTorch wheels bundle their own libamdhip64 under the same SONAME: audfprint_amd/_lib.py),
// out[2] = hipDriverGetVersion(), out[3] = devices visible.  Makes a HIP call: the runtime is initialised afterwards.
extern "C" int afp_runtime_info(int32_t* out)
{
    if (!out) return AFP_ERR_ARG;
    int rt = 0, drv = 0, n = 0;
    out[0] = (int32_t)HIP_VERSION;
    HIPCHK(hipRuntimeGetVersion(&rt));
    if (hipDriverGetVersion(&drv) != hipSuccess) drv = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    out[1] = rt; out[2] = drv; out[3] = n;
    return AFP_OK;
}

// Page-locked host memory for callers that have no allocator of their own for it (a host without torch): PCM handed to
// afp_extract_host* from such a buffer is uploaded asynchronously by the copy engine (Extractor.submit), pageable memory
// goes through the runtime's staging copies.
extern "C" int afp_pinned_alloc(int device, int64_t bytes, void** out)
{
    if (!out || bytes <= 0) return AFP_ERR_ARG;
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return AFP_OK;
}
extern "C" int afp_pinned_free(void* p)
{
    if (p) HIPCHK(hipHostFree(p));
    return AFP_OK;
}

HostPool* host_pool()
{
    static std::mutex mu;
    static HostPool* pool = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (pool && pool->pid == getpid()) return pool;
    HostPool* np = new HostPool();          // (a pool inherited through fork is abandoned, not destroyed: its threads are gone)
    np->pid = getpid();
    int want;
    const char* e = getenv("AFP_DL_THREADS");
    if (e) want = atoi(e);
    else {
        cpu_set_t set;
        CPU_ZERO(&set);
        const int nc = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        want = std::min(8, std::max(1, nc));
    }
    want = std::max(1, std::min(want, 64));
    for (int w = 1; w < want; w++) {
        try { np->th.emplace_back([np, w]() { np->worker(w); }); }
        catch (...) { break; }
    }
    for (auto& t : np->th) t.detach();      // they sleep on the condition variable until the process ends
    np->W = 1 + (int)np->th.size();
    pool = np;
    return pool;
}
extern "C" int afp_host_threads(void) { return host_pool()->W; }

// Populate the pages of a (large, freshly allocated) host array in the BACKGROUND: a HashTable's table is 420 MB of
// np.zeros -- untouched zero pages -- and the first write to each page costs a fault plus the kernel's zero fill; left to the
// table download at the end of a job that is 4-5 ms of its 6 (the scatter touches every page).  MADV_POPULATE_WRITE (Linux
// 5.14) faults the range in without changing its contents; a few detached threads do it while the device works on the
// job's first batches.  Best effort: an older kernel (EINVAL), a range that goes away meanwhile (ENOMEM) or a failed thread
// start just leave the pages to be faulted by their first real write, as before.  Returns the threads started.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
// The populating threads are PERSISTENT (round 6): starting a thread maps its stack, which needs the address space's lock
// for writing -- and MADV_POPULATE_WRITE holds it for reading for as long as it runs, so with threads created per call the
// SECOND std::thread already waited for the first one's populate: afp_host_prefault took 4.5-9.9 ms to return
// (tools/table_create_cost.py), all of it inside TableBuilder's creation.  The workers are started once per process (a
// forked child starts its own) and sleep on a condition variable; a call only queues ranges.
struct PrefaultPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::pair<uintptr_t, uintptr_t>> q;
    pid_t pid = 0;
    int n = 0;
    void worker()
    {
        for (;;) {
            std::pair<uintptr_t, uintptr_t> r;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                r = q.back();
                q.pop_back();
            }
            for (uintptr_t a = r.first; a < r.second; a += (uintptr_t)1 << 20)          // in 1 MB steps: a vanished range stops the loop early
                if (madvise((void*)a, (size_t)std::min<uintptr_t>((uintptr_t)1 << 20, r.second - a), MADV_POPULATE_WRITE) != 0) break;
        }
    }
};
static PrefaultPool* prefault_pool()
{
    static std::mutex mu;
    static PrefaultPool* pool = nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (pool && pool->pid == getpid()) return pool;
    PrefaultPool* np = new PrefaultPool();      // (a pool inherited through fork is abandoned: its threads are gone)
    np->pid = getpid();
    const char* e = getenv("AFP_PREFAULT_THREADS");
    int want = e ? atoi(e) : host_pool()->W;
    want = want < 1 ? 1 : want > 16 ? 16 : want;
    for (int t = 0; t < want; t++) {
        try { std::thread([np]() { np->worker(); }).detach(); np->n++; }
        catch (...) { break; }
    }
    pool = np;
    return pool;
}
extern "C" int afp_host_prefault(void* p, int64_t bytes)
{
    if (!p || bytes <= 0) return 0;
    static const bool off = getenv("AFP_NO_PREFAULT") != nullptr;
    if (off) return 0;
    const uintptr_t a0 = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)p + (uintptr_t)bytes) & ~(uintptr_t)4095;
    if (a1 <= a0) return 0;
    // The caller asks for this when it knows the host will sit idle meanwhile (a pipelined job waiting for its first
    // batches): while the threads populate, calls of the process that change its address space (allocations, thread starts)
    // wait -- a table store issued right behind the TableBuilder's creation took 3.7 ms instead of 1.3 with eight threads
    // (5 ms of populating) and 17 ms with two (20 ms of it): bench.py table_build, r05.  So: as many threads as the download
    // pool has (AFP_PREFAULT_THREADS), a short window, and TableBuilder does not start it unless told to (prefault=True).
    PrefaultPool* P = prefault_pool();
    if (P->n <= 0) return 0;
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(P->n, (int64_t)(a1 - a0) >> 25));
    const uintptr_t per = (((a1 - a0) / nth) + 4095) & ~(uintptr_t)4095;
    int queued = 0;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        for (int t = 0; t < nth; t++) {
            const uintptr_t lo = a0 + per * t, hi = std::min<uintptr_t>(a1, lo + per);
            if (hi <= lo) break;
            P->q.emplace_back(lo, hi);
            queued++;
        }
    }
    P->cv.notify_all();
    return queued;
}

int dl_ring(afp_handle* h)
{
    if (!h->h_dl) HIPCHK(hipHostMalloc(&h->h_dl, (size_t)(DL_R * DL_CH), hipHostMallocDefault));
    for (int i = 0; i < DL_R; i++) if (!h->dl_ev[i]) HIPCHK(hipEventCreateWithFlags(&h->dl_ev[i], hipEventDisableTiming));
    return AFP_OK;
}

// Device -> PAGEABLE host memory, large: the copy engine fills the ring and the pool's threads move each chunk on into the
// destination -- plain memcpy, whose page faults on a freshly allocated numpy array then run in parallel too.  The runtime's
// own pageable path does the same with one thread: ~17 GB/s.
int download_pageable(afp_handle* h, char* dst, const char* src, int64_t bytes, hipStream_t st)
{
    if (host_pool()->W <= 1 || bytes < 4 * DL_CH) {
        HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, st));
        return AFP_OK;
    }
    return ring_download(h, src, bytes, st, [&](int64_t k, int64_t n, const char* chunk, int w, int W) {
        const int64_t per = ((n + W - 1) / W + 4095) & ~(int64_t)4095;
        const int64_t a = std::min<int64_t>(n, w * per), b = std::min<int64_t>(n, a + per);
        if (b > a) memcpy(dst + k * DL_CH + a, chunk + a, (size_t)(b - a));
    });
}
