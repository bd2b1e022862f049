// libpyslam_hipvol.so — volume lifetime, block hash/pool allocation, staging, measurement hooks.
#include <cstdarg>
#include <cstdlib>

#include <immintrin.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "hv_common.h"

static thread_local std::string g_last_error;

void hv_set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// General 4x4 inverse by cofactors (adjugate / determinant), row-major doubles — the library's
// definition of Eigen's `extrinsic.inverse()` in Open3D's CreatePointCloudFromFloatDepthImage.
void hv_invert4x4(const double *m, double *out) {
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
    const double inv_det = 1.0 / det;
    for (int i = 0; i < 16; ++i) out[i] = a[i] * inv_det;
}

int hv_ensure_buffer(hv_volume *v, void **buf, size_t *cur, size_t want) {
    if (*cur >= want && *buf != nullptr) return HV_OK;
    if (*buf) {
        HV_HIP(hipStreamSynchronize(v->stream));
        HV_HIP(hipFree(*buf));
        *buf = nullptr;
        *cur = 0;
    }
    size_t bytes = want < 4096 ? 4096 : want + want / 4;
    HV_HIP(hipMalloc(buf, bytes));
    *cur = bytes;
    return HV_OK;
}

int hv_read_counters(hv_volume *v) {
    HV_HIP(hipMemcpyAsync(v->h_counters, v->table.counters, sizeof(int32_t) * HV_CNT_COUNT,
                          hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

// HV_HOST inputs are copied into a device staging buffer on the volume's stream; HV_DEVICE inputs
// are used in place.
int hv_stage_in(hv_volume *v, const void *src, size_t bytes, int32_t loc, int which, const void **dev) {
    if (src == nullptr || bytes == 0) {
        *dev = nullptr;
        return HV_OK;
    }
    if (loc == HV_DEVICE) {
        *dev = src;
        return HV_OK;
    }
    void **buf = which == 0 ? &v->stage_a : &v->stage_b;
    size_t *cur = which == 0 ? &v->stage_a_bytes : &v->stage_b_bytes;
    int rc = hv_ensure_buffer(v, buf, cur, bytes);
    if (rc != HV_OK) return rc;
    rc = hv_h2d(v, *buf, src, bytes);
    if (rc != HV_OK) return rc;
    *dev = *buf;
    return HV_OK;
}

// Host -> device copy of a caller's array on the volume's stream.  The C ABI borrows HV_HOST arrays "for the duration of the call":
// a pageable source is read by the runtime before hipMemcpyAsync returns, but a PAGE-LOCKED source (the front's shared-memory ring
// after hv_host_register, a torch pinned tensor, hipHostMalloc memory) is read by the DMA engine later, when the stream gets there -
// the caller may have refilled the memory by then (a ring slot released while its copy was still queued).  So the
// host waits for the copy when the source is page-locked; the kernels behind it stay asynchronous.
static bool hv_host_is_registered(const void *p, size_t bytes);
static bool hv_host_is_pinned(const void *p, size_t bytes) {
    if (hv_host_is_registered(p, bytes)) return true;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc / numpy memory: "invalid value", not an error of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

int hv_h2d_lazy(hv_volume *v, void *dst, const void *src, size_t bytes, bool *pending) {
    if (bytes == 0) return HV_OK;
    HV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, v->stream));
    if (hv_host_is_pinned(src, bytes)) *pending = true;
    return HV_OK;
}

// (an event right behind the copies, not the whole stream: a caller whose keyframes sit in page-locked memory would otherwise wait
// for every kernel queued before this call as well - ADVICE r05)
int hv_h2d_fence(hv_volume *v, bool pending) {
    if (!pending) return HV_OK;
    if (v->ev_h2d == nullptr) HV_HIP(hipEventCreateWithFlags(&v->ev_h2d, hipEventDisableTiming));
    HV_HIP(hipEventRecord(v->ev_h2d, v->stream));
    HV_HIP(hipEventSynchronize(v->ev_h2d));
    return HV_OK;
}

int hv_h2d(hv_volume *v, void *dst, const void *src, size_t bytes) {
    bool pending = false;
    const int rc = hv_h2d_lazy(v, dst, src, bytes, &pending);
    return rc != HV_OK ? rc : hv_h2d_fence(v, pending);
}

// ---- pipelined staging of host-resident frames ------------------------------------------------------------------------
// pySLAM hands every keyframe over as pageable numpy arrays (volumetric_integrator_base.py:101-137, _tsdf.py:215-223).  A
// pageable hipMemcpyAsync is staged by the runtime through its own bounce buffers on the calling thread and blocks the
// compute stream it is issued on.  Here instead: a few worker threads copy a sub-chunk of frames into one of two
// page-locked slots, the slot goes to the device with ONE asynchronous DMA on a copy stream, and while it is in flight
// the threads fill the other slot; the frames land in one of two device sets, so the DMA of batch k+1 runs while batch k is
// swept.  Measured in bench.py (`host_mode`).
namespace {
class HvCopyPool { // n persistent workers; run(fn) executes fn(i, n) on all of them and returns when every one is done
  public:
    explicit HvCopyPool(int n) : n_(n) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
        for (auto &t : th_) t.detach(); // process-lifetime pool (the library is never unloaded before exit)
    }
    int size() const { return n_; }
    void run(const std::function<void(int, int)> &fn) {
        // one job at a time: ctypes releases the GIL, so two volumes on two host threads may stage frames concurrently; a second
        // run() would overwrite job_ / pending_ / gen_ while the first caller waits and both would wake on pending_ == 0
        std::lock_guard<std::mutex> callers(run_m_);
        std::unique_lock<std::mutex> lk(m_);
        job_ = &fn;
        pending_ = n_;
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(int i) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)> *job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                job = job_;
            }
            (*job)(i, n_);
            {
                std::unique_lock<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_, run_m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *job_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
};

// Copy into a page-locked staging slot with non-temporal stores: the destination is read next by the DMA engine, not by a
// core, so write-allocating it (a plain memcpy of a 100 KB slice reads every destination line first) is a third of the memory
// traffic for nothing.  dst is 64-byte aligned by construction (slices are cut at multiples of 64 from page-aligned slots).
__attribute__((target("avx2"))) void hv_stream_copy_avx2(char *dst, const char *src, size_t n) {
    size_t i = 0;
    if (((uintptr_t)dst & 31) == 0) {
        for (; i + 128 <= n; i += 128) {
            const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i));
            const __m256i b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
            const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64));
            const __m256i d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
            _mm256_stream_si256((__m256i *)(dst + i), a);
            _mm256_stream_si256((__m256i *)(dst + i + 32), b);
            _mm256_stream_si256((__m256i *)(dst + i + 64), c);
            _mm256_stream_si256((__m256i *)(dst + i + 96), d);
        }
        _mm_sfence();
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
}

void hv_stream_copy(char *dst, const char *src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) hv_stream_copy_avx2(dst, src, n);
    else memcpy(dst, src, n);
}

HvCopyPool *hv_copy_pool() {
    static HvCopyPool *pool = [] {
        int n = 8; // measured on the 256-thread host of an MI355X box: 4 / 6 / 8 / 12 threads -> 0.63 / 0.64 / 0.72 / 0.71 of the H2D bound
        if (const char *e = getenv("HV_STAGE_THREADS")) n = atoi(e);
        n = std::min<int>(n, std::max(1u, std::thread::hardware_concurrency()));
        n = std::max(1, std::min(n, 32));
        return new HvCopyPool(n);
    }();
    return pool;
}
} // namespace

// ---- caller memory page-locked with hv_host_register (process-wide) ----
namespace {
struct HvHostRange {
    const char *lo, *hi;
};
std::mutex g_host_ranges_m;
std::vector<HvHostRange> g_host_ranges;

} // namespace
static bool hv_host_is_registered(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_ranges_m);
    for (const HvHostRange &r : g_host_ranges)
        if ((const char *)p >= r.lo && (const char *)p + bytes <= r.hi) return true;
    return false;
}

extern "C" int hv_host_register(void *ptr, int64_t bytes) {
    HV_REQUIRE(ptr != nullptr && bytes > 0, HV_ERR_INVALID, "hv_host_register: null pointer or empty range");
    HV_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lk(g_host_ranges_m);
    g_host_ranges.push_back({(const char *)ptr, (const char *)ptr + bytes});
    return HV_OK;
}

extern "C" int hv_host_unregister(void *ptr) {
    HV_REQUIRE(ptr != nullptr, HV_ERR_INVALID, "hv_host_unregister: null pointer");
    {
        std::lock_guard<std::mutex> lk(g_host_ranges_m);
        bool found = false;
        for (size_t i = 0; i < g_host_ranges.size(); ++i)
            if (g_host_ranges[i].lo == (const char *)ptr) {
                g_host_ranges.erase(g_host_ranges.begin() + (long)i);
                found = true;
                break;
            }
        HV_REQUIRE(found, HV_ERR_INVALID, "hv_host_unregister: the range was not registered with hv_host_register");
    }
    HV_HIP(hipHostUnregister(ptr));
    return HV_OK;
}

int hv_stage_frames(hv_volume *v, const void *const *depth_ptrs, const void *depth_base, size_t depth_frame_bytes,
                    const void *const *rgb_ptrs, const void *rgb_base, size_t rgb_frame_bytes, int n_frames,
                    const void **d_depth, const void **d_rgb, int *set_out) {
    if (v->hs_stream == nullptr) {
        HV_HIP(hipStreamCreateWithFlags(&v->hs_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HV_HIP(hipEventCreateWithFlags(&v->hs_pinned_done[i], hipEventDisableTiming));
            HV_HIP(hipEventCreateWithFlags(&v->hs_dev_ready[i], hipEventDisableTiming));
            HV_HIP(hipEventCreateWithFlags(&v->hs_dev_free[i], hipEventDisableTiming));
        }
    }
    const int set = v->hs_set;
    v->hs_set ^= 1;
    const size_t frame_bytes = depth_frame_bytes + rgb_frame_bytes;
    // device set (grown on demand; its previous readers are done once hs_dev_free has fired)
    const size_t want[2] = {depth_frame_bytes * (size_t)n_frames, rgb_frame_bytes * (size_t)n_frames};
    for (int a = 0; a < 2; ++a) {
        if (v->hs_dev_bytes[set][a] < want[a] || v->hs_dev[set][a] == nullptr) {
            if (v->hs_dev_free_valid[set]) HV_HIP(hipEventSynchronize(v->hs_dev_free[set]));
            HV_HIP(hipStreamSynchronize(v->hs_stream));
            if (v->hs_dev[set][a]) HV_HIP(hipFree(v->hs_dev[set][a]));
            v->hs_dev[set][a] = nullptr;
            v->hs_dev_bytes[set][a] = 0;
            const size_t bytes = std::max<size_t>(4096, want[a] + want[a] / 4);
            HV_HIP(hipMalloc(&v->hs_dev[set][a], bytes));
            v->hs_dev_bytes[set][a] = bytes;
        }
    }
    if (v->hs_dev_free_valid[set]) HV_HIP(hipStreamWaitEvent(v->hs_stream, v->hs_dev_free[set], 0));
    // Frames that already lie in page-locked memory (hv_host_register: the front's shared-memory ring) need no staging copy: the
    // DMA engine reads each of them in place.  The call returns when the copies have READ the caller's memory (the contract of
    // hv_tsdf_integrate_frames) - the host waits for the copy stream, not for the sweep that runs beside it.
    {
        bool pinned = depth_ptrs != nullptr && rgb_ptrs != nullptr;
        for (int f = 0; pinned && f < n_frames; ++f)
            pinned = hv_host_is_registered(depth_ptrs[f], depth_frame_bytes) && hv_host_is_registered(rgb_ptrs[f], rgb_frame_bytes);
        if (pinned) {
            for (int f = 0; f < n_frames; ++f) {
                HV_HIP(hipMemcpyAsync((char *)v->hs_dev[set][0] + depth_frame_bytes * (size_t)f, depth_ptrs[f], depth_frame_bytes,
                                      hipMemcpyHostToDevice, v->hs_stream));
                HV_HIP(hipMemcpyAsync((char *)v->hs_dev[set][1] + rgb_frame_bytes * (size_t)f, rgb_ptrs[f], rgb_frame_bytes,
                                      hipMemcpyHostToDevice, v->hs_stream));
            }
            HV_HIP(hipEventRecord(v->hs_dev_ready[set], v->hs_stream));
            HV_HIP(hipStreamSynchronize(v->hs_stream));
            *d_depth = v->hs_dev[set][0];
            *d_rgb = v->hs_dev[set][1];
            *set_out = set;
            return HV_OK;
        }
    }
    // sub-chunks of about 16 MB: long enough for the DMA engine to reach its rate, short enough that the first one is on its
    // way early
    size_t sub_target = 16u << 20; // (4 / 8 / 16 MB measured: 0.53 / 0.72 / 0.76 of the H2D bound)
    if (const char *e = getenv("HV_STAGE_CHUNK_MB")) sub_target = (size_t)std::max(1, atoi(e)) << 20; // (tests: small sub-chunks reuse slots and sets)
    const int per_sub = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, sub_target / std::max<size_t>(1, frame_bytes)));
    HvCopyPool *pool = hv_copy_pool();
    for (int c0 = 0; c0 < n_frames; c0 += per_sub) {
        const int nc = std::min(per_sub, n_frames - c0);
        const int slot = v->hs_slot;
        v->hs_slot ^= 1;
        const size_t need = frame_bytes * (size_t)nc;
        if (v->hs_pinned_bytes[slot] < need) {
            if (v->hs_pinned[slot]) {
                HV_HIP(hipEventSynchronize(v->hs_pinned_done[slot]));
                HV_HIP(hipHostFree(v->hs_pinned[slot]));
                v->hs_pinned[slot] = nullptr;
                v->hs_pinned_bytes[slot] = 0;
            }
            const size_t bytes = std::max(need, frame_bytes * (size_t)per_sub);
            HV_HIP(hipHostMalloc(&v->hs_pinned[slot], bytes));
            v->hs_pinned_bytes[slot] = bytes;
        } else {
            HV_HIP(hipEventSynchronize(v->hs_pinned_done[slot])); // (an event that was never recorded reports complete)
        }
        char *pd = (char *)v->hs_pinned[slot];
        char *pc = pd + depth_frame_bytes * (size_t)nc;
        // every worker copies its slice of every frame of the sub-chunk (64-byte aligned cuts)
        pool->run([&](int i, int n) {
            auto slice = [&](size_t len, size_t &lo, size_t &hi) {
                const size_t step = ((len + (size_t)n - 1) / (size_t)n + 63) & ~(size_t)63;
                lo = std::min(len, step * (size_t)i);
                hi = std::min(len, lo + step);
            };
            for (int f = 0; f < nc; ++f) {
                size_t lo, hi;
                const char *sd = depth_ptrs ? (const char *)depth_ptrs[c0 + f] : (const char *)depth_base + depth_frame_bytes * (size_t)(c0 + f);
                slice(depth_frame_bytes, lo, hi);
                if (hi > lo) hv_stream_copy(pd + depth_frame_bytes * (size_t)f + lo, sd + lo, hi - lo);
                const char *sc = rgb_ptrs ? (const char *)rgb_ptrs[c0 + f] : (const char *)rgb_base + rgb_frame_bytes * (size_t)(c0 + f);
                slice(rgb_frame_bytes, lo, hi);
                if (hi > lo) hv_stream_copy(pc + rgb_frame_bytes * (size_t)f + lo, sc + lo, hi - lo);
            }
        });
        HV_HIP(hipMemcpyAsync((char *)v->hs_dev[set][0] + depth_frame_bytes * (size_t)c0, pd, depth_frame_bytes * (size_t)nc,
                              hipMemcpyHostToDevice, v->hs_stream));
        HV_HIP(hipMemcpyAsync((char *)v->hs_dev[set][1] + rgb_frame_bytes * (size_t)c0, pc, rgb_frame_bytes * (size_t)nc,
                              hipMemcpyHostToDevice, v->hs_stream));
        HV_HIP(hipEventRecord(v->hs_pinned_done[slot], v->hs_stream));
    }
    HV_HIP(hipEventRecord(v->hs_dev_ready[set], v->hs_stream));
    *d_depth = v->hs_dev[set][0];
    *d_rgb = v->hs_dev[set][1];
    *set_out = set;
    return HV_OK;
}

int hv_stage_frames_consumed(hv_volume *v, int set, hipStream_t consumer) {
    HV_HIP(hipEventRecord(v->hs_dev_free[set], consumer));
    v->hs_dev_free_valid[set] = true;
    return HV_OK;
}

void hv_profile_begin(hv_volume *v) {
    if (!v->profiling) return;
    if (v->events_used == v->events.size()) {
        HvEventPair p;
        if (hipEventCreate(&p.start) != hipSuccess || hipEventCreate(&p.stop) != hipSuccess) return;
        v->events.push_back(p);
    }
    (void)hipEventRecord(v->events[v->events_used].start, v->stream);
}

void hv_profile_end(hv_volume *v, int64_t units) {
    if (!v->profiling || v->events_used >= v->events.size()) return;
    (void)hipEventRecord(v->events[v->events_used].stop, v->stream);
    v->events_used++;
    v->prof_units += units;
}

__global__ void k_publish_status(HvTable table, HvStatus *status, int32_t seq) { hv_publish_status(table, status, seq); }

int32_t hv_next_status_seq(hv_volume *v) {
    v->status_exact = false;
    return ++v->status_seq_issued;
}

void hv_launch_publish_status(hv_volume *v) {
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(1), 0, v->stream, v->table, v->d_status, hv_next_status_seq(v));
}

static int hv_rebuild(hv_volume *v, int64_t new_max_blocks, int64_t keep);
static int hv_rollback_claims(hv_volume *v, int64_t keep);
static uint64_t next_pow2(uint64_t x);

static bool hv_auto_grow() { return !(getenv("HV_AUTO_GROW") && atoi(getenv("HV_AUTO_GROW")) == 0); }

// Double the pool while it fits (60 % of the free HBM, 32-bit sort keys of the grid modes).
static int hv_grow(hv_volume *v, int64_t at_least) {
    int64_t want = v->cfg.max_blocks;
    while (want < at_least) want *= 2;
    if (want == v->cfg.max_blocks) want *= 2;
    size_t free_b = 0, total_b = 0;
    uint64_t cap = 1;
    while (cap < (uint64_t)want * 4) cap <<= 1;
    const bool fits_keys = v->cfg.mode == HV_MODE_TSDF || (cap << v->local_bits) < (1ull << 32);
    if (!hv_auto_grow() || !fits_keys || want >= (1ll << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
        (double)want * (double)v->bytes_per_block >= 0.6 * (double)free_b)
        return HV_ERR_CAPACITY;
    return hv_reserve_blocks(v, want);
}

int hv_capacity_gate(hv_volume *v, bool *checked) {
    // consume the newest published state (seq is written last, behind a system-scope fence)
    const volatile HvStatus *st = v->h_status;
    const int32_t seq = st->seq;
    int64_t overflow = 0;
    if (seq != v->status_seq_seen && seq > 0) {
        const int64_t blocks = st->blocks;
        overflow = st->overflow;
        if (st->seq == seq) { // not torn by a newer publication (then the next call sees it)
            const int64_t grown = std::max<int64_t>(0, blocks - v->known_blocks);
            const int64_t calls = std::max<int64_t>(1, seq - v->status_seq_seen);
            v->max_new_per_call = std::max<int64_t>(v->max_new_per_call, grown / calls + (grown % calls != 0));
            v->avg_new_per_call = 0.75 * v->avg_new_per_call + 0.25 * (double)grown / (double)calls;
            v->known_blocks = std::max<int64_t>(v->known_blocks, std::min<int64_t>(blocks, v->cfg.max_blocks));
            v->status_seq_seen = seq;
        }
    }
    // an earlier association (hv_assoc_vote / _decide in the device flow, which never fetches the map) dropped votes or voxels: say so
    // ONCE, at the first call that can - the volume itself is consistent, the caller decides whether to go on
    if (overflow != 0) v->overflow_latched = true; // stays set until hv_reserve_blocks / hv_reset repair the pool (a rebuild clears it)
    if (const int32_t af = st->assoc_flags) {
        v->h_status->assoc_flags = 0;
        HV_REQUIRE(false, HV_ERR_CAPACITY,
                   "an earlier assign_object_ids_to_instance_ids overflowed (%s%s%s): some voxels did not receive their object id; "
                   "the volume is otherwise unchanged",
                   (af & 2) ? "vote table full " : "", (af & 4) ? "pending list full " : "", (af & 1) ? "more than 4096 (instance, object) pairs" : "");
    }
    HV_REQUIRE(!v->overflow_latched, HV_ERR_CAPACITY,
               "block pool exhausted during an earlier integrate call (max_blocks=%lld): units that did not fit were not fused; "
               "nothing more is fused until hv_reserve_blocks or hv_reset",
               (long long)v->cfg.max_blocks);
    if (v->known_blocks * 2 > v->cfg.max_blocks) (void)hv_grow(v, v->cfg.max_blocks * 2); // may fail: the checked mode below covers it
    // Unchecked (fully asynchronous) only when the free part of the pool covers the largest growth a call ever caused, 4 x the
    // recent growth per call for every call still in flight, and a margin; otherwise the call verifies its claims.  A burst
    // beyond that inside the lag window cannot be lost silently either: the next gate sees the overflow and refuses.
    const int64_t in_flight = v->status_seq_issued - v->status_seq_seen;
    const int64_t need = v->max_new_per_call + (int64_t)(4.0 * v->avg_new_per_call * (double)(in_flight + 1)) + 1024;
    // max_new_per_call == 0: nothing is known yet about what a call allocates (first call after creation / reset)
    *checked = v->max_new_per_call == 0 || v->cfg.max_blocks - v->known_blocks < need;
    return HV_OK;
}

int hv_claims_fit(hv_volume *v) {
    int rc = hv_read_counters(v); // synchronises the stream
    if (rc != HV_OK) return rc;
    const int64_t blocks = v->h_counters[HV_CNT_BLOCKS];
    if (v->h_counters[HV_CNT_OVERFLOW] == 0) {
        const int64_t grown = std::max<int64_t>(0, blocks - v->known_blocks);
        v->max_new_per_call = std::max<int64_t>(v->max_new_per_call, std::max<int64_t>(grown, 1));
        v->avg_new_per_call = 0.75 * v->avg_new_per_call + 0.25 * (double)grown;
        v->known_blocks = blocks;
        return HV_OK;
    }
    // some claims did not fit: the pool index counter ran past max_blocks (the excess = the blocks that are missing)
    rc = hv_grow(v, blocks + blocks / 4);
    if (rc != HV_OK) {
        // roll the claim pass back: the blocks it did get are released again (nothing was written to them), the failed keys
        // leave the table - the volume is exactly what it was before the call
        // "before" = the larger of the last published occupancy and what the host knows exactly: units that arrived without a
        // publishing kernel (hv_tsdf_import_numerators on a gather root) are only in known_blocks - they must survive the rollback
        const int64_t published = v->h_status->seq == v->status_seq_issued && v->status_seq_issued > 0 ? v->h_status->blocks : 0;
        const int64_t before = std::min<int64_t>(std::max<int64_t>(published, v->known_blocks), v->cfg.max_blocks);
        const int64_t max_blocks = v->cfg.max_blocks;
        // in place: only the (small) table arrays are re-made - a second full-size pool is exactly what a box that could not
        // grow does not have
        const int rb = hv_rollback_claims(v, before);
        if (rb != HV_OK) {
            v->overflow_latched = true; // the hash may hold keys without a block: every integrate call fails until hv_reset / hv_reserve_blocks
            const std::string why = hv_last_error();
            hv_set_error("block pool exhausted: %lld blocks needed, max_blocks=%lld, the pool cannot grow AND the claim pass could not be "
                         "rolled back (%s): the frame was NOT fused, but the hash holds keys without a block - hv_reset or "
                         "hv_reserve_blocks before the volume is used again",
                         (long long)blocks, (long long)max_blocks, why.c_str());
            return HV_ERR_CAPACITY;
        }
        hv_set_error("block pool exhausted: %lld blocks needed, max_blocks=%lld and the pool cannot grow (HV_AUTO_GROW=0, no free HBM, "
                     "or the 32-bit sort keys of the grid modes); the frame was NOT fused, the volume is unchanged",
                     (long long)blocks, (long long)max_blocks);
        return HV_ERR_CAPACITY;
    }
    return HV_RETRY_CLAIM;
}

static uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

extern "C" {

const char *hv_last_error(void) { return g_last_error.c_str(); }

int hv_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        hv_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return HV_ERR_DEVICE;
    }
    return n;
}

void hv_default_config(int32_t mode, hv_config *cfg) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->mode = mode;
    cfg->device = 0;
    if (mode == HV_MODE_TSDF) {
        cfg->voxel_size = 0.015; // kVolumetricIntegrationVoxelLength, config_parameters.py:311
        cfg->sdf_trunc = 0.04;   // kVolumetricIntegrationTSdfTrunc, config_parameters.py:349
        cfg->block_size = 16;    // Open3D volume_unit_resolution
        cfg->depth_sampling_stride = 4;
        cfg->max_blocks = 1 << 16; // 65536 units * 80 KiB = 5 GiB of the 288 GB HBM
    } else {
        cfg->voxel_size = 0.015;
        cfg->sdf_trunc = 0.0;
        cfg->block_size = 8; // kVolumetricIntegrationBlockSize, config_parameters.py:313
        cfg->depth_sampling_stride = 1;
        cfg->max_blocks = 1 << 19; // 524288 blocks * 16 KiB = 8 GiB
    }
    cfg->max_points = 2 * 1024 * 1024;
}

int hv_create(const hv_config *cfg, hv_volume **out) {
    HV_REQUIRE(cfg != nullptr && out != nullptr, HV_ERR_INVALID, "hv_create: null argument");
    HV_REQUIRE(cfg->mode == HV_MODE_VOXEL_GRID || hv_mode_is_semantic(cfg->mode) || cfg->mode == HV_MODE_TSDF, HV_ERR_INVALID,
               "hv_create: unsupported mode %d", cfg->mode);
    HV_REQUIRE(cfg->voxel_size > 0.0, HV_ERR_INVALID, "hv_create: voxel_size must be > 0");
    HV_REQUIRE(cfg->max_blocks > 0 && cfg->max_blocks < (1ll << 30), HV_ERR_INVALID,
               "hv_create: max_blocks out of range");
    HV_REQUIRE(cfg->max_points > 0 && cfg->max_points < (1ll << 31), HV_ERR_INVALID,
               "hv_create: max_points out of range");
    if (cfg->mode == HV_MODE_TSDF) {
        HV_REQUIRE(cfg->block_size == 16, HV_ERR_INVALID,
                   "hv_create: TSDF mode supports volume_unit_resolution 16 only (Open3D default)");
        HV_REQUIRE(cfg->sdf_trunc > 0.0, HV_ERR_INVALID, "hv_create: sdf_trunc must be > 0");
        HV_REQUIRE(cfg->depth_sampling_stride >= 1, HV_ERR_INVALID, "hv_create: bad sampling stride");
    } else {
        HV_REQUIRE(cfg->block_size >= 1 && cfg->block_size <= 16, HV_ERR_INVALID,
                   "hv_create: block_size must be in [1,16]");
    }
    int ndev = 0;
    HV_HIP(hipGetDeviceCount(&ndev));
    HV_REQUIRE(ndev > 0 && cfg->device >= 0 && cfg->device < ndev, HV_ERR_DEVICE,
               "hv_create: HIP device %d not available (%d devices)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HV_HIP(hipGetDeviceProperties(&prop, cfg->device));
    HV_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, HV_ERR_DEVICE,
               "hv_create: device %d is %s; this library is built for gfx950 only", cfg->device,
               prop.gcnArchName);
    HV_HIP(hipSetDevice(cfg->device));

    hv_volume *v = new hv_volume();
    v->cfg = *cfg;
    v->device = cfg->device;
    int rc = HV_OK;
    auto fail = [&](int code) {
        hv_destroy(v);
        return code;
    };
#define HV_TRY(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            hv_set_error("%s failed: %s", #call, hipGetErrorString(e_));                           \
            return fail(HV_ERR_DEVICE);                                                            \
        }                                                                                          \
    } while (0)

    HV_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
    v->own_stream = true;

    const int64_t nvox = (int64_t)cfg->block_size * cfg->block_size * cfg->block_size;
    v->bytes_per_block = cfg->mode == HV_MODE_TSDF ? nvox * 4 * HV_TSDF_PLANES
                         : hv_mode_has_label_maps(cfg->mode) ? nvox * 128 /* HvProbVoxel, HvProb2Voxel */
                         : hv_mode_is_semantic(cfg->mode)    ? nvox * 64  /* HvSemVoxel, HvSem2Voxel */
                                                             : nvox * (int64_t)sizeof(HvVoxel);
    if (hv_mode_has_label_maps(cfg->mode)) v->sem_depth_threshold = 5.0f; // voxel_data_semantic.h:251-254, voxel_data_semantic2.h:258-261
    v->local_bits = 0;
    while ((1ll << v->local_bits) < nvox) v->local_bits++;

    v->table_capacity = next_pow2((uint64_t)cfg->max_blocks * 4);
    if (v->table_capacity < 1024) v->table_capacity = 1024;
    if (cfg->mode != HV_MODE_TSDF && (v->table_capacity << v->local_bits) >= (1ull << 32)) {
        hv_set_error("hv_create: max_blocks too large for 32-bit (slot, voxel) sort keys");
        return fail(HV_ERR_INVALID);
    }
    v->table.mask = (uint32_t)(v->table_capacity - 1);
    v->table.max_blocks = (int32_t)cfg->max_blocks;
    HV_TRY(hipMalloc(&v->table.keys, sizeof(uint64_t) * v->table_capacity));
    HV_TRY(hipMalloc(&v->table.vals, sizeof(int32_t) * v->table_capacity));
    HV_TRY(hipMalloc(&v->table.block_keys, sizeof(uint64_t) * cfg->max_blocks));
    HV_TRY(hipMalloc(&v->table.counters, sizeof(int32_t) * HV_CNT_COUNT));
    HV_TRY(hipHostMalloc((void **)&v->h_counters, sizeof(int32_t) * HV_CNT_COUNT));
    memset(v->h_counters, 0, sizeof(int32_t) * HV_CNT_COUNT);
    HV_TRY(hipHostMalloc((void **)&v->h_status, sizeof(HvStatus), hipHostMallocMapped));
    memset(v->h_status, 0, sizeof(HvStatus));
    HV_TRY(hipHostGetDevicePointer((void **)&v->d_status, v->h_status, 0));
    HV_TRY(hipMalloc(&v->pool, (size_t)cfg->max_blocks * v->bytes_per_block));

    if (cfg->mode == HV_MODE_TSDF) {
        HV_TRY(hipMalloc(&v->touched_stamp, sizeof(int32_t) * v->table_capacity));
        HV_TRY(hipMalloc(&v->touched_list, sizeof(int32_t) * HV_TSDF_SETS * cfg->max_blocks));
        HV_TRY(hipMalloc(&v->touched_mask, sizeof(uint64_t) * HV_TSDF_SETS * v->table_capacity));
        HV_TRY(hipMalloc(&v->frame_px, 8 * (size_t)cfg->max_points));
        if (const char *tb = getenv("HV_TSDF_TOUCH_BOX_BITS")) v->touch_box_bits = std::min(std::max(atoi(tb), 0), 2048);
    } else {
        HV_TRY(hipMalloc(&v->sort_keys_in, sizeof(uint32_t) * cfg->max_points));
        HV_TRY(hipMalloc(&v->sort_keys_out, sizeof(uint32_t) * cfg->max_points));
        HV_TRY(hipMalloc(&v->sort_vals_in, sizeof(uint32_t) * cfg->max_points));
        HV_TRY(hipMalloc(&v->sort_vals_out, sizeof(uint32_t) * cfg->max_points));
        HV_TRY(hipMalloc(&v->scratch_points, sizeof(float) * 3 * cfg->max_points));
        HV_TRY(hipMalloc(&v->scratch_colors, sizeof(float) * 3 * cfg->max_points));
        if (cfg->mode != HV_MODE_VOXEL_GRID) {
            const size_t occ_words = ((size_t)cfg->max_blocks * nvox + 63) / 64;
            HV_TRY(hipMalloc((void **)&v->occ, sizeof(unsigned long long) * occ_words));
            HV_TRY(hipMemsetAsync(v->occ, 0, sizeof(unsigned long long) * occ_words, v->stream));
        }
        if (hv_mode_has_label_maps(cfg->mode)) {
            // overflow nodes of the per-voxel label maps (hv_semantic.h): 4 per block of the pool - measured need under 5 % uniform
            // label noise: about one node per 100 occupied voxels.  HV_PROB_NODE_CAP overrides (tests of the exhausted pool).
            int64_t cap = std::max<int64_t>(1 << 16, (int64_t)cfg->max_blocks * 4);
            if (const char *e = getenv("HV_PROB_NODE_CAP")) cap = std::max<int64_t>(atoll(e), 1);
            cap = std::min<int64_t>(cap, INT32_MAX / 2);
            HV_TRY(hipMalloc(&v->table.prob_nodes, (size_t)cap * 128));
            v->table.prob_node_cap = (int32_t)cap;
        }
    }
#undef HV_TRY
    // first reset zeroes the whole pool (later resets only the used prefix)
    HV_HIP(hipMemsetAsync(v->pool, 0, (size_t)cfg->max_blocks * v->bytes_per_block, v->stream));
    rc = hv_reset(v);
    if (rc != HV_OK) return fail(rc);
    *out = v;
    return HV_OK;
}

void hv_destroy(hv_volume *v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream_aux) (void)hipStreamSynchronize(v->stream_aux);
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->hs_stream) (void)hipStreamSynchronize(v->hs_stream);
    for (int i = 0; i < 2; ++i) {
        if (v->hs_pinned[i]) (void)hipHostFree(v->hs_pinned[i]);
        for (int a = 0; a < 2; ++a)
            if (v->hs_dev[i][a]) (void)hipFree(v->hs_dev[i][a]);
        if (v->hs_pinned_done[i]) (void)hipEventDestroy(v->hs_pinned_done[i]);
        if (v->hs_dev_ready[i]) (void)hipEventDestroy(v->hs_dev_ready[i]);
        if (v->hs_dev_free[i]) (void)hipEventDestroy(v->hs_dev_free[i]);
    }
    if (v->hs_stream) (void)hipStreamDestroy(v->hs_stream);
    void *bufs[] = {v->table.keys, v->table.vals, v->table.block_keys, v->table.counters, v->pool,
                    v->touched_stamp, v->touched_list, v->touched_mask, v->frame_px,
                    v->stage_a, v->stage_b, v->sort_keys_in, v->sort_keys_out, v->sort_vals_in,
                    v->sort_vals_out, v->sort_tmp, v->scratch_points, v->scratch_colors, v->out_a,
                    v->out_b, v->out_c, v->rect_map_x, v->rect_map_y, v->rect_buf, v->batch_buf, v->assoc_buf, v->mult_table, v->bins.cnt, v->bins.inl, v->bins.touched, v->bins.len, v->bins.pg_keys, v->bins.pg_data, v->bin_rec, v->batch_buf2, v->unit_masks, v->occ, v->semb_tasks, v->table.prob_nodes, v->shadow_ring, v->mc_cache, v->pc_cache, v->kf_buf, v->halo_plan};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (v->h_counters) (void)hipHostFree(v->h_counters);
    if (v->h_status) (void)hipHostFree(v->h_status);
    hv_segments_cache_free(v->segments_cache);
    for (int i = 0; i < 4; ++i) {
        if (v->pinned_params[i]) (void)hipHostFree(v->pinned_params[i]);
        if (v->params_ev[i]) (void)hipEventDestroy(v->params_ev[i]);
    }
    for (auto &p : v->events) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    if (v->ev_h2d) (void)hipEventDestroy(v->ev_h2d);
    if (v->ev_prep) (void)hipEventDestroy(v->ev_prep);
    if (v->ev_presweep) (void)hipEventDestroy(v->ev_presweep);
    if (v->stream_aux) (void)hipStreamDestroy(v->stream_aux);
    if (v->stream && v->own_stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

int hv_reset(hv_volume *v) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_reset: null volume");
    HV_HIP(hipSetDevice(v->device));
    // zero only the used pool prefix (pool was fully zeroed at creation)
    int64_t used = 0;
    if (v->frame_counter != 0 || v->h_counters[HV_CNT_BLOCKS] != 0) {
        int rc = hv_read_counters(v);
        if (rc != HV_OK) return rc;
        used = v->h_counters[HV_CNT_BLOCKS];
        if (used > v->cfg.max_blocks) used = v->cfg.max_blocks;
    }
    if (used > 0) HV_HIP(hipMemsetAsync(v->pool, 0, (size_t)used * v->bytes_per_block, v->stream));
    if (used > 0 && v->occ) {
        const int64_t nvox = (int64_t)v->cfg.block_size * v->cfg.block_size * v->cfg.block_size;
        HV_HIP(hipMemsetAsync(v->occ, 0, sizeof(unsigned long long) * (size_t)((used * nvox + 63) / 64), v->stream));
    }
    HV_HIP(hipMemsetAsync(v->table.keys, 0xFF, sizeof(uint64_t) * v->table_capacity, v->stream));
    HV_HIP(hipMemsetAsync(v->table.vals, 0xFF, sizeof(int32_t) * v->table_capacity, v->stream));
    HV_HIP(hipMemsetAsync(v->table.counters, 0, sizeof(int32_t) * HV_CNT_COUNT, v->stream));
    if (v->touched_stamp) HV_HIP(hipMemsetAsync(v->touched_stamp, 0, sizeof(int32_t) * v->table_capacity, v->stream));
    if (v->touched_mask) HV_HIP(hipMemsetAsync(v->touched_mask, 0, sizeof(uint64_t) * HV_TSDF_SETS * v->table_capacity, v->stream));
    memset(v->h_counters, 0, sizeof(int32_t) * HV_CNT_COUNT);
    v->content_version += 1;
    v->extract_epoch += 1;
    v->frame_counter = 0;
    v->merge_stamp = 0;
    // pool occupancy: known exactly again (the stream is drained first so that no kernel publishes a stale state later)
    HV_HIP(hipStreamSynchronize(v->stream));
    memset(v->h_status, 0, sizeof(HvStatus));
    v->status_seq_issued = v->status_seq_seen = 0;
    v->known_blocks = 0;
    v->max_new_per_call = 0;
    v->avg_new_per_call = 0.0;
    v->status_exact = true;
    v->bins_clean = false; // the grid modes' per-call bins restart clean (hv_bins_ensure)
    v->last_touch_parity = 0;
    v->touch_counters_clean = true;
    v->overflow_latched = false;
    return HV_OK;
}

int hv_synchronize(hv_volume *v) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_synchronize: null volume");
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}

int hv_set_stream(hv_volume *v, void *hip_stream) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_set_stream: null volume");
    HV_HIP(hipStreamSynchronize(v->stream));
    if (v->own_stream && v->stream) HV_HIP(hipStreamDestroy(v->stream));
    v->stream = (hipStream_t)hip_stream;
    v->own_stream = false;
    return HV_OK;
}

void *hv_get_stream(hv_volume *v) { return v ? (void *)v->stream : nullptr; }

// Re-insert the allocated blocks' keys into a fresh (larger) table: vals[slot] = pool index.
__global__ void k_rehash(HvTable t, const unsigned long long *__restrict__ block_keys, int32_t n) {
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const unsigned long long key = block_keys[idx];
    uint32_t s = hv_slot_hash(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        if (atomicCAS(&t.keys[s], HV_EMPTY_KEY, key) == HV_EMPTY_KEY) {
            t.vals[s] = idx;
            return;
        }
        s = (s + 1) & t.mask;
    }
}

// Carry the per-slot "last touched" stamps of the TSDF mode over to a rebuilt table (slots move; pool indices do not).
__global__ void k_restamp(HvTable old_t, HvTable new_t, const int32_t *__restrict__ old_stamp, int32_t *__restrict__ new_stamp, int32_t n) {
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const unsigned long long key = new_t.block_keys[idx];
    const int32_t so = hv_table_find(old_t, key), sn = hv_table_find(new_t, key);
    if (so >= 0 && sn >= 0) new_stamp[sn] = old_stamp[so];
}

// Grow the block pool and the hash to `new_max_blocks` (>= the current capacity), keeping every block: the pool is
// copied (block indices persist), the table is rebuilt at 4x the new capacity.  Synchronises the stream.
// Rebuild pool + table at `new_max_blocks` keeping the first `keep` blocks (-1: all that have a pool slot).
static int hv_rebuild(hv_volume *v, int64_t new_max_blocks, int64_t keep);

int hv_reserve_blocks(hv_volume *v, int64_t new_max_blocks) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_reserve_blocks: null volume");
    if (new_max_blocks <= v->cfg.max_blocks) return HV_OK;
    return hv_rebuild(v, new_max_blocks, -1);
}

static int hv_rebuild(hv_volume *v, int64_t new_max_blocks, int64_t keep) {
    HV_REQUIRE(new_max_blocks < (1ll << 30), HV_ERR_INVALID, "hv_reserve_blocks: max_blocks out of range");
    const uint64_t new_cap = next_pow2((uint64_t)new_max_blocks * 4);
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF || (new_cap << v->local_bits) < (1ull << 32), HV_ERR_CAPACITY,
               "hv_reserve_blocks: %lld blocks are too many for 32-bit (slot, voxel) sort keys", (long long)new_max_blocks);
    HV_HIP(hipSetDevice(v->device));
    HV_HIP(hipStreamSynchronize(v->stream));
    v->extract_epoch += 1; // (the per-unit extraction caches are recomputed in full after a rebuild: rare, and the stamps move with the table)
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    int64_t used = std::min<int64_t>(v->h_counters[HV_CNT_BLOCKS], v->cfg.max_blocks);
    if (keep >= 0) used = std::min(used, keep);
    size_t free_b = 0, total_b = 0;
    HV_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)new_max_blocks * v->bytes_per_block + new_cap * 24 + (size_t)new_max_blocks * 16;
    HV_REQUIRE(need < free_b, HV_ERR_CAPACITY, "hv_reserve_blocks: %.1f GiB needed, %.1f GiB of HBM free", need / 1073741824.0,
               free_b / 1073741824.0);
    void *pool = nullptr;
    unsigned long long *keys = nullptr, *block_keys = nullptr, *occ = nullptr;
    int32_t *vals = nullptr, *stamp = nullptr, *list = nullptr;
    uint64_t *mask = nullptr;
    // every allocation is released again if a later step fails (HV_HIP returns from the middle)
    auto release = [&]() {
        for (void *p : {pool, (void *)keys, (void *)vals, (void *)block_keys, (void *)stamp, (void *)list, (void *)mask, (void *)occ})
            if (p) (void)hipFree(p);
    };
#define HV_TRY_GROW(call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            hv_set_error("hv_reserve_blocks: %s failed: %s", #call, hipGetErrorString(e_));        \
            release();                                                                             \
            return HV_ERR_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)
    HV_TRY_GROW(hipMalloc(&pool, (size_t)new_max_blocks * v->bytes_per_block));
    HV_TRY_GROW(hipMalloc((void **)&keys, sizeof(uint64_t) * new_cap));
    HV_TRY_GROW(hipMalloc((void **)&vals, sizeof(int32_t) * new_cap));
    HV_TRY_GROW(hipMalloc((void **)&block_keys, sizeof(uint64_t) * new_max_blocks));
    HV_TRY_GROW(hipMemcpyAsync(pool, v->pool, (size_t)used * v->bytes_per_block, hipMemcpyDeviceToDevice, v->stream));
    HV_TRY_GROW(hipMemsetAsync((char *)pool + (size_t)used * v->bytes_per_block, 0, (size_t)(new_max_blocks - used) * v->bytes_per_block,
                               v->stream));
    HV_TRY_GROW(hipMemcpyAsync(block_keys, v->table.block_keys, sizeof(uint64_t) * used, hipMemcpyDeviceToDevice, v->stream));
    if (v->occ) { // the occupancy bits follow the pool (pool indices are kept)
        const int64_t nvox = (int64_t)v->cfg.block_size * v->cfg.block_size * v->cfg.block_size;
        const size_t new_words = ((size_t)new_max_blocks * nvox + 63) / 64, used_words = ((size_t)used * nvox + 63) / 64;
        HV_TRY_GROW(hipMalloc((void **)&occ, sizeof(unsigned long long) * new_words));
        HV_TRY_GROW(hipMemsetAsync(occ, 0, sizeof(unsigned long long) * new_words, v->stream));
        if (used_words) HV_TRY_GROW(hipMemcpyAsync(occ, v->occ, sizeof(unsigned long long) * used_words, hipMemcpyDeviceToDevice, v->stream));
    }
    HV_TRY_GROW(hipMemsetAsync(keys, 0xFF, sizeof(uint64_t) * new_cap, v->stream));
    HV_TRY_GROW(hipMemsetAsync(vals, 0xFF, sizeof(int32_t) * new_cap, v->stream));
    HvTable nt = v->table;
    nt.keys = keys;
    nt.vals = vals;
    nt.block_keys = block_keys;
    nt.mask = (uint32_t)(new_cap - 1);
    nt.max_blocks = (int32_t)new_max_blocks;
    // the table is rebuilt from the keys of the blocks that HAVE a pool slot: keys whose claim overflowed the old pool
    // (vals = -1) disappear, so a caller that saw the overflow can simply claim again
    if (used > 0) hipLaunchKernelGGL(k_rehash, dim3((unsigned)((used + 255) / 256)), dim3(256), 0, v->stream, nt, block_keys, (int32_t)used);
    HV_TRY_GROW(hipGetLastError());
    if (v->cfg.mode == HV_MODE_TSDF) { // per-slot frame stamps / masks and the touched list follow the table
        HV_TRY_GROW(hipMalloc((void **)&stamp, sizeof(int32_t) * new_cap));
        HV_TRY_GROW(hipMalloc((void **)&list, sizeof(int32_t) * HV_TSDF_SETS * new_max_blocks));
        HV_TRY_GROW(hipMalloc((void **)&mask, sizeof(uint64_t) * HV_TSDF_SETS * new_cap));
        HV_TRY_GROW(hipMemsetAsync(stamp, 0, sizeof(int32_t) * new_cap, v->stream));
        HV_TRY_GROW(hipMemsetAsync(mask, 0, sizeof(uint64_t) * HV_TSDF_SETS * new_cap, v->stream));
        // stamps carry "touched since the last merge": re-stamp the surviving blocks' slots in the new table
        if (used > 0 && v->touched_stamp != nullptr)
            hipLaunchKernelGGL(k_restamp, dim3((unsigned)((used + 255) / 256)), dim3(256), 0, v->stream, v->table, nt,
                               (const int32_t *)v->touched_stamp, stamp, (int32_t)used);
        HV_TRY_GROW(hipGetLastError());
    }
    // the block counter may have run past the old capacity and the overflow counter is set: both describe claims that
    // no longer exist
    const int32_t fixed[2] = {(int32_t)used, 0};
    HV_TRY_GROW(hipMemcpyAsync(&v->table.counters[HV_CNT_BLOCKS], fixed, sizeof(fixed), hipMemcpyHostToDevice, v->stream));
    if (v->cfg.mode == HV_MODE_TSDF) {
        HV_TRY_GROW(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream)); // lists are void now
        v->touch_counters_clean = true;
    }
    HV_TRY_GROW(hipStreamSynchronize(v->stream));
#undef HV_TRY_GROW
    if (v->cfg.mode == HV_MODE_TSDF) {
        (void)hipFree(v->touched_stamp);
        (void)hipFree(v->touched_list);
        (void)hipFree(v->touched_mask);
        v->touched_stamp = stamp;
        v->touched_list = list;
        v->touched_mask = mask;
    }
    if (occ) {
        (void)hipFree(v->occ);
        v->occ = occ;
    }
    (void)hipFree(v->pool);
    (void)hipFree(v->table.keys);
    (void)hipFree(v->table.vals);
    (void)hipFree(v->table.block_keys);
    v->pool = pool;
    v->table = nt;
    v->table_capacity = new_cap;
    v->cfg.max_blocks = new_max_blocks;
    // occupancy is exact again; states published by kernels that ran before the rebuild are stale
    v->h_counters[HV_CNT_BLOCKS] = (int32_t)used;
    v->h_counters[HV_CNT_OVERFLOW] = 0;
    v->h_status->blocks = (int32_t)used;
    v->h_status->overflow = 0;
    v->h_status->seq = v->status_seq_issued;
    v->status_seq_seen = v->status_seq_issued;
    v->known_blocks = used;
    v->status_exact = true;
    v->overflow_latched = false;
    v->bins_clean = false; // grid modes: the per-slot bin arrays follow the table (hv_bins_ensure re-makes / clears them)
    return HV_OK;
}

// Undo a claim pass that did not fit, without touching the pool: the table is emptied and re-filled from the keys of the
// first `keep` blocks (the keys claimed by the failed call - with or without a block - are gone), the TSDF mode's per-slot
// stamps follow their keys, the counters are repaired.  Blocks beyond `keep` were handed out by the failed call and never
// written (a claim pass writes no voxel), so they are still zero.  Temporary memory: a copy of the old table (20 B per slot).
static int hv_rollback_claims(hv_volume *v, int64_t keep) {
    HV_HIP(hipSetDevice(v->device));
    HV_HIP(hipStreamSynchronize(v->stream));
    v->extract_epoch += 1;
    const uint64_t cap = v->table_capacity;
    keep = std::max<int64_t>(0, std::min<int64_t>(keep, v->cfg.max_blocks));
    unsigned long long *old_keys = nullptr;
    int32_t *old_vals = nullptr, *old_stamp = nullptr;
    auto release = [&]() {
        for (void *p : {(void *)old_keys, (void *)old_vals, (void *)old_stamp})
            if (p) (void)hipFree(p);
    };
#define HV_TRY_RB(call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            hv_set_error("%s failed: %s", #call, hipGetErrorString(e_));                           \
            release();                                                                             \
            return HV_ERR_DEVICE;                                                                  \
        }                                                                                          \
    } while (0)
    const bool tsdf = v->cfg.mode == HV_MODE_TSDF && v->touched_stamp != nullptr;
    if (tsdf) {
        HV_TRY_RB(hipMalloc((void **)&old_keys, sizeof(uint64_t) * cap));
        HV_TRY_RB(hipMalloc((void **)&old_vals, sizeof(int32_t) * cap));
        HV_TRY_RB(hipMalloc((void **)&old_stamp, sizeof(int32_t) * cap));
        HV_TRY_RB(hipMemcpyAsync(old_keys, v->table.keys, sizeof(uint64_t) * cap, hipMemcpyDeviceToDevice, v->stream));
        HV_TRY_RB(hipMemcpyAsync(old_vals, v->table.vals, sizeof(int32_t) * cap, hipMemcpyDeviceToDevice, v->stream));
        HV_TRY_RB(hipMemcpyAsync(old_stamp, v->touched_stamp, sizeof(int32_t) * cap, hipMemcpyDeviceToDevice, v->stream));
    }
    HV_TRY_RB(hipMemsetAsync(v->table.keys, 0xFF, sizeof(uint64_t) * cap, v->stream));
    HV_TRY_RB(hipMemsetAsync(v->table.vals, 0xFF, sizeof(int32_t) * cap, v->stream));
    if (keep > 0) hipLaunchKernelGGL(k_rehash, dim3((unsigned)((keep + 255) / 256)), dim3(256), 0, v->stream, v->table, v->table.block_keys, (int32_t)keep);
    HV_TRY_RB(hipGetLastError());
    if (tsdf) {
        HV_TRY_RB(hipMemsetAsync(v->touched_stamp, 0, sizeof(int32_t) * cap, v->stream));
        HV_TRY_RB(hipMemsetAsync(v->touched_mask, 0, sizeof(uint64_t) * HV_TSDF_SETS * cap, v->stream));
        HvTable old_t = v->table;
        old_t.keys = old_keys;
        old_t.vals = old_vals;
        if (keep > 0)
            hipLaunchKernelGGL(k_restamp, dim3((unsigned)((keep + 255) / 256)), dim3(256), 0, v->stream, old_t, v->table,
                               (const int32_t *)old_stamp, v->touched_stamp, (int32_t)keep);
        HV_TRY_RB(hipGetLastError());
        HV_TRY_RB(hipMemsetAsync(&v->table.counters[HV_CNT_TOUCH0], 0, HV_CNT_TOUCH_SPAN_BYTES, v->stream));
        v->touch_counters_clean = true;
    }
    const int32_t fixed[2] = {(int32_t)keep, 0};
    HV_TRY_RB(hipMemcpyAsync(&v->table.counters[HV_CNT_BLOCKS], fixed, sizeof(fixed), hipMemcpyHostToDevice, v->stream));
    HV_TRY_RB(hipStreamSynchronize(v->stream));
#undef HV_TRY_RB
    release();
    v->h_counters[HV_CNT_BLOCKS] = (int32_t)keep;
    v->h_counters[HV_CNT_OVERFLOW] = 0;
    v->h_status->blocks = (int32_t)keep;
    v->h_status->overflow = 0;
    v->h_status->seq = v->status_seq_issued;
    v->status_seq_seen = v->status_seq_issued;
    v->known_blocks = keep;
    v->status_exact = true;
    v->overflow_latched = false;
    v->bins_clean = false; // grid modes: the per-slot bin arrays are keyed by slots that just moved - re-made / cleared by the next call
    return HV_OK;
}

// Every call that returns data to the host passes through here (one counter read-back): the natural place to grow
// the pool before it runs out.  Policy: when more than half of the blocks are in use, double the capacity (while the
// doubled pool fits in 60 % of the free HBM and the sort keys).  HV_AUTO_GROW=0 disables it.
int hv_num_blocks(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_num_blocks: null argument");
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    int64_t nb = v->h_counters[HV_CNT_BLOCKS];
    if (nb > v->cfg.max_blocks) nb = v->cfg.max_blocks;
    *n = nb;
    HV_REQUIRE(v->h_counters[HV_CNT_OVERFLOW] == 0, HV_ERR_CAPACITY,
               "block pool exhausted: max_blocks=%lld; recreate the volume with a larger pool (or call hv_reserve_blocks "
               "earlier: blocks that did not fit were dropped)",
               (long long)v->cfg.max_blocks);
    v->known_blocks = nb; // the stream is idle: exact
    v->status_seq_seen = v->status_seq_issued;
    v->status_exact = true;
    if (nb * 2 > v->cfg.max_blocks) (void)hv_grow(v, v->cfg.max_blocks * 2); // best effort; the integrate calls check again
    return HV_OK;
}

int hv_max_blocks(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_max_blocks: null argument");
    *n = v->cfg.max_blocks;
    return HV_OK;
}

int hv_block_size(hv_volume *v, int32_t *bs) {
    HV_REQUIRE(v != nullptr && bs != nullptr, HV_ERR_INVALID, "hv_block_size: null argument");
    *bs = v->cfg.block_size;
    return HV_OK;
}

int hv_bytes_per_block(hv_volume *v, int64_t *bytes) {
    HV_REQUIRE(v != nullptr && bytes != nullptr, HV_ERR_INVALID, "hv_bytes_per_block: null argument");
    *bytes = v->bytes_per_block;
    return HV_OK;
}

int hv_dropped_points(hv_volume *v, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_dropped_points: null argument");
    int rc = hv_read_counters(v);
    if (rc != HV_OK) return rc;
    *n = v->h_counters[HV_CNT_DROPPED];
    return HV_OK;
}

int hv_profile_enable(hv_volume *v, int32_t on) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_profile_enable: null volume");
    v->profiling = on != 0;
    v->events_used = 0;
    v->prof_units = 0;
    return HV_OK;
}

int hv_profile_read(hv_volume *v, double *kernel_ms_total, int64_t *kernel_launches, int64_t *units_processed) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_profile_read: null volume");
    HV_HIP(hipStreamSynchronize(v->stream));
    double total = 0.0;
    for (size_t i = 0; i < v->events_used; ++i) {
        float ms = 0.f;
        HV_HIP(hipEventElapsedTime(&ms, v->events[i].start, v->events[i].stop));
        total += ms;
    }
    if (kernel_ms_total) *kernel_ms_total = total;
    if (kernel_launches) *kernel_launches = (int64_t)v->events_used;
    if (units_processed) *units_processed = v->prof_units;
    v->events_used = 0;
    v->prof_units = 0;
    return HV_OK;
}

int hv_profile_read_launches(hv_volume *v, float *launch_ms, int64_t cap, int64_t *n) {
    HV_REQUIRE(v != nullptr && n != nullptr, HV_ERR_INVALID, "hv_profile_read_launches: null argument");
    HV_HIP(hipStreamSynchronize(v->stream));
    *n = (int64_t)v->events_used;
    for (size_t i = 0; i < v->events_used && (int64_t)i < cap && launch_ms != nullptr; ++i)
        HV_HIP(hipEventElapsedTime(&launch_ms[i], v->events[i].start, v->events[i].stop));
    return HV_OK;
}

} // extern "C"
