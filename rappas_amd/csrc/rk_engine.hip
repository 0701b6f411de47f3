// rk_engine.hip -- host side of the placement engine: DB image construction (open-addressed / direct table
// + CSR row blob resident in HBM), launch geometry, and the C ABI of include/rappas_place.h.
// Product path: there is NO CPU fallback in this file; every compute entry point needs a HIP device.
#include "rk_kernels.hip"
#include "rk_internal.h"
#include "rk_pack_host.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <cerrno>
#include <cstddef>
#include <deque>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <atomic>
#include <mutex>
#include <tuple>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <fstream>
#include <pthread.h>
#include <sched.h>

using namespace rk;

#ifndef RK_WG_MIN_BRANCHES
#define RK_WG_MIN_BRANCHES 8192u  // above this a single-wave score vector leaves <= 4 waves per CU: use place_wg_kernel
#endif
#ifndef RK_WG_MIN_MEAN_ROW
#define RK_WG_MIN_MEAN_ROW 320.0    // ... for rows at least this long on average (scripts/long_rows_big_tree.py: up to ~300 entries the windowed kernel is ahead:
                                    // 15 999 branches, mean row 70 / 150 / 300: 54 / 36 / 19 against 22 / 19 / 17 Mreads/s) ...
#endif
#ifndef RK_WG_ALWAYS_BRANCHES
#define RK_WG_ALWAYS_BRANCHES 65535u  // ... or, whatever the rows, above this many branches (none: the windowed kernel's 64 windows of 1 024 cover every tree)
#endif
#ifndef RK_WINDOW_MIN_BRANCHES
#define RK_WINDOW_MIN_BRANCHES 1276u  // the dense 16-lane geometry keeps eight waves per CU up to 1 116 branches and seven up to 1 276; the switch sits where the seventh wave goes (round 3, scripts/tree_size_sweep.py, dense against windowed Mreads/s: 1 117 branches 291 / 247, 1 200: 257 / ~245, 1 290 (six waves): 219 / 242, 1 400: 218 / 242, 2 800: 110 / 188)
#endif
#ifndef RK_RING
#define RK_RING 8  // depth of the row-chunk register ring (chunks in flight per lane)
#endif
#ifndef RK_WG_RING
#define RK_WG_RING 20  // the same for the large-tree kernel, whose waves stream long row slices from HBM: what counts there is bytes in flight
                       // (C5s: rings of 8 / 12 / 16 / 20 / 24 / 32 give 0.72 / 0.76 / 0.77 / 0.79 / 0.78 / 0.77 of the byte roofline;
                       // C5 at 200 GB 0.61 -> 0.65, C5m 0.60 -> 0.63; short-row trees beyond 32 000 branches lose 7 %)
#endif
#ifndef RK_WRING
#define RK_WRING RK_RING  // the same for the windowed kernel
#endif
#ifndef RK_RING64
#define RK_RING64 8  // ... and for the 64-lane geometry (one wave per read: large trees with long rows, one or two waves per SIMD)
#endif
#ifndef RK_HRING
#define RK_HRING 2  // ... and for the hash-accumulator kernel (place_hash64_kernel: steps of RK_HNPL entries per lane = 4 * RK_HNPL row units)
#endif
#ifndef RK_HNPL
#define RK_HNPL 4
#endif

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace rk {
int fail_msg(int code, const char *fmt, ...) {  // rk_internal.h: the same sink for the other translation units
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rk

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                     \
    } while (0)

extern "C" const char *rk_last_error(void) { return g_err; }

// The placement kernels are launched as persistent grids: one block per slot a CU really has, every block walking the batch with
// the grid's stride.  The slots are what the runtime says fit (registers, the LDS allocation granule), not LDS size / LDS
// per block: a grid one block per CU too large runs a second round and costs up to 2 x (seen: 23 360 B of LDS per block, seven
// by division, six resident; 32-lane geometry with 11 by division, 8 by registers).
template <typename K>
static int resident_blocks(K kern, int block_threads, size_t lds, uint64_t by_lds, uint64_t &out) {
    int n = 0;
    {  // (asked once per kernel, block size, LDS size and device: the query is not free next to a small launch)
        static std::mutex mu;
        static std::map<std::tuple<const void *, int, size_t, int>, int> known;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        const auto key = std::make_tuple((const void *)kern, block_threads, lds, dev);
        std::lock_guard<std::mutex> lock(mu);
        auto it = known.find(key);
        if (it != known.end()) {
            n = it->second;
        } else {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)kern, block_threads, lds));
            known.emplace(key, n);
        }
    }
    if (n < 1) return fail(RK_ERR_UNSUPPORTED, "internal: kernel does not fit a CU (%d threads, %zu B of LDS per block)", block_threads, lds);
    out = by_lds < (uint64_t)n ? by_lds : (uint64_t)n;
    // The LDS is handed out in granules of 512 bytes, which the occupancy query does not count in: 23 232 B per block -- seven by
    // division and by the query -- are 23 552 B each, and six are resident (round 3, scripts/tree_size_sweep.py at 1 290 branches:
    // the seventh block of every CU ran in a second round, 140 instead of 250 Mreads/s).
    if (lds) {
        const uint64_t by_granule = (160ull * 1024) / ((lds + 511) & ~(size_t)511);
        if (by_granule >= 1 && by_granule < out) out = by_granule;
    }
    static const bool trace = rk_knob("RK_TRACE_GRID") != nullptr;  // developer knob
    if (trace) fprintf(stderr, "[rk] grid: %d threads, %zu B LDS per block -> %d resident per CU (by LDS size %llu)\n", block_threads, lds, n, (unsigned long long)by_lds);
    return RK_OK;
}
extern "C" int rk_version(void) { return RK_VERSION; }

// Main_DBBUILD_3.java:165-166
extern "C" void rk_thresholds(float omega, uint32_t n_states, uint32_t k, float *thr, float *thr_log10) {
    float ratio = omega / (float)n_states;
    float p = (float)std::pow(0.0 + (double)ratio, (double)k);
    if (thr) *thr = p;
    if (thr_log10) *thr_log10 = (float)std::log10((double)p);
}

// ------------------------------------------------------------------------------------------------
// alphabets (host tables uploaded with the DB)
// DNA: src/core/DNAStatesShifted.java:45-96 (ambiguity sets), :182-209 (states).
// AA : src/core/AAStates.java:23-28, :68-123.
// ------------------------------------------------------------------------------------------------
struct Alphabet {
    unsigned char table[256];      // state | 0x80|class | 0xFF
    unsigned char alts[16 * 20];   // alternatives per class
    unsigned char alt_count[16];
};

static void build_alphabet(uint32_t alphabet, bool convert_uo, Alphabet &A) {
    memset(A.table, 0xFF, sizeof(A.table));
    memset(A.alts, 0, sizeof(A.alts));
    memset(A.alt_count, 0, sizeof(A.alt_count));
    auto both = [&](char up, unsigned char v) {
        A.table[(unsigned char)up] = v;
        A.table[(unsigned char)(up + 32)] = v;
    };
    if (alphabet == RK_ALPHABET_DNA) {
        both('A', 0); both('T', 1); both('U', 1); both('C', 2); both('G', 3);
        const unsigned char a = 0, t = 1, c = 2, g = 3;
        struct { char ch; int n; unsigned char s[4]; } cls[] = {
            {'R', 2, {a, g}}, {'Y', 2, {c, t}}, {'S', 2, {c, g}}, {'W', 2, {a, t}}, {'K', 2, {g, t}},
            {'M', 2, {a, c}}, {'B', 3, {c, g, t}}, {'D', 3, {a, g, t}}, {'H', 3, {a, c, t}},
            {'V', 3, {a, c, g}}, {'N', 4, {a, c, g, t}},
        };
        int ci = 0;
        for (auto &e : cls) {
            both(e.ch, (unsigned char)(0x80 | ci));
            A.alt_count[ci] = (unsigned char)e.n;
            for (int i = 0; i < e.n; i++) A.alts[ci * 20 + i] = e.s[i];
            ci++;
        }
        // '.' and '-' : four never-filled (zero) alternatives
        A.table[(unsigned char)'.'] = A.table[(unsigned char)'-'] = (unsigned char)(0x80 | ci);
        A.alt_count[ci] = 4;
    } else {
        const char *order = "RHKDESTNQCGPAILMFWYV";
        for (int i = 0; i < 20; i++) both(order[i], (unsigned char)i);
        if (convert_uo) { both('U', 9); both('O', 14); }
        // class 0: any
        for (char ch : {'-', '*', '!', 'X', 'x'}) A.table[(unsigned char)ch] = 0x80;
        A.alt_count[0] = 20;
        for (int i = 0; i < 20; i++) A.alts[i] = (unsigned char)i;
        both('B', 0x81); A.alt_count[1] = 2; A.alts[20 + 0] = 3;  A.alts[20 + 1] = 7;
        both('Z', 0x82); A.alt_count[2] = 2; A.alts[40 + 0] = 4;  A.alts[40 + 1] = 8;
        both('J', 0x83); A.alt_count[3] = 2; A.alts[60 + 0] = 13; A.alts[60 + 1] = 14;
    }
}

// ------------------------------------------------------------------------------------------------
// DB object
// ------------------------------------------------------------------------------------------------
namespace {
struct GrowBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return RK_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return RK_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace

namespace {
// The host CPUs next to a GPU (its PCI device's NUMA node, from sysfs), cut to what this process may run on: the staging threads of
// the host path and the page-locked buffers they fill are kept there (round 3 measured 2.0 - 2.6e8 reads/s for the same call
// depending on where the scheduler had put them).  `ok` false = unknown / one node / nothing left after the cut: nothing is pinned.
struct NodeCpus {
    cpu_set_t set;
    bool ok = false;
    int node = -1;
};
const NodeCpus &gpu_node_cpus(int device) {
    static std::mutex mu;
    static std::map<int, NodeCpus> known;
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    NodeCpus nc;
    CPU_ZERO(&nc.set);
    char bus[64] = "";
    if (rk_knob("RK_NO_NUMA")) return known.emplace(device, nc).first->second;  // developer knob (A/B)
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) == hipSuccess && bus[0]) {
        for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
        int node = -1;
        { std::ifstream f(std::string("/sys/bus/pci/devices/") + bus + "/numa_node"); if (f) f >> node; }
        std::string list;
        if (node >= 0) { std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"); if (f) std::getline(f, list); }
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (!list.empty() && sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
            int n_set = 0;
            const char *q = list.c_str();
            while (*q) {  // "0-23,96-119"
                char *e;
                long a = strtol(q, &e, 10), b = a;
                if (e == q) break;
                if (*e == '-') { q = e + 1; b = strtol(q, &e, 10); }
                for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                    if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, &nc.set); n_set++; }
                q = (*e == ',') ? e + 1 : e;
                if (*e != ',' ) break;
            }
            nc.ok = n_set >= 4 && n_set < CPU_COUNT(&allowed);  // (all of the allowed CPUs on that node: nothing to choose)
            nc.node = node;
        }
    } else {
        (void)hipGetLastError();
    }
    return known.emplace(device, nc).first->second;
}
void pin_this_thread(const NodeCpus *nc) {
    if (nc && nc->ok) (void)pthread_setaffinity_np(pthread_self(), sizeof(nc->set), &nc->set);
}

struct PinBuf {  // page-locked host staging, grow-only
    void *p = nullptr;
    size_t cap = 0;
    // (node: allocated by a short-lived thread that runs next to the GPU, so that the pages -- pinned as they are allocated -- come
    //  from that node's memory; the caller's own thread is never moved)
    int reserve(size_t n, const NodeCpus *node = nullptr, int device = 0) {
        if (n <= cap) return RK_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + n / 4 + 256;
        hipError_t e = hipSuccess;
        bool done = false;
        if (node && node->ok) {
            try {
                std::thread t([&]() {
                    pin_this_thread(node);
                    (void)hipSetDevice(device);
                    e = hipHostMalloc(&p, want, hipHostMallocDefault);
                });
                t.join();
                done = true;
            } catch (...) {  // no thread to be had: allocate here
            }
        }
        if (!done) e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        return RK_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() { return (T *)p; }
};

// memcpy split over a few host threads: one thread moves ~10 GB/s, the PCIe link five times that
void parallel_copy(void *dst, const void *src, size_t bytes) {
    const size_t min_part = 2u << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t parts = std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 12u), bytes / min_part);
    if (parts <= 1) { if (bytes) memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    for (size_t i = 1; i < parts; i++) {
        const size_t a = bytes * i / parts, b = bytes * (i + 1) / parts;
        th.emplace_back([=]() { memcpy((char *)dst + a, (const char *)src + a, b - a); });
    }
    memcpy(dst, src, bytes / parts);
    for (std::thread &t : th) t.join();
}

// A few worker threads that live for the duration of ONE host call: run(fn) executes fn(part, parts) on every worker and on the
// caller and returns when all are done (a chunk of 2^18 reads is packed in under a millisecond -- starting threads per chunk would
// cost as much as the work).  Joined in the destructor, so no path out of the call leaves a thread behind.
class ForkJoin {
  public:
    explicit ForkJoin(unsigned workers, const NodeCpus *node = nullptr) {  // node: the workers run on the CPUs next to the GPU
        for (unsigned i = 0; i < workers; i++) th_.emplace_back([this, i, node]() { pin_this_thread(node); loop(i + 1); });
    }
    ~ForkJoin() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_.fetch_add(1); }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    unsigned parts() const { return (unsigned)th_.size() + 1; }
    void run(const std::function<void(unsigned, unsigned)> &fn) {  // the workers and the caller, each one part; returns when all are done
        if (th_.empty()) { fn(0, 1); return; }
        post(&fn, 0, parts());
        fn(0, parts());
        wait();
    }
    // the workers alone, while the caller does something else; wait() before the next start() / run().  Without workers the
    // function runs in start().
    void start(std::function<void(unsigned, unsigned)> fn) {
        if (th_.empty()) { fn(0, 1); return; }
        own_ = std::move(fn);
        post(&own_, 1, (unsigned)th_.size());
    }
    void wait() {
        for (int spin = 0; spin < 4000 && left_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
        if (left_.load(std::memory_order_acquire) == 0) return;
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return left_.load() == 0; });
    }

  private:
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void post(const std::function<void(unsigned, unsigned)> *fn, unsigned base, unsigned parts) {
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = fn; base_ = base; parts_ = parts;
            left_.store((unsigned)th_.size(), std::memory_order_release);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
    }
    // A chunk of the host path is staged in well under a millisecond, so a worker that has just finished one job spins for a few
    // tens of microseconds before it blocks: the next job usually arrives within that time and a futex wake-up costs as much.
    void loop(unsigned me) {
        uint64_t seen = 0;
        while (true) {
            for (int spin = 0; spin < 3000 && gen_.load(std::memory_order_acquire) == seen; spin++) cpu_relax();  // (~30 us; a hosting JVM has pools of its own to feed)
            const std::function<void(unsigned, unsigned)> *fn;
            unsigned part, parts;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return gen_.load() != seen; });
                seen = gen_.load();
                if (stop_) return;
                fn = fn_;
                part = me - base_;
                parts = parts_;
            }
            (*fn)(part, parts);
            if (left_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)> *fn_ = nullptr;
    std::function<void(unsigned, unsigned)> own_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<unsigned> left_{0};
    unsigned base_ = 0, parts_ = 1;
    bool stop_ = false;
};

// rk_place_batch_multi runs one host call per GPU at the same time: each takes its share of the thread budget
thread_local unsigned tl_concurrent_calls = 1;
unsigned host_threads(uint64_t n_reads, unsigned asked) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned T = asked ? asked : std::max(2u, std::min(hw ? hw : 1u, 16u * tl_concurrent_calls) / tl_concurrent_calls);
    if (!asked && T > 16u) T = 16u;
    return n_reads < 4096 ? 1u : T;
}
}  // namespace

struct rk_workspace {
    GrowBuf ascii, off, packed, lens, flags, nrows, branch, score, lwr, oflags;
    // page-locked staging for callers that hand over pageable memory (a JVM heap array, a numpy array): copies to / from
    // it run on a few host threads, the DMA itself is then asynchronous and overlaps the other workspace's chunk
    PinBuf h_ascii, h_off, h_packed, h_nrows, h_branch, h_score, h_lwr, h_oflags;
    bool pending = false;       // results of the last chunk are still in the staging buffers
    uint64_t pend_r0 = 0, pend_n = 0;
    hipStream_t stream = nullptr;
    void release() {
        for (GrowBuf *b : {&ascii, &off, &packed, &lens, &flags, &nrows, &branch, &score, &lwr, &oflags}) b->release();
        for (PinBuf *b : {&h_ascii, &h_off, &h_packed, &h_nrows, &h_branch, &h_score, &h_lwr, &h_oflags}) b->release();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

// mid-size trees: the score vector of a read is held one window of W branches at a time (place_packed16w_kernel)
constexpr uint32_t RK_MAX_WINDOWS = 64;  // 6-bit window ids in winspec and in the item tags
// winspec byte of a row that reaches the windows first .. last: first | span << 6, span 3 = "at least three more: to the last window"
static inline unsigned char winspec_byte(uint32_t first, uint32_t last) {
    const uint32_t span = last - first;
    return (unsigned char)(first | ((span < 3u ? span : 3u) << 6));
}
constexpr uint64_t RK_WINDOW_MAX_BLOB = 1ull << 31;  // list items carry a 24-bit index of 128-byte units
// The windowed kernel holds a read's row units in a list of a few hundred items: beyond ~2.2 units per k-mer code (a 150-bp read
// then brings ~300) a read is emitted in many window ranges and the dense kernels are ahead (scripts/row_length_sweep.py, 3 999
// branches, mean row 100 / 250 entries: 113 / 35 against 87 / 49 Mreads/s) -- unless the tree is so large that they hold one read
// in two or three waves per CU (beyond 8 192 branches; scripts/dense_rows_mid_tree.py, every k-mer present with rows of 30 / 60
// entries: 9 001 branches 66 / 45 against 41 / 36, 15 999: 49 / 36 against 17 / 15; longer rows there take the large-tree image)
static inline bool windows_pay(uint32_t nb, uint64_t blob_units, uint64_t space) {
    if (rk_knob("RK_WINDOW_ALWAYS")) return true;  // developer / test knob: the windowed kernel whatever the row density
    return 5 * blob_units <= 11 * space || nb > RK_WG_MIN_BRANCHES;
}
struct WindowPlan {
    uint32_t W = 0, n_win = 0, s_stride = 0, main_cap = 0, work_cap = 0;
    bool stream = false;          // place_packed16s_kernel first (decided when the image is built: its windows are narrower)
    double units_per_code = 0.0;  // row units / k-mer codes of the alphabet: what a read's k-mer brings on average
};

// The dense 16-lane geometry keeps eight waves per CU up to 1 116 branches (choose_geometry); beyond that -- and up to the
// 65 535 branches of the reference -- the tree is cut into windows of <= 1 024 branches, at most 64 of them
// (6-bit window ids in winspec and in the item tags), sized so that a wave's four reads fit 20 KB of LDS: 8 waves per CU again.
#ifndef RK_WSTREAM_MIN_BRANCHES
#define RK_WSTREAM_MIN_BRANCHES 4500u
#endif
// Row units a 150-symbol read brings (its ~150 k-mers x the row units per k-mer CODE of the image) beyond which the sorted list of
// place_packed16s_kernel stops fitting and place_packed16w_kernel alone is ahead: scripts/long_rows_big_tree.py, 9 001 branches,
// a quarter of the 9-mers present, rows of 70 / 150 / 300 entries (~170 / 350 / 680 units a read): 73 / 35 / 15 Mreads/s with the
// second kernel first, 77 / 46 / 21 without (profiles/r03_long_rows_big_tree.txt)
#ifndef RK_WSTREAM_MAX_UNITS
#define RK_WSTREAM_MAX_UNITS 220.0
#endif
static bool wstream_tree(uint32_t nb, uint32_t bits, double units_per_code, double units_per_row) {  // images whose tiles go to place_packed16s_kernel first
    if (rk_knob("RK_WSTREAM_ALWAYS")) return true;  // developer / test knob: the sorted-stream kernel on every windowed tree
    if (150.0 * units_per_code > RK_WSTREAM_MAX_UNITS) return false;
    // rows of several units (70 entries: 5) put few, long runs of slots into a window -- three or four rounds at its end -- and few
    // units per window and read, so that most of the list is padding: 9 001 / 15 999 / 25 001 branches 73 / 44 / 28 Mreads/s against
    // 77 / 53 / 39 (same profile); short rows only
    if (units_per_row > 2.5) return false;
    // amino acids: every windowed tree (the first kernel runs with register spills to keep two waves per SIMD: 274 / 248 / 194
    // Mreads/s at 1 400 / 1 999 / 3 999 branches against 280 / 272 / 241, C4-like rows; profiles/r03_wstream_crossover.txt)
    return bits == 5 || nb > RK_WSTREAM_MIN_BRANCHES;
}
static bool window_plan(uint32_t nb, uint32_t bits, double units_per_code, double units_per_row, WindowPlan &wp) {
    if (nb <= RK_WINDOW_MIN_BRANCHES || nb > RK_WG_ALWAYS_BRANCHES) return false;
    // up to ~4 500 branches place_packed16w_kernel is ahead (windows of <= 1 024 branches, as few as possible: it pays per window);
    // beyond, place_packed16s_kernel (round 3: a window's cost follows what the read touches in it, so more and smaller windows --
    // about 500 branches, one bitmap word a lane -- cost little and leave the LDS to the list).  At most 64 windows: up to 1 024
    // branches each on the largest trees.  scripts/wstream_crossover.py, Mreads/s at 3 999 / 4 999 / 5 999 / 7 999 branches:
    // 164 / 147 / 132 / 113 with the first kernel, 159 / 150 / 144 / 135 with the second (profiles/r03_wstream_crossover.txt)
    wp.stream = wstream_tree(nb, bits, units_per_code, units_per_row);
    wp.units_per_code = units_per_code;
    uint32_t n_win = wp.stream ? (nb + 511) / 512 : (nb + 895) / 896;
    if (n_win > RK_MAX_WINDOWS) n_win = RK_MAX_WINDOWS;
    wp.n_win = n_win;
    wp.W = ((nb + n_win - 1) / n_win + 3) & ~3u;
    wp.s_stride = wp.W + 4;
    const uint32_t per_group_words = 160 * 1024 / 8 / 4 / 4;  // 1280 u32 words per read = eight waves per CU
    // place_packed16w_kernel: the main list has to hold a whole read (C2-like reads: 145 row units on average, 250 at the tail); what
    // is left goes to the per-window work list, so that a window is normally applied in one accumulate call
    const uint32_t avail = per_group_words - wp.s_stride;
    // (88 words = the 44 keys the exact select of a window needs as scratch)
    uint32_t work = avail > 256 + 88 ? avail - 256 : 88;
    if (work > 200) work = 200;
    wp.work_cap = work & ~1u;
    wp.main_cap = (avail - wp.work_cap) & ~1u;
    if (wp.main_cap > 640) wp.main_cap = 640;
    return wp.main_cap >= 160;
}

// Which image a database gets, from its row lengths alone (slot_units = sum of ceil(len / 16) + 1, max_len, mean_len):
//   windowed (slot-offset rows + winspec, place_packed16w_kernel) when window_plan has a plan, the compact table applies (DIRECT or
//     AUTO over a small enough key space, no row beyond 255 units), the blob stays below 2 GB and windows_pay says so;
//   large-tree (indexed rows, place_wg_kernel) beyond 8 192 branches when the rows are long (mean >= RK_WG_MIN_MEAN_ROW), or when
//     no windowed image is possible beyond 16 000 branches (there the dense kernels hold one read per CU, or none at all beyond
//     ~39 000);
//   the plain slot-offset image of the dense kernels otherwise.
struct ImageKind { bool indexed, windowable; };
static ImageKind image_kind(uint32_t nb, uint32_t bits, uint32_t table_mode, uint64_t space, bool space_ok, uint64_t slot_units, uint32_t max_len, double mean_len) {
    WindowPlan wp;
    const bool direct = table_mode == RK_TABLE_DIRECT || (table_mode == RK_TABLE_AUTO && space_ok && space <= (1ull << 28));
    // Just beyond 8 192 branches the dense 64-lane kernel (one wave per read, the whole score vector in the LDS) still holds three or
    // four reads per CU, and with rows of a few hundred entries it is ahead of both its neighbours (scripts/long_rows_big_tree.py,
    // scripts/long_rows_lanes64.py, profiles/r03_long_rows_big_tree.txt; Mreads/s, dense against the other):
    //   the windowed kernel, rows of 150 / 300: 9 001 branches 46.7 / 36.3 against 45.6 / 20.6; 13 001: 30.4 / 24.5 against 39.3 / 19.1;
    //   the workgroup-per-read kernel between its two regimes (slices of a row shorter than a turn of the ring), rows of 400 / 1 000:
    //   9 001 branches 31.0 / 14.3 against 16.2 / 12.3; 11 001: 22.2 / 11.0 against 16.0 / 12.1; 13 001: 21.3 / 10.7 against 15.9 / 12.1;
    //   15 999: 13.5 / 7.2 against 15.8 / 11.9
    const bool dense64_ahead = nb > RK_WG_MIN_BRANCHES && ((nb <= 9900u && mean_len >= 200.0 && mean_len < 1200.0) || (nb <= 13300u && mean_len >= 250.0 && mean_len < 750.0));
    const bool windowable = (!dense64_ahead || rk_knob("RK_WINDOW_ALWAYS")) && window_plan(nb, bits, space ? (double)slot_units / (double)space : 0.0, mean_len / ROW_UNIT + 0.5, wp) && direct &&
                            (max_len + ROW_UNIT - 1) / ROW_UNIT <= 255 && slot_units * 128 < RK_WINDOW_MAX_BLOB && windows_pay(nb, slot_units, space);
    const bool long_rows = mean_len >= RK_WG_MIN_MEAN_ROW;
    const bool indexed = nb > RK_WG_MIN_BRANCHES && ((long_rows && !dense64_ahead) || (!windowable && !dense64_ahead && nb > 16000u));
    return {indexed, windowable && !indexed};
}

struct rk_db {
    rk_db_info info{};
    uint32_t convert_uo = 0;
    DbView view{};
    void *d_table = nullptr;
    void *d_rows = nullptr;
    unsigned char *d_alpha = nullptr;  // table[256] | alts[320] | alt_count[16]
    unsigned char *d_winspec = nullptr;  // [sigma^k] window span per k-mer (windowed images only)
    bool windowed = false;
    bool has_pos = false;              // d_winspec holds position keys only (dense kernels)
    bool compact_nib = false;          // the compact table holds 4-bit unit counts
    WindowPlan wp;
    uint32_t lanes_per_read = 0;       // 0 = auto
    uint32_t waves_per_block = 1;
    bool indexed = false;              // rows carry an index line (large trees, place_wg_kernel)
    int cu_count = 256;
    size_t lds_per_cu = 160 * 1024;
    hipStream_t stream = nullptr;      // spare stream
    std::mutex host_mutex;             // rk_place_batch (host path) owns the workspaces below
    rk_workspace ws[4];                // device buffers + stream per in-flight chunk (grow-only)
    std::string kernel_name;
    // Scratch of the launches themselves (the tile order's keys / histogram / permutation, the marks of the tiles one kernel hands to
    // the next): one grow-only block per stream a caller has launched on, owned by the handle -- the library allocates from no pool
    // the hosting process shares and changes no attribute of one (launch_scratch)
    struct LaunchScratch { hipStream_t s; void *p; size_t cap; uint64_t used; };
    mutable std::mutex scratch_mu;
    mutable std::vector<LaunchScratch> scratch;
    mutable uint64_t scratch_clock = 0;
};

static uint64_t host_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static bool ipow_fits(uint64_t base, uint32_t e, uint64_t limit, uint64_t &out) {
    uint64_t v = 1;
    for (uint32_t i = 0; i < e; i++) {
        if (v > limit / base) return false;
        v *= base;
    }
    out = v;
    return true;
}

extern "C" void rk_db_destroy(rk_db *db) {
    if (!db) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(db->info.device);
    if (db->d_table) (void)hipFree(db->d_table);
    if (db->d_rows) (void)hipFree(db->d_rows);
    if (db->d_alpha) (void)hipFree(db->d_alpha);
    if (db->d_winspec) (void)hipFree(db->d_winspec);
    if (db->stream) (void)hipStreamDestroy(db->stream);
    for (rk_workspace &w : db->ws) w.release();
    for (rk_db::LaunchScratch &b : db->scratch)
        if (b.p) (void)hipFree(b.p);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete db;
}

// scalars every database carries, whatever built its image
struct DbMeta {
    uint32_t alphabet, convert_uo, k, n_branches;
    float thr_log10, thr;
};

// device selection, properties, spare stream and the alphabet tables; the caller restores the current device
static int open_db(const DbMeta &m, int device, rk_db **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RK_ERR_NO_DEVICE, "rk_db_create: no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return fail(RK_ERR_INVALID, "rk_db_create: device %d out of range (0..%d)", device, ndev - 1);
    rk_db *db = new (std::nothrow) rk_db();
    if (!db) return fail(RK_ERR_NOMEM, "rk_db_create: host OOM");
    db->info.device = device;
#define OPEN_TRY(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int c_ = fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rk_db_destroy(db);                                                                    \
            return c_;                                                                            \
        }                                                                                         \
    } while (0)
    OPEN_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    OPEN_TRY(hipGetDeviceProperties(&prop, device));
    db->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    db->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 64 * 1024;
    OPEN_TRY(hipStreamCreateWithFlags(&db->stream, hipStreamNonBlocking));
    OPEN_TRY(hipMalloc((void **)&db->d_alpha, 256 + 320 + 16));
    Alphabet A;
    build_alphabet(m.alphabet, m.convert_uo != 0, A);
    OPEN_TRY(hipMemcpy(db->d_alpha, A.table, 256, hipMemcpyHostToDevice));
    OPEN_TRY(hipMemcpy(db->d_alpha + 256, A.alts, 320, hipMemcpyHostToDevice));
    OPEN_TRY(hipMemcpy(db->d_alpha + 576, A.alt_count, 16, hipMemcpyHostToDevice));
#undef OPEN_TRY
    *out = db;
    return RK_OK;
}

// info + device view once d_table / d_rows hold the image
static void finish_db(rk_db *db, const DbMeta &m, uint32_t mode, bool indexed, bool mono, uint64_t n_keys, uint64_t n_entries,
                      uint64_t slots, uint64_t hash_mask, uint64_t table_bytes, uint64_t blob_bytes, uint32_t max_len) {
    const uint32_t bits = m.alphabet == RK_ALPHABET_DNA ? 2 : 5;
    db->convert_uo = m.convert_uo;
    db->indexed = indexed;
    db->info.alphabet = m.alphabet; db->info.k = m.k; db->info.n_branches = m.n_branches;
    db->info.table_mode = mode; db->info.thr_log10 = m.thr_log10; db->info.thr = m.thr;
    db->info.n_keys = n_keys; db->info.n_entries = n_entries; db->info.table_slots = slots;
    db->info.table_bytes = table_bytes; db->info.rows_bytes = blob_bytes; db->info.bits_per_symbol = bits;
    db->info.max_row_len = max_len;
    db->view.direct = mode == RK_TABLE_DIRECT8 ? (const u64 *)db->d_table : nullptr;
    db->view.compact = mode == RK_TABLE_DIRECT ? (const uint4 *)db->d_table : nullptr;
    db->view.compact_nib = db->compact_nib ? 1u : 0u;
    db->view.slots = mode == RK_TABLE_HASH ? (const uint4 *)db->d_table : nullptr;
    db->view.hash_mask = hash_mask;
    db->view.rows = (const unsigned char *)db->d_rows;
    db->view.rows_bytes = db->info.rows_bytes;
    db->view.k = m.k; db->view.bits = bits; db->view.n_branches = m.n_branches; db->view.alphabet = m.alphabet;
    db->view.T = m.thr_log10; db->view.P = m.thr; db->view.convert_uo = m.convert_uo;
    db->view.soa = indexed ? 1u : 0u;
    db->view.mono = mono ? 1u : 0u;
    db->view.winspec = (db->windowed || db->has_pos) ? db->d_winspec : nullptr;
    db->view.win_w = db->windowed ? db->wp.W : 0u;
    db->view.n_win = db->windowed ? db->wp.n_win : 0u;
}

static int check_launchable(const rk_db *db);

// ---- k-mer -> row descriptor table, keys given in ascending dense-index order through the accessors ----
// DIRECT  : compact blocks, 16 bytes per 12 consecutive k-mers {u32 first row unit, 12 x u8 units per row}: 1.33 bytes
//           per k-mer (1.4 MiB at k=10) -- or per 24 k-mers with 4-bit unit counts (see `nib` below) -- small enough to live in the XCD L2s, one dwordx4 gather per probe; a row's
//           offset is the block base plus a byte prefix sum.  Needs rows of <= 255 units (4080 entries) and a blob
//           of < 2^32 units (512 GiB); otherwise DIRECT falls back to DIRECT8.
// DIRECT8 : one 8-byte descriptor per k-mer.
// HASH    : open addressing, linear probing, 16-byte slots {key+1, descriptor}, load <= 0.5.
template <class FDense, class FDesc, class FCode>
static int build_table(uint32_t &mode, uint64_t space, uint64_t n_keys, bool indexed, uint64_t max_units, uint64_t blob_units,
                       FDense dense_of, FDesc desc_of, FCode code_of, std::vector<uint64_t> &table, uint64_t &slots, uint64_t &hash_mask,
                       bool &nib) {
    if (mode == RK_TABLE_DIRECT && (max_units > 255 || blob_units >= (1ull << 32) || indexed)) mode = RK_TABLE_DIRECT8;
    slots = 0; hash_mask = 0;
    // 4-bit unit counts (24 k-mers per block, 0.67 bytes per k-mer) whenever no row exceeds 15 units = 240 entries: half the table
    // means its lines are re-touched twice as often and survive the rows streaming through the same L2 sets (C2: 15 of a read's 141
    // probes missed the L2 with the byte form)
    static const bool bytes_only = rk_knob("RK_COMPACT_BYTES") != nullptr;  // developer knob: A/B against the byte form
    // ... where rows stream at all: with fewer than one row unit per two k-mer codes (C4: 0.13) the probes outnumber the row lines,
    // nothing evicts the table and the longer decode is all that is left (C4: 5.37e8 against 5.52e8 reads/s)
    static const bool nibbles_always = rk_knob("RK_COMPACT_NIBBLES") != nullptr;  // developer knob: the half-size form whatever the density
    nib = mode == RK_TABLE_DIRECT && max_units <= 15 && (2 * blob_units >= space || nibbles_always) && !bytes_only;
    try {
        if (mode == RK_TABLE_DIRECT) {
            slots = space;
            const uint64_t per = nib ? 2 * COMPACT_KMERS : COMPACT_KMERS;
            const uint64_t n_blocks = (space + per - 1) / per;
            table.assign(n_blocks * 2, 0);
            unsigned char *tb = (unsigned char *)table.data();
            uint64_t next_unit = 1, ki = 0;
            for (uint64_t blk = 0; blk < n_blocks; blk++) {
                const uint32_t base32 = (uint32_t)next_unit;
                memcpy(tb + blk * 16, &base32, 4);
                while (ki < n_keys && dense_of(ki) / per == blk) {
                    const uint64_t units = ((uint32_t)desc_of(ki) & DESC_LEN_MASK) / ROW_UNIT;
                    const uint64_t j = dense_of(ki) % per;
                    if (nib) tb[blk * 16 + 4 + j / 2] |= (unsigned char)(units << (4 * (j & 1)));
                    else tb[blk * 16 + 4 + j] = (unsigned char)units;
                    next_unit += units;
                    ki++;
                }
            }
        } else if (mode == RK_TABLE_DIRECT8) {
            slots = space;
            table.assign(slots, 0);
            for (uint64_t i = 0; i < n_keys; i++) table[dense_of(i)] = desc_of(i);
        } else {
            slots = 16;
            while (slots < 2 * n_keys) slots <<= 1;
            hash_mask = slots - 1;
            table.assign(slots * 2, 0);
            for (uint64_t i = 0; i < n_keys; i++) {
                const uint64_t code = code_of(i);
                uint64_t h = host_mix64(code) & hash_mask;
                while (table[2 * h]) h = (h + 1) & hash_mask;
                table[2 * h] = code + 1;
                table[2 * h + 1] = desc_of(i);
            }
        }
    } catch (const std::bad_alloc &) {
        return fail(RK_ERR_NOMEM, "rk_db_create: host OOM building the k-mer table");
    }
    return RK_OK;
}

// Validation + host-side construction of the HBM image (no HIP call in here: rk_db_validate runs it without a device).
struct DbImage {
    uint32_t mode = 0, bits = 0, max_len = 0;
    bool indexed = false;
    bool mono = true;  // every score >= thr_log10 (true of every database RAPPAS builds: words below the threshold are not stored)
    uint64_t n_keys = 0, n_entries = 0, blob_bytes = 0, slots = 0, hash_mask = 0;
    std::vector<Entry> blob;
    std::vector<uint64_t> table;
    bool nib = false;                   // compact table in its 4-bit form
    bool windowed = false;              // place_packed16w_kernel can serve this image
    bool has_pos = false;               // not windowed, but winspec holds the rows' position in the tree (64 ranges): the key reads are grouped by
    WindowPlan wp;
    std::vector<unsigned char> winspec;  // [sigma^k]
};

static int build_image(const rk_db_desc *d, DbImage &img) {
    if (!d) return fail(RK_ERR_INVALID, "rk_db_create: null argument");
    if (d->alphabet != RK_ALPHABET_DNA && d->alphabet != RK_ALPHABET_AA)
        return fail(RK_ERR_INVALID, "rk_db_create: alphabet must be 4 (DNA) or 20 (AA), got %u", d->alphabet);
    const uint32_t bits = d->alphabet == RK_ALPHABET_DNA ? 2 : 5;
    // DNA: 2 bits per base in a 64-bit code; from k = 16 on the reference allows two ambiguity codes per k-mer
    // (maxAmbigPerMer = floor(k^(1/4)), AmbigSequenceKnife.java:95), three only from k = 81.  AA: 5 bits per residue.
    const uint32_t kmax = d->alphabet == RK_ALPHABET_DNA ? 31 : 12;
    if (d->k < 2 || d->k > kmax)
        return fail(RK_ERR_UNSUPPORTED, "rk_db_create: k=%u outside supported range 2..%u for this alphabet", d->k, kmax);
    if (d->n_branches < 1 || d->n_branches > 65535)
        return fail(RK_ERR_INVALID, "rk_db_create: n_branches=%u must be in 1..65535 (branch ids are 16-bit)", d->n_branches);
    if (!std::isfinite(d->thr_log10) || !std::isfinite(d->thr))
        return fail(RK_ERR_INVALID, "rk_db_create: thresholds must be finite");
    if (d->n_keys && (!d->key_codes || !d->row_offsets)) return fail(RK_ERR_INVALID, "rk_db_create: null key arrays");
    const uint64_t n_keys = d->n_keys;
    const uint64_t n_entries = n_keys ? d->row_offsets[n_keys] : 0;
    if (n_entries && (!d->branch_ids || !d->scores)) return fail(RK_ERR_INVALID, "rk_db_create: null entry arrays");
    if (n_keys && d->row_offsets[0] != 0) return fail(RK_ERR_INVALID, "rk_db_create: row_offsets[0] must be 0");

    img.bits = bits; img.n_keys = n_keys; img.n_entries = n_entries;
    // ---- table mode ----
    uint64_t space = 0;
    const bool space_ok = ipow_fits(d->alphabet, d->k, 1ull << 40, space);
    uint32_t &mode = img.mode;
    mode = d->table_mode;
    if (mode == RK_TABLE_AUTO) {
        // Direct addressing whenever all sigma^k codes fit 2^28 slots; DIRECT = compact 2-byte-per-k-mer blocks that
        // stay in the XCD L2s (measured on C2: 2.36e8 reads/s vs 2.18e8 with 8-byte descriptors, whose probes push the
        // Infinity Fabric to its ~6e10 requests/s ceiling -- scripts/ubench/gather_rate.hip).
        mode = (space_ok && space <= (1ull << 28)) ? RK_TABLE_DIRECT : RK_TABLE_HASH;
    }
    if ((mode == RK_TABLE_DIRECT || mode == RK_TABLE_DIRECT8) && !(space_ok && space <= (1ull << 31)))
        return fail(RK_ERR_UNSUPPORTED, "rk_db_create: direct table needs sigma^k <= 2^31 slots");
    if (mode != RK_TABLE_DIRECT && mode != RK_TABLE_DIRECT8 && mode != RK_TABLE_HASH)
        return fail(RK_ERR_INVALID, "rk_db_create: bad table_mode %u", mode);

    // ---- validate keys; rows are laid out in dense-index order of their k-mer ----
    auto code_ok = [&](uint64_t code, uint64_t &dense) -> bool {
        if (bits == 2) {
            if (d->k * 2 < 64 && (code >> (2 * d->k))) return false;
            dense = code;
            return true;
        }
        if (d->k * 5 < 64 && (code >> (5 * d->k))) return false;
        uint64_t idx = 0, pw = 1;
        for (uint32_t i = 0; i < d->k; i++) {
            uint64_t dig = (code >> (5 * i)) & 31;
            if (dig >= 20) return false;
            idx += dig * pw;
            pw *= 20;
        }
        dense = idx;
        return true;
    };
    std::vector<std::pair<uint64_t, uint64_t>> order;  // (dense index, key number)
    try { order.resize(n_keys); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create: host OOM"); }
    for (uint64_t r = 0; r < n_keys; r++) {
        uint64_t dense;
        if (!code_ok(d->key_codes[r], dense)) return fail(RK_ERR_INVALID, "rk_db_create: key %llu has an invalid k-mer code", (unsigned long long)r);
        order[r] = {dense, r};
    }
    std::sort(order.begin(), order.end());
    for (uint64_t i = 1; i < n_keys; i++)
        if (order[i].first == order[i - 1].first)
            return fail(RK_ERR_INVALID, "rk_db_create: duplicate k-mer code at key %llu", (unsigned long long)order[i].second);

    // ---- row blob: every row starts on a 128-byte unit and is padded to whole units (16 entries) with zero entries,
    //      so a row of n entries costs exactly ceil(n/16) aligned 128-byte requests; unit 0 is reserved (all padding) ----
    // Large trees (n_branches > RK_WG_MIN_BRANCHES): the score vector of one read fills most of a CU's LDS, so a whole
    // workgroup shares it and every wave owns a branch range (place_wg_kernel).  Rows are then sorted by branch id
    // and preceded by one 64-byte INDEX line: u16 split[i-1] = number of entries with branch < floor(i * n_branches / 32),
    // i = 1..32, so a wave finds its slice of a row with two 2-byte loads.  Descriptors still point at the first
    // entry line; the other kernels never look at the index line.
    // ... unless the rows are short: a workgroup's eight waves then each scan a sliver of every row, and one wave per read on
    // a few resident score vectors does better (measured, rows of ~13 entries: 12 001 branches 29 vs 20 Mreads/s; the
    // workgroup kernel is ahead again from ~16 000 branches on, where two score vectors fill a CU, and always for long rows).
    const double mean_len = n_keys ? (double)n_entries / (double)n_keys : 0.0;
    uint64_t slot_units = 1;
    uint32_t longest = 0;
    for (uint64_t r = 0; r < n_keys; r++) {
        const uint64_t len = d->row_offsets[r + 1] - d->row_offsets[r];
        slot_units += (len + ROW_UNIT - 1) / ROW_UNIT;
        if (len > longest) longest = (uint32_t)(len > 0xFFFFFFFFull ? 0xFFFFFFFFull : len);
    }
    const ImageKind kind = image_kind(d->n_branches, bits, d->table_mode, space, space_ok, slot_units, longest, mean_len);
    const bool indexed = kind.indexed;
    img.indexed = indexed;
    std::vector<uint64_t> desc(n_keys);  // by key number
    WindowPlan wp;
    const bool want_windows = kind.windowable && window_plan(d->n_branches, bits, space ? (double)slot_units / (double)space : 0.0, mean_len / ROW_UNIT + 0.5, wp);
    // Images the dense kernels serve get the same byte per k-mer code with the tree cut into 64 equal ranges: not for any window
    // -- the kernels hold the whole score vector -- but as the position a batch's reads are grouped by (reads of one clade read the
    // same rows: taken together they find them in the L2; scripts/clade_sorted_probe.py: 283 -> 406 Mreads/s on C2's shape)
    const bool want_pos = !want_windows && !indexed && space_ok && space <= (1ull << 26);
    const uint32_t key_w = want_windows ? wp.W : std::max<uint32_t>(1u, (d->n_branches + 63u) / 64u);
    std::vector<unsigned char> ws_by_key;  // winspec_byte(first window, last window) of every row
    if (want_windows || want_pos) {
        try { ws_by_key.assign(n_keys, 0); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create: host OOM"); }
    }
    // slot-offset images: 128-byte units (ROW_UNIT = 16 entries), so a chunk of 16 entries is ONE aligned 128-byte request;
    // raw-id (indexed) images: 64-byte units
    const uint64_t unit_bytes = indexed ? 64 : ROW_UNIT * 8;
    uint64_t blob_units = 1;  // unit 0 is reserved (padding pattern)
    uint32_t max_len = 0;
    uint64_t max_units = 0;
    for (uint64_t i = 0; i < n_keys; i++) {
        const uint64_t r = order[i].second;
        uint64_t b = d->row_offsets[r], e = d->row_offsets[r + 1];
        if (e < b) return fail(RK_ERR_INVALID, "rk_db_create: row_offsets not monotone at key %llu", (unsigned long long)r);
        uint64_t len = e - b;
        if (len == 0) return fail(RK_ERR_INVALID, "rk_db_create: key %llu has an empty row", (unsigned long long)r);
        if (len > d->n_branches)
            return fail(RK_ERR_INVALID, "rk_db_create: row %llu has %llu entries (> n_branches)", (unsigned long long)r, (unsigned long long)len);
        if (len > max_len) max_len = (uint32_t)len;
        uint64_t units = (len + ROW_UNIT - 1) / ROW_UNIT;  // 128-byte units of 16 {slot offset, score} entries
        uint64_t lenp = units * ROW_UNIT;
        if (indexed) {
            // [index line][u16 branch[lenp]][f32 score[lenp]], lenp a multiple of 32 so that the row is whole lines
            blob_units += 1;
            lenp = (len + 31) / 32 * 32;
            units = lenp * 6 / 64;
        }
        if (units > max_units) max_units = units;
        desc[r] = ((blob_units * (unit_bytes / 8)) << DESC_LEN_BITS) | lenp;
        blob_units += units;
    }
    const uint64_t blob_bytes = blob_units * unit_bytes;
    if ((blob_bytes >> 3) >= (1ull << 40)) return fail(RK_ERR_UNSUPPORTED, "rk_db_create: row blob exceeds 8 TiB");
    std::vector<Entry> &blob = img.blob;
    // padding / reserved line 0: raw-id images (large trees) skip on 0xFFFF, slot-offset images update scratch slot 0
    try { blob.assign(blob_bytes / 8, indexed ? Entry{0xFFFFFFFFu, 0.0f} : Entry{0u, 0.0f}); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create: host OOM for %llu-byte row blob", (unsigned long long)blob_bytes); }
    {
        // validation + fill, rows are independent: a few host threads over contiguous key ranges (6e8 entries took 14 s on one)
        struct RowErr { int kind = 0; uint64_t r = ~0ull; uint32_t x = 0; };
        auto fill = [&](uint64_t r_lo, uint64_t r_hi, RowErr &err, bool &mono) {
            std::vector<uint32_t> stamp(d->n_branches, 0xFFFFFFFFu);
            std::vector<Entry> tmp;
            for (uint64_t r = r_lo; r < r_hi; r++) {
                uint64_t b = d->row_offsets[r], len = d->row_offsets[r + 1] - b;
                Entry *ep = blob.data() + (desc[r] >> DESC_LEN_BITS);
                if (indexed) { tmp.resize(len); ep = tmp.data(); }
                uint32_t xmin = 0xFFFFu, xmax = 0;
                for (uint64_t i = 0; i < len; i++) {
                    uint16_t x = d->branch_ids[b + i];
                    xmin = x < xmin ? x : xmin;
                    xmax = x > xmax ? x : xmax;
                    float v = d->scores[b + i];
                    if (x >= d->n_branches) { err = {1, r, x}; return; }
                    if (stamp[x] == (uint32_t)r) { err = {2, r, x}; return; }
                    stamp[x] = (uint32_t)r;
                    if (!std::isfinite(v)) { err = {3, r, x}; return; }
                    if (!(v >= d->thr_log10)) mono = false;
                    ep[i].branch = indexed ? (uint32_t)x : ((uint32_t)x + 1u) * 4u;  // raw id (sorted, SoA below) | slot byte offset
                    ep[i].score = v;
                }
                if (want_windows || want_pos) ws_by_key[r] = winspec_byte(xmin / key_w, xmax / key_w);
                if (indexed) {
                    std::sort(ep, ep + len, [](const Entry &p, const Entry &q) { return p.branch < q.branch; });
                    unsigned char *row = (unsigned char *)(blob.data() + (desc[r] >> DESC_LEN_BITS));
                    const uint64_t lenp = (uint32_t)desc[r] & DESC_LEN_MASK;
                    uint16_t *split = (uint16_t *)(row - 64);  // the 64-byte line in front of the row
                    uint16_t *bp = (uint16_t *)row;
                    float *sp = (float *)(row + 2 * lenp);
                    uint64_t e = 0;
                    for (uint32_t i = 1; i <= 32; i++) {
                        const uint32_t bound = (uint32_t)(((uint64_t)i * d->n_branches) / 32);
                        while (e < len && ep[e].branch < bound) e++;
                        split[i - 1] = (uint16_t)e;
                    }
                    for (uint64_t i = 0; i < lenp; i++) {
                        bp[i] = i < len ? (uint16_t)ep[i].branch : (uint16_t)0xFFFFu;  // padding = skip entries
                        sp[i] = i < len ? ep[i].score : 0.0f;
                    }
                }
            }
        };
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned T = n_entries > (1u << 22) ? std::max(1u, std::min(hw ? hw : 1u, 16u)) : 1u;
        std::vector<RowErr> errs(T);
        std::vector<char> monos(T, 1);
        std::vector<std::thread> th;
        // ranges of (roughly) equal entry counts
        std::vector<uint64_t> cut(T + 1, n_keys);
        cut[0] = 0;
        for (unsigned t = 1; t < T; t++) {
            const uint64_t want = n_entries * t / T;
            cut[t] = (uint64_t)(std::lower_bound(d->row_offsets, d->row_offsets + n_keys, want) - d->row_offsets);
        }
        for (unsigned t = 0; t < T; t++) {
            auto job = [&, t]() { bool mono = true; fill(cut[t], cut[t + 1], errs[t], mono); monos[t] = mono ? 1 : 0; };
            if (t + 1 < T) th.emplace_back(job); else job();
        }
        for (std::thread &x : th) x.join();
        const RowErr *first = nullptr;
        for (const RowErr &e : errs)
            if (e.kind && (!first || e.r < first->r)) first = &e;
        if (first) {
            if (first->kind == 1) return fail(RK_ERR_INVALID, "rk_db_create: branch id %u >= n_branches in row %llu", first->x, (unsigned long long)first->r);
            if (first->kind == 2) return fail(RK_ERR_INVALID, "rk_db_create: branch id %u repeated inside row %llu", first->x, (unsigned long long)first->r);
            return fail(RK_ERR_INVALID, "rk_db_create: non-finite score in row %llu", (unsigned long long)first->r);
        }
        for (char mflag : monos)
            if (!mflag) img.mono = false;
    }

    // ---- table ----
    {
        int rc = build_table(img.mode, space, n_keys, indexed, max_units, blob_units,
                             [&](uint64_t i) { return order[i].first; }, [&](uint64_t i) { return desc[order[i].second]; },
                             [&](uint64_t i) { return d->key_codes[order[i].second]; }, img.table, img.slots, img.hash_mask, img.nib);
        if (rc) return rc;
    }

    img.blob_bytes = blob_bytes;
    img.max_len = max_len;
    // windows need the compact table (rows of <= 255 units) and 32-bit row offsets
    if (want_windows && img.mode == RK_TABLE_DIRECT && blob_bytes < RK_WINDOW_MAX_BLOB) {
        try { img.winspec.assign(space, 0); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create: host OOM"); }
        for (uint64_t i = 0; i < n_keys; i++) img.winspec[order[i].first] = ws_by_key[order[i].second];
        img.windowed = true;
        img.wp = wp;
    } else if ((want_windows || want_pos) && (img.mode == RK_TABLE_DIRECT || img.mode == RK_TABLE_DIRECT8)) {
        try { img.winspec.assign(space, 0); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create: host OOM"); }
        for (uint64_t i = 0; i < n_keys; i++) img.winspec[order[i].first] = ws_by_key[order[i].second];
        img.has_pos = true;
    }
    return RK_OK;
}

extern "C" int rk_db_validate(const rk_db_desc *d, rk_db_info *info) {
    RK_GUARD_BEGIN
    DbImage img;
    int rc = build_image(d, img);
    if (rc) return rc;
    if (info) {
        memset(info, 0, sizeof(*info));
        info->alphabet = d->alphabet; info->k = d->k; info->n_branches = d->n_branches; info->table_mode = img.mode;
        info->thr_log10 = d->thr_log10; info->thr = d->thr; info->n_keys = img.n_keys; info->n_entries = img.n_entries;
        info->table_slots = img.slots; info->table_bytes = img.table.size() * sizeof(uint64_t); info->rows_bytes = img.blob_bytes;
        info->bits_per_symbol = img.bits; info->max_row_len = img.max_len; info->device = -1;
    }
    return RK_OK;
    RK_GUARD_END("rk_db_validate")
}

extern "C" int rk_db_create(const rk_db_desc *d, rk_db **out) {
    RK_GUARD_BEGIN
    if (!d || !out) return fail(RK_ERR_INVALID, "rk_db_create: null argument");
    *out = nullptr;
    DbImage img;
    {
        int rc = build_image(d, img);  // argument errors are reported before the device is looked at
        if (rc) return rc;
    }
    const uint32_t mode = img.mode, max_len = img.max_len;
    const bool indexed = img.indexed;
    const uint64_t n_keys = img.n_keys, n_entries = img.n_entries, blob_bytes = img.blob_bytes, slots = img.slots, hash_mask = img.hash_mask;
    std::vector<Entry> &blob = img.blob;
    std::vector<uint64_t> &table = img.table;

    // ---- upload ----
    rk_db *db = nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    DbMeta meta{d->alphabet, d->convert_uo, d->k, d->n_branches, d->thr_log10, d->thr};
    int rc = open_db(meta, d->device, &db);
    if (rc) return rc;
#define DB_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int c_ = fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rk_db_destroy(db);                                                                    \
            return c_;                                                                            \
        }                                                                                         \
    } while (0)
    const size_t table_bytes = table.size() * sizeof(uint64_t);
    DB_TRY(hipMalloc(&db->d_table, table_bytes ? table_bytes : 8));
    DB_TRY(hipMalloc(&db->d_rows, blob_bytes));
    if (table_bytes) DB_TRY(hipMemcpy(db->d_table, table.data(), table_bytes, hipMemcpyHostToDevice));
    DB_TRY(hipMemcpy(db->d_rows, blob.data(), blob_bytes, hipMemcpyHostToDevice));
    if (img.windowed || img.has_pos) {
        DB_TRY(hipMalloc((void **)&db->d_winspec, img.winspec.size()));
        DB_TRY(hipMemcpy(db->d_winspec, img.winspec.data(), img.winspec.size(), hipMemcpyHostToDevice));
        db->windowed = img.windowed;
        db->has_pos = img.has_pos;
        if (img.windowed) db->wp = img.wp;
    }
#undef DB_TRY
    db->compact_nib = img.nib;
    finish_db(db, meta, mode, indexed, img.mono, n_keys, n_entries, slots, hash_mask, table_bytes, blob_bytes, max_len);
    rc = check_launchable(db);  // a tree whose score vector no kernel geometry can hold is refused here, not at the first batch
    if (rc) { rk_db_destroy(db); return rc; }
    *out = db;
    return RK_OK;
    RK_GUARD_END("rk_db_create")
}

// A second handle of the same database on another (or the same) device, copied device to device: the image is not rebuilt and
// nothing goes back through the host (a C5-class image is 200 GB; xGMI moves it in seconds, the host could not even hold it).
extern "C" int rk_db_clone(const rk_db *src, int32_t device, rk_db **out) {
    RK_GUARD_BEGIN
    if (!src || !out) return fail(RK_ERR_INVALID, "rk_db_clone: null argument");
    *out = nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    DbMeta meta{src->info.alphabet, src->convert_uo, src->info.k, src->info.n_branches, src->info.thr_log10, src->info.thr};
    rk_db *db = nullptr;
    int rc = open_db(meta, device, &db);
    if (rc) return rc;
#define CL_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int c_ = fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rk_db_destroy(db);                                                                    \
            return c_;                                                                            \
        }                                                                                         \
    } while (0)
    const size_t table_bytes = src->info.table_bytes, blob_bytes = src->info.rows_bytes;
    const int sdev = src->info.device;
    auto copy = [&](void *dst, const void *from, size_t bytes) {
        return device == sdev ? hipMemcpy(dst, from, bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, device, from, sdev, bytes);
    };
    CL_TRY(hipMalloc(&db->d_table, table_bytes ? table_bytes : 8));
    CL_TRY(hipMalloc(&db->d_rows, blob_bytes));
    if (table_bytes) CL_TRY(copy(db->d_table, src->d_table, table_bytes));
    CL_TRY(copy(db->d_rows, src->d_rows, blob_bytes));
    if (src->windowed || src->has_pos) {
        uint64_t space = 0;
        (void)ipow_fits(src->info.alphabet, src->info.k, 1ull << 40, space);
        CL_TRY(hipMalloc((void **)&db->d_winspec, space));
        CL_TRY(copy(db->d_winspec, src->d_winspec, space));
        db->windowed = src->windowed;
        db->has_pos = src->has_pos;
        db->wp = src->wp;
    }
    CL_TRY(hipDeviceSynchronize());
#undef CL_TRY
    db->compact_nib = src->compact_nib;
    finish_db(db, meta, src->info.table_mode, src->indexed, src->view.mono != 0, src->info.n_keys, src->info.n_entries, src->info.table_slots,
              src->view.hash_mask, table_bytes, blob_bytes, src->info.max_row_len);
    db->lanes_per_read = src->lanes_per_read;
    *out = db;
    return RK_OK;
    RK_GUARD_END("rk_db_clone")
}

extern "C" int rk_db_get_info(const rk_db *db, rk_db_info *info) {
    if (!db || !info) return fail(RK_ERR_INVALID, "rk_db_get_info: null argument");
    *info = db->info;
    return RK_OK;
}

static int check_params(const rk_params *p);

// one host thread per device handle; contiguous shards; see include/rappas_place.h
extern "C" int rk_place_batch_multi(rk_db *const *dbs, uint32_t n_dbs, const rk_params *p, uint64_t n_reads,
                                    const uint8_t *seq_ascii, const uint64_t *seq_off, rk_result *out, rk_counters *counters) {
    if (!dbs || n_dbs == 0 || !out) return fail(RK_ERR_INVALID, "rk_place_batch_multi: null argument");
    for (uint32_t g = 0; g < n_dbs; g++)
        if (!dbs[g]) return fail(RK_ERR_INVALID, "rk_place_batch_multi: dbs[%u] is null", g);
    int rc = check_params(p);
    if (rc) return rc;
    if (n_dbs == 1 || n_reads == 0) return rk_place_batch(dbs[0], p, n_reads, seq_ascii, seq_off, out, counters);
    if (!seq_ascii || !seq_off) return fail(RK_ERR_INVALID, "rk_place_batch_multi: null reads");
    if (!out->n_rows || !out->branch || !out->score || !out->lwr || !out->flags) return fail(RK_ERR_INVALID, "rk_place_batch_multi: null result array");
    const uint32_t K = p->keep_at_most;
    RK_GUARD_BEGIN
    std::vector<int> codes(n_dbs, RK_OK);
    std::vector<std::string> msgs(n_dbs);
    std::vector<rk_counters> cts(n_dbs);
    // developer / test knob: the first attempt of this shard reports a device failure (exercises the re-queue below)
#ifdef RK_DEV_KNOBS
    const int inject = rk_knob("RK_TEST_FAIL_SHARD") ? atoi(rk_knob("RK_TEST_FAIL_SHARD")) : -1;
#endif
    auto run_shard = [&](uint32_t g, uint32_t on, bool first_attempt) {  // shard g of the batch on handle `on`, in the calling thread
        const uint64_t lo = n_reads * g / n_dbs, hi = n_reads * (g + 1) / n_dbs;
        cts[g] = rk_counters{};
        codes[g] = RK_OK;
        if (hi == lo) return;
#ifdef RK_DEV_KNOBS
        if (first_attempt && inject == (int)g) {
            codes[g] = RK_ERR_HIP;
            msgs[g] = "injected failure (RK_TEST_FAIL_SHARD)";
            return;
        }
#else
        (void)first_attempt;
#endif
        rk_result r{out->n_rows + lo, out->branch + lo * K, out->score + lo * K, out->lwr + lo * K, out->flags + lo};
        tl_concurrent_calls = first_attempt ? n_dbs : 1u;  // (this shard's thread: the host threads its call starts are 1 / n_dbs of the budget)
        codes[g] = rk_place_batch(dbs[on], p, hi - lo, seq_ascii, seq_off + lo, &r, &cts[g]);
        tl_concurrent_calls = 1;
        try {
            if (codes[g] != RK_OK) msgs[g] = rk_last_error();  // the message lives in this thread: hand it over
        } catch (...) {  // (nothing may leave a thread's function: std::terminate would take the hosting process down)
        }
    };
    struct JoinAll {  // joined on every way out of the scope, a throwing emplace_back included
        std::vector<std::thread> v;
        ~JoinAll() { for (std::thread &t : v) if (t.joinable()) t.join(); }
    };
    {
        JoinAll workers;
        workers.v.reserve(n_dbs);
        for (uint32_t g = 0; g < n_dbs; g++) workers.v.emplace_back([&, g]() { run_shard(g, g, true); });
    }
    // A shard whose device failed (per-GPU failure => shard re-queued on another GPU) is placed again on
    // the handles that did finish, one after the other, each attempt in a fresh host thread; the process is never restarted.
    // Argument errors (RK_ERR_INVALID / RK_ERR_UNSUPPORTED) would fail anywhere and are not retried.
    std::vector<char> healthy(n_dbs);
    for (uint32_t g = 0; g < n_dbs; g++) healthy[g] = codes[g] == RK_OK;
    std::string note;
    for (uint32_t g = 0; g < n_dbs; g++) {
        if (codes[g] == RK_OK || codes[g] == RK_ERR_INVALID || codes[g] == RK_ERR_UNSUPPORTED) continue;
        const std::string first_msg = msgs[g];
        const int first_code = codes[g];
        for (uint32_t h = 0; h < n_dbs && codes[g] != RK_OK; h++) {
            if (!healthy[h]) continue;
            {
                JoinAll one;
                one.v.emplace_back([&, g, h]() { run_shard(g, h, false); });
            }
            if (codes[g] == RK_OK) {
                char buf[256];
                snprintf(buf, sizeof(buf), "shard %u failed on device %d (%d: %.120s) and was placed on device %d; ", g, dbs[g]->info.device,
                         first_code, first_msg.c_str(), dbs[h]->info.device);
                note += buf;
            }
        }
        if (codes[g] != RK_OK) { codes[g] = first_code; msgs[g] = first_msg; }
    }
    rk_counters total{};
    for (uint32_t g = 0; g < n_dbs; g++) {
        if (codes[g] != RK_OK) return fail(codes[g], "rk_place_batch_multi: shard %u (device %d): %s", g, dbs[g]->info.device, msgs[g].c_str());
        total.reads += cts[g].reads; total.placed += cts[g].placed; total.unplaced += cts[g].unplaced;
        total.bad_char += cts[g].bad_char; total.too_short += cts[g].too_short; total.ambiguous += cts[g].ambiguous;
    }
    if (counters) *counters = total;
    // success, but the caller can still learn which device dropped out: rk_last_error() carries the note (empty otherwise)
    (void)fail(RK_OK, "%s", note.c_str());
    return RK_OK;
    RK_GUARD_END("rk_place_batch_multi")
}

extern "C" void *rk_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)fail(e == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "rk_host_alloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

extern "C" void rk_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" uint32_t rk_packed_words(const rk_db *db, uint32_t max_len) {
    if (!db) return 0;
    uint64_t bits = (uint64_t)max_len * db->info.bits_per_symbol;
    uint32_t w = (uint32_t)((bits + 31) / 32);
    return w ? w : 1;
}

extern "C" int rk_set_lanes_per_read(rk_db *db, uint32_t lanes) {
    if (!db) return fail(RK_ERR_INVALID, "rk_set_lanes_per_read: null db");
    if (lanes != 0 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64)
        return fail(RK_ERR_INVALID, "rk_set_lanes_per_read: lanes must be 0 (auto), 8, 16, 32 or 64");
    if (lanes != 0 && db->indexed)
        return fail(RK_ERR_UNSUPPORTED, "rk_set_lanes_per_read: this database uses the large-tree image (n_branches > %u, long rows): always the workgroup-per-read kernel", RK_WG_MIN_BRANCHES);
    db->lanes_per_read = lanes;
    return RK_OK;
}

// ------------------------------------------------------------------------------------------------
// launch geometry
// ------------------------------------------------------------------------------------------------
struct Geometry {
    uint32_t G, NG, s_stride, list_cap, pu;
    size_t lds_per_wave;
    uint32_t waves_per_cu;
};

// PU*G >= 144 positions per probe batch.  Amino acids through the 16-lane kernels: a packed record of <= 16 words holds <= 102
// residues, so seven rounds of sixteen are a whole read (nine made C4 look up 48 k-mers per read that do not exist)
static uint32_t probe_unroll(uint32_t G, uint32_t bits) { return G <= 16 ? (bits == 5 && G == 16 ? 7 : 9) : (G == 32 ? 5 : 3); }

static int choose_geometry(const rk_db *db, uint32_t keep_at_most, Geometry &g) {
    const uint32_t nb = db->info.n_branches;
    const uint32_t s_stride = (nb + 4) & ~3u;  // >= nb + 1 (word nb is the scratch slot of apply_entry), multiple of 4 (b128 scans)
    const size_t target = db->lds_per_cu / 8;  // aim for >= 8 waves per CU
    const size_t fixed = (size_t)s_stride * 4;  // per read, besides the hit list
    auto bytes_for = [&](uint32_t G, uint32_t cap) { return (size_t)(64 / G) * (fixed + (size_t)cap * 8); };
    uint32_t G = db->lanes_per_read;
    if (G == 0) {
        // Throughput follows the reads in flight per CU (LDS capacity / score-vector size) and, at equal reads in flight, prefers
        // narrower groups as long as enough waves remain to hide latency.  Measured on C2-like DBs (scripts/tree_size_sweep.py):
        // 999 branches: 16 lanes (8 waves) 335 Mreads/s; 1300 / 1500: 16 lanes (6 waves) 219 / 214 vs 32 lanes (11 / 10 waves) 156 / 141;
        // 1999: 32 lanes (8 waves) 199 vs 16 lanes (4 waves) 165 vs 64 lanes 120; 3999: 32 lanes (4 waves) 88 vs 64 lanes (9 waves) 62;
        // 7999: 64 lanes (4 waves) 46 vs 32 lanes (2 waves) 35.  (Between 1 117 and 16 000 branches these dense geometries only serve
        // what the windowed kernel does not take: records of more than 16 words, a forced lane width.)
        G = 64;
        if (keep_at_most <= 16 && bytes_for(16, 16 + 3 * RK_RING + 40) <= db->lds_per_cu / 6) G = 16;
        else if (keep_at_most <= 32 && bytes_for(32, 32 + 3 * RK_RING + 40) <= db->lds_per_cu / 4) G = 32;
    }
    if (G < keep_at_most) return fail(RK_ERR_INVALID, "keep_at_most=%u needs lanes_per_read >= %u", keep_at_most, keep_at_most);
    const uint32_t NG = 64 / G, pu = probe_unroll(G, db->info.bits_per_symbol);
    const uint32_t min_cap = G + 3 * (G == 64 ? RK_RING64 : RK_RING) + 40;  // one sub-batch of rows + sentinel, ring slack, 16 winner slots + margin
    // list capacity: whatever is left of the per-wave LDS target, clamped to [min_cap, 256]
    size_t per_group_target = target / NG;
    uint32_t cap = min_cap;
    if (per_group_target > fixed + (size_t)min_cap * 8) {
        size_t c = (per_group_target - fixed) / 8;
        cap = (uint32_t)(c > 256 ? 256 : c);
        if (cap < min_cap) cap = min_cap;
    }
    if (const char *e = rk_knob("RK_LIST_CAP")) {  // developer knob: trade hit-list room for occupancy
        uint32_t v = (uint32_t)atoi(e);
        if (v >= min_cap && v <= 4096) cap = v;
    }
    cap &= ~1u;  // keeps every group's score vector 16-byte aligned
    g.G = G; g.NG = NG; g.s_stride = s_stride; g.list_cap = cap; g.pu = pu;
    g.lds_per_wave = bytes_for(G, cap);
    if (g.lds_per_wave > db->lds_per_cu)
        return fail(RK_ERR_UNSUPPORTED, "n_branches=%u needs %zu B of LDS per read, more than one CU has (%zu B)", nb, g.lds_per_wave, db->lds_per_cu);
    uint32_t w = (uint32_t)(db->lds_per_cu / g.lds_per_wave);
    g.waves_per_cu = w > 32 ? 32 : w;
    if (const char *e = rk_knob("RK_WAVES_PER_CU")) {  // developer knob for occupancy experiments
        uint32_t v = (uint32_t)atoi(e);
        if (v >= 1 && v < g.waves_per_cu) g.waves_per_cu = v;
    }
    return RK_OK;
}

// 16 lanes per read, direct table, 32-bit row offsets, packed record of <= 16 words: the tile-pipelined kernel
static bool use_pipelined16(const rk_db *db, const Geometry &g, const PlaceArgs &args) {
    static const bool off = rk_knob("RK_NO_PIPE") != nullptr;  // developer knob: A/B against place_packed_kernel
    return !off && g.G == 16 && db->info.table_mode != RK_TABLE_HASH && db->info.rows_bytes < ROWS_FIT32_LIMIT && args.words_per_read <= 16;
}

template <int G, int BITS, int TM, bool WIDE>
static int launch_variant(const rk_db *db, const Geometry &g, const PlaceArgs &args, hipStream_t stream) {
    constexpr int PU = G <= 16 ? (BITS == 5 && G == 16 ? 7 : 9) : (G == 32 ? 5 : 3);  // (= probe_unroll)
    constexpr int U = G == 64 ? RK_RING64 : RK_RING;
    auto kern = place_packed_kernel<G, BITS, TM, WIDE, U, PU>;
    if constexpr (G == 16 && !WIDE && TM != TM_HASH) {
        if (use_pipelined16(db, g, args)) {
            kern = place_packed16_kernel<BITS, TM, U, PU>;
            if constexpr (BITS == 5) {
                // every read of the batch has the same, known length and at most 96 k-mers (C4: 100 residues, k = 5): six rounds
                if (!args.lens && args.fixed_len >= db->info.k && args.fixed_len - db->info.k + 1 <= 96u) kern = place_packed16_kernel<BITS, TM, U, 6>;
            }
        }
    }
    const uint32_t wpb = db->waves_per_block;
    const size_t lds = g.lds_per_wave * wpb;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t n_tiles = (args.n_reads + g.NG - 1) / g.NG;
    uint64_t per_cu = 0;
    if (int rc = resident_blocks(kern, 64 * (int)wpb, lds, (g.waves_per_cu + wpb - 1) / wpb, per_cu)) return rc;
    uint64_t blocks = (uint64_t)db->cu_count * per_cu;
    const uint64_t need = (n_tiles + wpb - 1) / wpb;
    if (blocks > need) blocks = need;
    if (blocks == 0) return RK_OK;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, stream, args);
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

template <int G, int BITS, int TM>
static int launch_w(const rk_db *db, const Geometry &g, const PlaceArgs &a, hipStream_t s) {
    // 32-bit row offsets whenever the row blob is < 4 GiB
    return db->info.rows_bytes < ROWS_FIT32_LIMIT ? launch_variant<G, BITS, TM, false>(db, g, a, s)
                                              : launch_variant<G, BITS, TM, true>(db, g, a, s);
}
template <int G, int BITS>
static int launch_t(const rk_db *db, const Geometry &g, const PlaceArgs &a, hipStream_t s) {
    switch (db->info.table_mode) {
    case RK_TABLE_DIRECT: return launch_w<G, BITS, TM_COMPACT>(db, g, a, s);
    case RK_TABLE_DIRECT8: return launch_w<G, BITS, TM_DIRECT8>(db, g, a, s);
    default: return launch_w<G, BITS, TM_HASH>(db, g, a, s);
    }
}
template <int G>
static int launch_b(const rk_db *db, const Geometry &g, const PlaceArgs &a, hipStream_t s) {
    return db->info.bits_per_symbol == 2 ? launch_t<G, 2>(db, g, a, s) : launch_t<G, 5>(db, g, a, s);
}
// mid-size trees: the windowed kernel whenever the image carries window spans, nobody forced a lane-group width, the K best
// of the tree fit one 16-lane row (keep_at_most <= 16) and the packed record fits one word per lane
static bool use_windowed(const rk_db *db, uint32_t keep_at_most, uint32_t words_per_read) {
    static const bool off = rk_knob("RK_NO_WINDOW") != nullptr;  // developer knob: A/B against the dense kernels
    // (scripts/keep_at_most_sweep.py, windowed against dense, Mreads/s: 3 999 branches K = 9 / 12 / 16: 156 / 141 / 105 against 91 / 84 / 76)
    // Records of more than 16 words (the kernel then reads its k-mers from memory, and a long read is emitted in several window
    // ranges): ahead of the dense kernels while a read has fewer symbols than about a ninth of the tree's branches
    // (scripts/read_length_sweep.py, G k-mers/s windowed against dense: 3 999 branches 300 / 450 / 600 bp: 25.6 / 19.8 / 16.4 against
    // 18.3 / 20.6 / 22.0; 7 999: 450 / 600 / 1000 bp: 17.5 / 15.2 / 10.8 against 11.8 / 12.7 / 14.4; 15 999: 1000 bp 9.7 against 6.8)
    const uint64_t max_symbols = (uint64_t)words_per_read * 32 / db->info.bits_per_symbol;
    const bool fits = words_per_read <= 16 || max_symbols * 9 <= db->info.n_branches || db->info.n_branches > 16000;  // (25 001 branches: 10.2 against 3.3 at 1000 bp)
    return !off && db->windowed && db->lanes_per_read == 0 && keep_at_most <= 16 && fits;
}

// ---- scratch of a launch: owned by the handle, one grow-only block per stream a caller launches on.  Nothing is taken from (and no
//      attribute is set on) the device's default memory pool, which the hosting process -- a JVM, PyTorch -- shares.  Returns nullptr
//      when the block would have to grow and cannot right now (the stream is being captured into a graph, the device is out of
//      memory): the callers then do without it (batch order kept, place_packed16w_kernel alone).  Calls on ONE stream must not
//      overlap in time (include/rappas_place.h): growing the block waits for the stream's work and frees the old one ----
static void *launch_scratch(const rk_db *db, hipStream_t s, size_t bytes) {
    std::lock_guard<std::mutex> lock(db->scratch_mu);
    rk_db::LaunchScratch *b = nullptr;
    for (auto &x : db->scratch)
        if (x.s == s) { b = &x; break; }
    if (!b) {
        if (db->scratch.size() >= 16) {  // a caller that makes a stream per call: the least recently used block goes (its stream may be gone, so the whole device is waited for)
            auto lru = std::min_element(db->scratch.begin(), db->scratch.end(), [](const rk_db::LaunchScratch &x, const rk_db::LaunchScratch &y) { return x.used < y.used; });
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone || hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            if (lru->p) (void)hipFree(lru->p);
            db->scratch.erase(lru);
        }
        db->scratch.push_back({s, nullptr, 0, 0});
        b = &db->scratch.back();
    }
    b->used = ++db->scratch_clock;
    if (b->cap < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (b->p) {
            if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            (void)hipFree(b->p);
            b->p = nullptr;
            b->cap = 0;
        }
        const size_t want = ((bytes + bytes / 4) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (hipMalloc(&b->p, want) != hipSuccess || !b->p) {
            (void)hipGetLastError();
            b->p = nullptr;
            return nullptr;
        }
        b->cap = want;
    }
    return b->p;
}

// ---- tiles of reads that sit in the same part of the tree (rk_kernels.hip: retile_*): keys, counting sort, the order the kernels take
//      their tiles in (PlaceArgs::perm); and the marks of the tiles a first kernel hands to place_packed16w_kernel
//      (PlaceArgs::tile_marks, one byte per tile of four reads).  Without scratch, or below 32 768 reads (the pre-pass's launches
//      cost more than they can win), the batch keeps its order ----
struct TileOrder {
    unsigned char *marks = nullptr;  // zeroed by prepare() when asked for
    int prepare(const rk_db *db, PlaceArgs &a, hipStream_t s, bool want_marks) {
        a.perm = nullptr;
        a.keep_order = nullptr;
        a.tile_marks = nullptr;
        uint64_t retile_min = 32768;
        if (const char *e = rk_knob("RK_RETILE_MIN_READS")) retile_min = (uint64_t)atoll(e);  // developer / test knob (0 = always)
        // (images without a position byte per k-mer -- hashed tables, the large-tree image -- keep their order)
        const bool retile = db->view.winspec && a.n_reads >= retile_min && a.n_reads < (1ull << 32) && !rk_knob("RK_NO_RETILE");
        if (!retile && !want_marks) return RK_OK;
        const size_t n_tiles = (size_t)((a.n_reads + 3) / 4);
        const size_t marks_off = 1024, perm_off = marks_off + (want_marks ? ((n_tiles + 255) & ~(size_t)255) : 0);
        const size_t keys_off = perm_off + (retile ? (((size_t)a.n_reads * 4 + 255) & ~(size_t)255) : 0);
        // (the marked tiles as a list + the queue's two counters, in front of the marks: place_packed16w_kernel as the second launch)
        const bool want_list = want_marks && n_tiles < (1ull << 32);
        const size_t list_off = (keys_off + (retile ? a.n_reads : 0) + 255) & ~(size_t)255, total = list_off + (want_list ? n_tiles * 4 : 0);
        unsigned char *base = (unsigned char *)launch_scratch(db, s, total);
        if (!base) return RK_OK;
        if (want_marks) {
            marks = base + marks_off;
            HIP_TRY(hipMemsetAsync(marks - 8, 0, n_tiles + 8, s));
            a.tile_marks = marks;
            if (want_list) {
                a.marked_ctl = (uint32_t *)(marks - 8);
                a.marked_list = (uint32_t *)(base + list_off);
            }
        }
        if (!retile) return RK_OK;
        uint32_t *hist = (uint32_t *)base, *cursor = hist + 128, *perm = (uint32_t *)(base + perm_off);
        unsigned char *keys = base + keys_off;
        HIP_TRY(hipMemsetAsync(base, 0, marks_off - 8, s));  // (the last 8 bytes: the hand-over queue's counters, zeroed with the marks)
        const unsigned nblk = (unsigned)std::min<uint64_t>((a.n_reads + 255) / 256, 2048);  // (grid-stride: a batch that keeps its order ends 2 048 blocks, not a million threads)
        const unsigned sblk = (unsigned)((a.n_reads / 64 + 255) / 256 + 1);
        if (db->info.bits_per_symbol == 2) hipLaunchKernelGGL(retile_sample_kernel<2>, dim3(sblk), dim3(256), 0, s, a, hist);
        else hipLaunchKernelGGL(retile_sample_kernel<5>, dim3(sblk), dim3(256), 0, s, a, hist);
        // (a batch is "sparse" when its sampled k-mers have a row at most 1.3 times as often as a random read's: the share of the alphabet's
        //  k-mer codes that carry one)
        double space = 1.0;
        for (uint32_t i = 0; i < db->info.k; i++) space *= (double)db->info.alphabet;
        const double share = std::min(1.0, 1.3 * (double)db->info.n_keys / space);
        hipLaunchKernelGGL(retile_decide_kernel, dim3(1), dim3(64), 0, s, hist, (uint32_t)(share * 65536.0));
        if (db->info.bits_per_symbol == 2) hipLaunchKernelGGL(retile_key_kernel<2>, dim3(nblk), dim3(256), 0, s, a, keys, hist);
        else hipLaunchKernelGGL(retile_key_kernel<5>, dim3(nblk), dim3(256), 0, s, a, keys, hist);
        hipLaunchKernelGGL(retile_scan_kernel, dim3(1), dim3(64), 0, s, hist, cursor);
        hipLaunchKernelGGL(retile_scatter_kernel, dim3(nblk), dim3(256), 0, s, a.n_reads, (const unsigned char *)keys, (const uint32_t *)hist, cursor, perm);
        HIP_TRY(hipGetLastError());
        a.perm = perm;
        a.keep_order = hist + 65;  // (RETILE_BINS + 1)
        return RK_OK;
    }
};

// place_hash64_kernel's geometry: NS = 2 048 slots (16 KB) + a word per lane + a list of 320 items = 17 920 B per wave, nine waves per CU
#ifndef RK_HASH_LOGS
#define RK_HASH_LOGS 11
#endif
constexpr uint32_t RK_HASH_LOG_SLOTS = RK_HASH_LOGS, RK_HASH_MAIN_CAP = 320;
#ifndef RK_HASH_KEY_SLACK
#define RK_HASH_KEY_SLACK 48u  // slots kept free: a read's table takes NS - this many keys (the kernel counts a step's entries before it takes the step)
#endif
// From this many branches on place_hash64_kernel is ahead of place_packed16s_kernel, whose cost grows with the windows a tree is cut
// into (profiles/r04_hash_crossover.txt and DESIGN.md 4.1d; C2-like rows, Mreads/s hash / sorted-stream: uniform reads 19 999 branches
// 94 / 108, 28 001: 95 / 95, 39 999: 95 / 71, 65 535: 94 / 35; clade-shaped reads 82 / 119 at 28 001, 81 / 95 at 39 999, 82 / 47 at
// 65 535; 92 / 81 at 50 001, 84 / 81 at 55 001, 48 / 81 at 60 001).  Uniform reads cross at ~28 000 branches, clade-shaped ones at ~56 000: in between BOTH kernels are launched and the
// batch's shape -- what the re-tiling pre-pass found, on the device -- says which of them runs (PlaceArgs::only_if); a batch without
// the pre-pass (fewer than 32 768 reads) goes by the single rule in the middle.
#ifndef RK_HASH_MIN_BRANCHES_CLADE
#define RK_HASH_MIN_BRANCHES_CLADE 56000u
#endif
// amino acids (k = 5, 100 residues, C4-like rows: a quarter of the k-mers present, ~300 entries a read): the hash kernel runs 196 Mreads/s
// at any size, place_packed16s_kernel 231 / 205 / 177 / 119 / 51 at 9 001 / 12 001 / 15 999 / 33 001 / 65 535 branches -- they cross at
// ~13 500; reads cut from the sequence the k-mers come from 107 against 198 / 168 / 136 / 65 at 9 001 / 25 001 / 46 001 / 60 001: ~56 000
// as for DNA (profiles/r04_hash_crossover_aa.txt).
// The hash kernel's cost follows a read's row units, place_packed16s_kernel's the units AND the windows: through the two measured
// crossings -- 145 units a read (C2, 150 bp) at 28 000 branches, 33 (C4-like, 100 residues) at 13 500 -- the uniform crossing is taken as
// 130 branches per unit + 9 250 for other row densities and read lengths; batches too small for the pre-pass go by that + 8 000
// (36 000 for C2-like rows).  -DRK_HASH_MIN_BRANCHES_UNIFORM_FIXED=n replaces the fit by a constant.
// Reads that bring few row entries even when every k-mer of theirs has a row take the table of 1 024 slots (hash_small_table): sixteen waves
// per CU, half the reset and the scan.  Forced onto the protein sweep above (RK_HASH_SMALL_TABLE; its reads hit a quarter of their k-mers, a real
// read would overflow that table) it gives 310 Mreads/s on uniform reads and 170 on clade-shaped ones at every size, and the crossings move to ~2 600 branches
// (place_packed16s_kernel: 329 / 297 at 2 001 / 3 100) and ~24 000 (181 / 168 / 151 at 15 999 / 25 001 / 33 001).  One measured crossing only in
// this regime: 19 branches per unit + 2 000 (a tree of four windows: the table's reset and scan do not shrink with the read) passes through it.
// (clade-shaped batches through the 1 024-slot table with the large one behind it: DNA 96 - 100 Mreads/s at any size against place_packed16s_kernel's
//  128 / 106 / 95 / 92 / 84 / 48 at 19 999 / 33 001 / 46 001 / 50 001 / 55 001 / 60 001 branches -- they cross at ~42 000; amino acids 170 against
//  181 / 168 / 151 at 15 999 / 25 001 / 33 001: ~24 000.  Through both: 160 branches per row unit of a read + 18 700)
static uint32_t hash_min_clade_small(double est_units) {
#ifdef RK_HASH_MIN_BRANCHES_CLADE_SMALL
    (void)est_units;
    return RK_HASH_MIN_BRANCHES_CLADE_SMALL;
#else
    const double nb = 160.0 * est_units + 18700.0;
    return nb > 65535.0 ? 65535u : (uint32_t)nb;
#endif
}
static uint32_t hash_min_clade(bool small_table) { return small_table ? 24000u : RK_HASH_MIN_BRANCHES_CLADE; }
static uint32_t hash_min_uniform(double est_units, bool small_table) {
#ifdef RK_HASH_MIN_BRANCHES_UNIFORM_FIXED
    (void)est_units;
    return RK_HASH_MIN_BRANCHES_UNIFORM_FIXED;
#else
    const double nb = small_table ? 19.0 * est_units + 2000.0 : 130.0 * est_units + 9250.0;
    return nb > (double)hash_min_clade(small_table) ? hash_min_clade(small_table) : (uint32_t)nb;
#endif
}
static uint32_t hash_min_single(double est_units, bool small_table) {
    const uint32_t u = hash_min_uniform(est_units, small_table) + 8000u;
    return u > hash_min_clade(small_table) ? hash_min_clade(small_table) : u;
}
static bool hash_capable(const rk_db *db) {  // images whose tiles can go to place_hash64_kernel first
    if (rk_knob("RK_NO_HASH") || rk_knob("RK_NO_WSTREAM") || db->info.rows_bytes >= ROWS_FIT32_LIMIT) return false;
    return rk_knob("RK_HASH_ALWAYS") || db->wp.stream;
}
static bool hash_tree(const rk_db *db, double est_units, bool small_table) {  // ... by the single rule
    if (!hash_capable(db)) return false;
    return rk_knob("RK_HASH_ALWAYS") || db->info.n_branches > hash_min_single(est_units, small_table);
}
static uint32_t hash_key_limit(uint32_t log_slots);
// The table of 1 024 slots: est_units < 0 asks "in no case"; otherwise est_units holds the row ENTRIES of a read all of whose k-mers have a
// row (mean row length x its k-mers) -- what a read from an organism of the reference brings, four times the uniform estimate of C4-like
// rows -- and the small table is taken only when even that fits: a read that overflows costs a tile of place_packed16w_kernel.
static bool hash_small_table(double full_hit_entries) {
    if (rk_knob("RK_HASH_BIG_TABLE") || full_hit_entries < 0.0) return false;  // (developer knob: A/B)
    return rk_knob("RK_HASH_SMALL_TABLE") != nullptr || full_hit_entries <= 0.8 * hash_key_limit(RK_HASH_LOG_SLOTS - 1);
}
static double full_hit_entries(const rk_db *db, uint32_t symbols) {
    const double kmers = symbols > db->info.k ? (double)(symbols - db->info.k + 1) : 0.0;
    return db->info.n_keys ? kmers * (double)db->info.n_entries / (double)db->info.n_keys : 0.0;
}
static uint32_t hash_key_limit(uint32_t log_slots = RK_HASH_LOG_SLOTS) {
    uint32_t slack = RK_HASH_KEY_SLACK;
    if (const char *e = rk_knob("RK_HASH_KEY_SLACK")) slack = (uint32_t)atoi(e);  // developer knob
    const uint32_t ns = 1u << log_slots;
    if (slack < 16u) slack = 16u;
    if (slack > ns - 64u) slack = ns - 64u;
    return ns - slack;
}

// ---- which kernel goes first on a windowed tree with short rows, per class of batch ----
enum First { F_NONE, F_SORTED, F_HASH_BIG, F_HASH_SMALL };
struct FirstPlan { First for_uniform, for_sparse, for_clade; };
static FirstPlan first_kernel_plan(const rk_db *db, double est_units, bool hash_small, bool hash_fits, bool sorted_fits, bool verdict) {
    const uint32_t nb_tree = db->info.n_branches;
    const bool forced = rk_knob("RK_HASH_ALWAYS") != nullptr;
    const First table_u = hash_small ? F_HASH_SMALL : F_HASH_BIG;
    FirstPlan p;
    if (forced && hash_fits) {
        p.for_uniform = p.for_sparse = p.for_clade = table_u;
        return p;
    }
    const uint32_t min_u = verdict ? hash_min_uniform(est_units, hash_small) : hash_min_single(est_units, hash_small);
    p.for_uniform = hash_fits && nb_tree > min_u ? table_u : sorted_fits ? F_SORTED : hash_fits && hash_tree(db, est_units, hash_small) ? table_u : F_NONE;
    p.for_clade = p.for_sparse = p.for_uniform;
    if (verdict) {
        // reads of a clade touch a third of the branches uniform reads do (profiles/r04_lsize_hist.txt: ~500 against ~1 300; max 1 135): their
        // tables fit the 1 024-slot instantiation -- sixteen waves per CU -- and the few that do not are placed by the large one, launched
        // behind it on the tiles it hands over
        const bool clade_small = hash_fits && !hash_small && (rk_knob("RK_HASH_CLADE_SMALL") || nb_tree > hash_min_clade_small(est_units)) && !rk_knob("RK_HASH_BIG_TABLE");
        p.for_clade = clade_small ? F_HASH_SMALL : hash_fits && nb_tree > hash_min_clade(hash_small) ? table_u : sorted_fits ? F_SORTED : p.for_uniform;
        // uniform batches whose k-mers hit no more often than a random read's (the pre-pass's second verdict): the uniform estimate of a read's
        // entries holds, and where that fits the 1 024-slot table the small instantiation serves them (the large one behind it, as for clades)
        if (hash_fits && !hash_small && !rk_knob("RK_HASH_BIG_TABLE") && est_units * 9.3 <= 0.6 * hash_key_limit(RK_HASH_LOG_SLOTS - 1) && nb_tree > hash_min_uniform(est_units, true))
            p.for_sparse = F_HASH_SMALL;
    }
    return p;
}

static int launch_windowed(const rk_db *db, PlaceArgs a, hipStream_t stream) {
    WindowPlan wp = db->wp;
    const uint64_t n_tiles = (a.n_reads + 3) / 4;
    if (!n_tiles) return RK_OK;
    static const bool no_stream = rk_knob("RK_NO_WSTREAM") != nullptr;  // developer knob: place_packed16w_kernel alone (A/B)
    // (the reads of this batch may be longer than the 150 symbols the image was judged for: their k-mers x the image's row units per code)
    const uint32_t max_syms = a.lens ? a.words_per_read * 32 / db->info.bits_per_symbol : a.fixed_len;
    const double est_units = (max_syms > db->info.k ? max_syms - db->info.k + 1 : 0) * wp.units_per_code;
    // (reads of one known length whose k-mers do not fit its single probe batch would all be handed over: not launched for those)
    const uint32_t probe_cap = (db->info.bits_per_symbol == 5 ? 7u : 9u) * 16u;
    const bool one_batch = a.lens != nullptr || a.fixed_len < db->info.k || a.fixed_len - db->info.k + 1 <= probe_cap;
    // place_hash64_kernel first: images of short rows (the rule place_packed16s_kernel had), reads whose distinct branches -- at most
    // their entries, ~9.3 a unit with C2-like rows -- fit the table (profiles/r04_lsize_hist.txt)
    const bool hash_fits = !no_stream && hash_capable(db) && (est_units * 9.3 <= 0.8 * hash_key_limit() || rk_knob("RK_HASH_ALWAYS"));
    // reads of few row units (a protein database: ~300 entries a read): a table of 1 024 slots -- half the reset and the scan, 9.7 KB a
    // wave, sixteen waves per CU instead of nine
    const bool hash_small = hash_small_table(full_hit_entries(db, max_syms));
    const bool sorted_fits = a.words_per_read <= 16 && !no_stream && wp.stream && one_batch && (est_units <= 1.25 * RK_WSTREAM_MAX_UNITS || rk_knob("RK_WSTREAM_ALWAYS"));
    TileOrder order;
    if (int rc = order.prepare(db, a, stream, hash_fits || sorted_fits)) return rc;
    const bool first_ok = a.tile_marks != nullptr;  // (no scratch to be had for the marks: place_packed16w_kernel alone)
    // Which kernel goes first for each class of batch (first_kernel_plan); when the batch went through the re-tiling pre-pass (a.perm) its
    // verdicts are on the device: the kernels that differ between the classes are launched side by side and return at once when the batch
    // is not theirs (PlaceArgs::only_if).  Without the verdicts (small batches) one rule serves all.
    const bool verdict = a.perm != nullptr;
    const FirstPlan plan = first_kernel_plan(db, est_units, hash_small, hash_fits, sorted_fits, verdict);
    const First for_uniform = plan.for_uniform, for_sparse = plan.for_sparse, for_clade = plan.for_clade;
    // every distinct kernel once, with the classes of batches it serves (PlaceArgs::only_if: bit 0 uniform reads that hit often, bit 1
    // uniform reads that hit like random ones, bit 2 reads of a clade; all three = unconditional)
    const First by_class[3] = {for_uniform, for_sparse, for_clade};
    auto class_mask = [&](First f) -> uint32_t {
        uint32_t m = 0;
        for (int c = 0; c < 3; c++) m |= by_class[c] == f ? 1u << c : 0u;
        return m;
    };
    auto launch_hash = [&](uint32_t log_slots, uint32_t only_if, uint32_t only_marked) -> int {
        PlaceArgs b = a;
        b.only_if = only_if;
        b.s_stride = 1u << log_slots; b.main_cap = RK_HASH_MAIN_CAP; b.work_cap = hash_key_limit(log_slots); b.list_cap = 0; b.only_marked = only_marked;
        const size_t lds_wave = (size_t)(2 * b.s_stride + 64 + b.main_cap) * 4;
        auto launch = [&](auto kern) -> int {
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wave));
            uint64_t per_cu = 0;
            uint64_t want = db->lds_per_cu / lds_wave;
            if (const char *e = rk_knob("RK_HASH_WAVES")) want = std::min<uint64_t>(want, (uint64_t)atoi(e));  // developer knob
            if (int rc = resident_blocks(kern, 64, lds_wave, want, per_cu)) return rc;
            uint64_t blocks = (uint64_t)db->cu_count * per_cu;
            if (blocks > n_tiles) blocks = n_tiles;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds_wave, stream, b);
            return RK_OK;
        };
        int rc;
        if (log_slots != RK_HASH_LOG_SLOTS) rc = db->info.bits_per_symbol == 2 ? launch(place_hash64_kernel<2, RK_HRING, RK_HNPL, 3, RK_HASH_LOG_SLOTS - 1>) : launch(place_hash64_kernel<5, RK_HRING, RK_HNPL, 2, RK_HASH_LOG_SLOTS - 1>);
        else rc = db->info.bits_per_symbol == 2 ? launch(place_hash64_kernel<2, RK_HRING, RK_HNPL, 3, RK_HASH_LOG_SLOTS>) : launch(place_hash64_kernel<5, RK_HRING, RK_HNPL, 2, RK_HASH_LOG_SLOTS>);
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
        return RK_OK;
    };
    auto compact_marks = [&]() -> int {  // the marked tiles as a list + the queue's counters, for the next launch
        if (!a.marked_list) return RK_OK;
        HIP_TRY(hipMemsetAsync(a.marked_ctl, 0, 8, stream));
        const unsigned nblk = (unsigned)std::min<uint64_t>((n_tiles + 255) / 256, 1024);
        hipLaunchKernelGGL(compact_marks_kernel, dim3(nblk), dim3(256), 0, stream, (const unsigned char *)a.tile_marks, n_tiles, a.marked_list, a.marked_ctl);
        HIP_TRY(hipGetLastError());
        return RK_OK;
    };
    bool any_first = false, small_on_trust = false;
    if (first_ok) {
        for (First f : {F_HASH_SMALL, F_HASH_BIG}) {
            const uint32_t m = class_mask(f);
            if (!m) continue;
            if (int rc = launch_hash(f == F_HASH_SMALL ? RK_HASH_LOG_SLOTS - 1 : RK_HASH_LOG_SLOTS, m == 7u ? 0u : m, 0u)) return rc;
            any_first = true;
            small_on_trust = small_on_trust || (f == F_HASH_SMALL && !hash_small);
        }
        if (small_on_trust && a.marked_list) {  // the tiles the small table handed over: the large one next, then place_packed16w_kernel for what is left
            if (int rc = compact_marks()) return rc;
            const uint32_t m = class_mask(F_HASH_SMALL);  // (only for the batches the small table took)
            if (int rc = launch_hash(RK_HASH_LOG_SLOTS, m == 7u ? 0u : m, 1u)) return rc;
        }
    }
    const bool sorted_first = first_ok && class_mask(F_SORTED) != 0u;
    const uint32_t sorted_only_if = class_mask(F_SORTED) == 7u ? 0u : class_mask(F_SORTED);
    const bool hash_first = any_first;
    if (sorted_first) {
        // ---- place_packed16s_kernel: the sorted list of a tile's four reads + their touched bitmaps.  Seven waves per CU on
        //      the largest windows, eight otherwise; the list holds a C2-like read (145 units, 250 at the tail) with the padding of
        //      its window segments ----
        PlaceArgs b = a;
        b.only_if = sorted_only_if;  // (when another kernel takes batches of the other shape)
        const uint32_t work_min = 96u;  // scratch of the second pass: 48 candidate keys
        // ring of row loads: eight deep, a window's segment padded to half turns of it (four deep it left the stream waiting
        // on HBM: ~280 cycles a step; segments padded to whole turns of eight made the largest trees' lists half filler)
        const uint32_t ring = 8u;
        uint32_t work = std::max(work_min, 64u + 3u * ring);  // (also: the 64 window counters of the emit, the touched bitmap of the stream)
        work = (work + 1) & ~1u;
        // the list: a C2-like read's 145 units (250 at the tail) + the padding of its window segments to the tile's longest
        const uint32_t need = 200 + 5 * wp.n_win + 3 * ring;
        const uint32_t s_str = wp.W + 16;  // a scratch word per lane in front of the window's slots
        uint32_t budget = 0;
        // two waves per SIMD (182 registers for DNA, 255 for amino acids).  Tried: three (167 registers, 28 bytes of scratch) and
        // the nine to eleven waves per CU the LDS then allows -- T8k 137 -> 97, T20k 102 -> 93 Mreads/s: the kernel is bound by
        // instruction issue, not by latency
        const uint32_t max_waves = 8u;
        for (uint32_t waves = max_waves; waves >= 5u; waves--) {
            budget = (160 * 1024 / waves / 512 * 512) / 4 / 4;  // (whole 512-byte granules per wave: see resident_blocks)
            if (budget >= s_str + work + need) break;
        }
        uint32_t mainc = budget - s_str - work;
        if (mainc > 640) mainc = 640;
        mainc &= ~3u;  // (the list is read four items at a time)
        b.s_stride = s_str; b.main_cap = mainc; b.work_cap = work; b.list_cap = work / 2; b.only_marked = 0;
        const size_t lds_wave = (size_t)4 * (b.s_stride + b.main_cap + b.work_cap) * 4;
        uint32_t waves_cu = (uint32_t)(db->lds_per_cu / lds_wave);
        if (waves_cu < 1) return fail(RK_ERR_UNSUPPORTED, "internal: windowed geometry does not fit the LDS");
        if (waves_cu > max_waves) waves_cu = max_waves;
        auto launch = [&](auto kern) -> int {
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wave));
            uint64_t per_cu = 0;
            if (int rc = resident_blocks(kern, 64, lds_wave, waves_cu, per_cu)) return rc;
            uint64_t blocks = (uint64_t)db->cu_count * per_cu;
            if (blocks > n_tiles) blocks = n_tiles;
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds_wave, stream, b);
            return RK_OK;
        };
        int rc;
        const bool wide = wp.W > 512;  // two bitmap words a lane
        if (db->info.bits_per_symbol == 2) rc = wide ? launch(place_packed16s_kernel<2, 8, 9, true>) : launch(place_packed16s_kernel<2, 8, 9, false>);
        else rc = wide ? launch(place_packed16s_kernel<5, 8, 7, true>) : launch(place_packed16s_kernel<5, 8, 7, false>);  // (<= 102 residues in 16 words)
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
    }
    // ---- place_packed16w_kernel: every tile (records of more than 16 words), or the tiles the first kernel handed over ----
    a.only_marked = ((hash_first || sorted_first) && first_ok) ? 1u : 0u;
    if (a.only_marked)
        if (int rc = compact_marks()) return rc;
    // 88 words = the 44 keys the exact select of a window needs as scratch for keep_at_most <= 8 (K + 16 candidates + 16 winners); 96 beyond
    const uint32_t work_min = a.keep_at_most > 8 ? 96u : 88u;
    if (wp.work_cap < work_min) {
        wp.main_cap -= work_min - wp.work_cap;
        wp.work_cap = work_min;
    }
    if (a.words_per_read > 10 && wp.work_cap > work_min) {  // reads beyond ~160 bases: a whole read in the main list matters more than one accumulate call per window
        wp.main_cap += wp.work_cap - work_min;
        wp.work_cap = work_min;
    }
    a.s_stride = wp.s_stride; a.main_cap = wp.main_cap; a.work_cap = wp.work_cap; a.list_cap = wp.work_cap / 2;
    const size_t lds_wave = (size_t)4 * (wp.s_stride + wp.main_cap + wp.work_cap) * 4;
    uint32_t waves_cu = (uint32_t)(db->lds_per_cu / lds_wave);
    if (waves_cu < 1) return fail(RK_ERR_UNSUPPORTED, "internal: windowed geometry does not fit the LDS");
    if (waves_cu > 8) waves_cu = 8;  // two waves per SIMD: the kernel's register budget
    auto launch = [&](auto kern) -> int {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wave));
        uint64_t per_cu = 0;
        if (int rc = resident_blocks(kern, 64, lds_wave, waves_cu, per_cu)) return rc;
        uint64_t blocks = (uint64_t)db->cu_count * per_cu;
        if (blocks > n_tiles) blocks = n_tiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds_wave, stream, a);
        return RK_OK;
    };
    if (int rc = db->info.bits_per_symbol == 2 ? launch(place_packed16w_kernel<2, RK_WRING, 9>) : launch(place_packed16w_kernel<5, RK_WRING, 9>)) return rc;
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

static int launch_place(const rk_db *db, const Geometry &g, const PlaceArgs &a_in, hipStream_t s) {
    // (reads of one clade read the same rows: taken together they find them in the L2 -- scripts/clade_sorted_probe.py)
    PlaceArgs a = a_in;
    TileOrder order;
    if (int rc = order.prepare(db, a, s, false)) return rc;
    switch (g.G) {
    case 8: return launch_b<8>(db, g, a, s);
    case 16: return launch_b<16>(db, g, a, s);
    case 32: return launch_b<32>(db, g, a, s);
    default: return launch_b<64>(db, g, a, s);
    }
}

// ---- large trees: one workgroup per read ----
struct WgGeometry {
    uint32_t nw, s_stride, list_cap, wgs_per_cu, n_pass;
    size_t lds;
};

// The score vector of one read (4 bytes per branch) is shared by the NW waves of a workgroup.  While it fits, one pass:
// two workgroups of 8 waves per CU when two vectors fit (C5: 19 999 branches = 80 KB), else one of 16 waves.  Trees beyond one
// CU's LDS (about 39 000 branches; the reference's limit is the 16-bit id: 65 534) take 2 or 4 branch-range passes per read.
static int choose_wg_geometry(const rk_db *db, WgGeometry &g, uint32_t keep_at_most = 16) {
    const uint32_t nb = db->info.n_branches;
    uint32_t min_pass = 1;
    if (const char *e = rk_knob("RK_WG_PASSES")) min_pass = (uint32_t)atoi(e);  // developer / test knob: force 2 or 4 passes on a tree that fits in one
    for (uint32_t P : {1u, 2u, 4u}) {
        if (P < min_pass) continue;
        const uint32_t span = 32 / P;
        uint32_t win = 0;
        for (uint32_t p = 0; p < P; p++) {
            const uint32_t lo = (uint32_t)(((uint64_t)p * span * nb) / 32), hi = (uint32_t)(((uint64_t)(p + 1) * span * nb) / 32);
            win = std::max(win, hi - lo);
        }
        const uint32_t s_stride = (win + 4) & ~3u;  // >= win + 1 (word win is the scratch slot), multiple of 4 (b128 scans)
        const size_t s_bytes = (size_t)s_stride * 4;
        for (uint32_t wgs : {2u, 1u}) {
            if (P > 1 && wgs == 2) continue;  // (a tree that needs passes with 8 waves fits whole with 16)
            const uint32_t nw = wgs == 2 ? 8 : 16;
            const size_t budget = db->lds_per_cu / wgs;
            // P == 1: the wave winners of the level-1 select go to the hit list (free once every wave has its slices of the batch's rows);
            // P > 1: a region of their own.  (C5's two workgroups of 80 000-byte score vectors per CU leave ~1.9 KB)
            const size_t cand = P > 1 ? (size_t)(P * nw * keep_at_most + 16 + 2) * 8 : 0;
            const size_t extra = 256 + cand;  // per-wave hit counters
            const size_t min_list = P > 1 ? 66 : std::max<size_t>(66, (size_t)nw * keep_at_most + 18);
            if (budget < s_bytes + extra + min_list * 8) continue;
            size_t cap = (budget - s_bytes - extra) / 8;
            if (cap > 512) cap = 512;
            g.list_cap = (uint32_t)cap & ~1u;
            g.wgs_per_cu = wgs;
            g.nw = nw;
            g.n_pass = P;
            g.s_stride = s_stride;
            g.lds = s_bytes + (size_t)g.list_cap * 8 + extra;
            return RK_OK;
        }
    }
    return fail(RK_ERR_UNSUPPORTED, "n_branches=%u: no score-vector window fits one CU's LDS (%zu B)", nb, db->lds_per_cu);
}

static int check_launchable(const rk_db *db) {
    if (db->windowed) return RK_OK;  // (the windowed and the ambiguity kernel hold windows of any tree; a forced dense geometry is checked at launch)
    if (db->indexed) {
        WgGeometry wg;
        return choose_wg_geometry(db, wg);
    }
    Geometry g;
    return choose_geometry(db, 7, g);
}

template <int BITS, int TM>
static int launch_wg_v(const rk_db *db, const WgGeometry &g, PlaceArgs a, hipStream_t stream) {
    a.s_stride = g.s_stride;
    a.list_cap = g.list_cap;
    a.n_pass = g.n_pass;
    if (!a.n_reads) return RK_OK;
    auto launch = [&](auto kern) -> int {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
        uint64_t per_cu = 0;
        if (int rc = resident_blocks(kern, 64 * (int)g.nw, g.lds, g.wgs_per_cu, per_cu)) return rc;
        uint64_t blocks = (uint64_t)db->cu_count * per_cu;
        if (blocks > a.n_reads) blocks = a.n_reads;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * g.nw), g.lds, stream, a);
        return RK_OK;
    };
    if (int rc = db->info.rows_bytes < ROWS_FIT32_LIMIT ? launch(place_wg_kernel<BITS, TM, false, RK_WG_RING>) : launch(place_wg_kernel<BITS, TM, true, RK_WG_RING>)) return rc;
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

static int launch_wg(const rk_db *db, const WgGeometry &g, const PlaceArgs &a, hipStream_t s) {
    const bool dna = db->info.bits_per_symbol == 2;
    if (db->info.table_mode == RK_TABLE_HASH) return dna ? launch_wg_v<2, TM_HASH>(db, g, a, s) : launch_wg_v<5, TM_HASH>(db, g, a, s);
    return dna ? launch_wg_v<2, TM_DIRECT8>(db, g, a, s) : launch_wg_v<5, TM_DIRECT8>(db, g, a, s);
}

template <int BITS, int TM>
static int launch_ascii_v(const rk_db *db, PlaceArgs args, AmbArgs m, hipStream_t stream) {
    // LDS: S[s_stride] + candidate list + Samb/Camb windows of `chunk` branches.  S holds the whole tree while that leaves room
    // for the list and a minimal Samb/Camb window (then large trees take several ambiguity passes over the alternatives);
    // beyond that (about 39 000 branches) S itself becomes a window of the tree and the read is walked once per window.
    const uint32_t nb = db->info.n_branches;
    const size_t list_bytes = (size_t)ASCII_LIST_CAP * 8 + ASCII_TABLE_BYTES;  // (+ the alphabet's tables, behind Camb)
    uint32_t s_win = nb;
    size_t chunk;
    const bool force_windows = db->indexed && rk_knob("RK_WG_PASSES") && atoi(rk_knob("RK_WG_PASSES")) > 1;  // same test knob
    if (!force_windows && (size_t)((nb + 4) & ~3u) * 4 + list_bytes + 8 * 64 <= db->lds_per_cu) {
        args.s_stride = (nb + 4) & ~3u;
        const size_t fixed = (size_t)args.s_stride * 4 + list_bytes;
        chunk = args.s_stride;
        const size_t budget = 64 * 1024;  // prefer several waves per CU; grow only if a single pass would not fit
        if (fixed + 8 * chunk > budget) {
            size_t avail = (fixed + 8 * 64 <= budget ? budget : db->lds_per_cu) - fixed;
            chunk = avail / 8;
            if (chunk > args.s_stride) chunk = args.s_stride;
        }
    } else {
        // 12 bytes per branch of the window (S + Samb + Camb): the window is as large as one CU allows, split evenly
        const size_t per = (db->lds_per_cu - list_bytes - 64) / 12;
        uint32_t n_win = (uint32_t)((nb + per - 1) / per);
        if (force_windows && n_win < 3) n_win = 3;
        s_win = ((nb + n_win - 1) / n_win + 3) & ~3u;
        args.s_stride = s_win + 4;
        chunk = s_win;
    }
    if (chunk == 0 || args.s_stride == 0) return fail(RK_ERR_INVALID, "internal: ambiguity kernel launched without a score-vector geometry");
    m.amb_chunk = (uint32_t)chunk;
    m.s_win = s_win;
    const size_t lds = (size_t)args.s_stride * 4 + list_bytes + 8 * chunk;  // S | list | Samb | Camb | tables
    const uint64_t groups = (args.n_reads + 63) / 64;
    uint64_t waves_cu = db->lds_per_cu / lds;
    if (waves_cu > 32) waves_cu = 32;
    if (!groups) return RK_OK;
    auto launch = [&](auto kern) -> int {
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        uint64_t per_cu = 0;
        if (int rc = resident_blocks(kern, 64, lds, waves_cu, per_cu)) return rc;
        uint64_t blocks = (uint64_t)db->cu_count * per_cu;
        if (blocks > groups) blocks = groups;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds, stream, args, m);
        return RK_OK;
    };
    if (int rc = db->indexed ? launch(place_ascii_kernel<BITS, TM, true>) : launch(place_ascii_kernel<BITS, TM, false>)) return rc;
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

static int launch_ascii(const rk_db *db, const PlaceArgs &a, const AmbArgs &m, hipStream_t s) {
    const bool dna = db->info.bits_per_symbol == 2;
    switch (db->info.table_mode) {
    case RK_TABLE_DIRECT: return dna ? launch_ascii_v<2, TM_COMPACT>(db, a, m, s) : launch_ascii_v<5, TM_COMPACT>(db, a, m, s);
    case RK_TABLE_DIRECT8: return dna ? launch_ascii_v<2, TM_DIRECT8>(db, a, m, s) : launch_ascii_v<5, TM_DIRECT8>(db, a, m, s);
    default: return dna ? launch_ascii_v<2, TM_HASH>(db, a, m, s) : launch_ascii_v<5, TM_HASH>(db, a, m, s);
    }
}

static int check_params(const rk_params *p) {
    if (!p) return fail(RK_ERR_INVALID, "null rk_params");
    if (p->keep_at_most < 1 || p->keep_at_most > 16) return fail(RK_ERR_INVALID, "keep_at_most=%u outside 1..16", p->keep_at_most);
    if (p->amb_mode > RK_AMB_MAX) return fail(RK_ERR_INVALID, "amb_mode=%u invalid", p->amb_mode);
    if (std::isnan(p->keep_factor) || std::isnan(p->ns_bound)) return fail(RK_ERR_INVALID, "NaN in rk_params");
    return RK_OK;
}

extern "C" const char *rk_kernel_name(const rk_db *db) {
    if (!db) return "";
    Geometry g;
    rk_db *m = const_cast<rk_db *>(db);
    char buf[800];
    if (db->indexed && db->lanes_per_read == 0) {
        WgGeometry wg;
        if (choose_wg_geometry(db, wg, 7) != RK_OK) return "";
        snprintf(buf, sizeof(buf), "place_wg_kernel<BITS=%u,%s,%s,U=%d> waves/WG=%u lds/WG=%zuB rows/batch=%u WGs/CU=%u passes=%u",
                 db->info.bits_per_symbol, db->info.table_mode == RK_TABLE_HASH ? "HASH" : "DIRECT8",
                 db->info.rows_bytes < ROWS_FIT32_LIMIT ? "OFF32" : "OFF64", RK_WG_RING, wg.nw, wg.lds, wg.list_cap, wg.wgs_per_cu, wg.n_pass);
        m->kernel_name = buf;
        return m->kernel_name.c_str();
    }
    if (use_windowed(db, 7, 16)) {
        // (for the reads of BASELINE's configs -- 150 bases / 100 residues -- and a batch large enough for the pre-pass's verdicts)
        const uint32_t syms_name = db->info.bits_per_symbol == 5 ? 100u : 150u;
        const double est_name = (double)(syms_name - db->info.k + 1) * db->wp.units_per_code;
        const bool hash_fits = hash_capable(db) && (rk_knob("RK_HASH_ALWAYS") || est_name * 9.3 <= 0.8 * hash_key_limit());
        const bool small_name = hash_small_table(full_hit_entries(db, syms_name));
        const bool sorted_name = db->wp.stream && !rk_knob("RK_NO_WSTREAM") && (est_name <= 1.25 * RK_WSTREAM_MAX_UNITS || rk_knob("RK_WSTREAM_ALWAYS"));
        const FirstPlan pl = first_kernel_plan(db, est_name, small_name, hash_fits, sorted_name, true);
        auto what = [](First f) { return f == F_HASH_SMALL ? "place_hash64_kernel with 1 024 slots (the 2 048-slot one behind it)" : f == F_HASH_BIG ? "place_hash64_kernel" : f == F_SORTED ? "place_packed16s_kernel" : "place_packed16w_kernel"; };
        const First shown = pl.for_uniform == F_HASH_BIG || pl.for_uniform == F_HASH_SMALL ? pl.for_uniform : (pl.for_sparse == F_HASH_SMALL || pl.for_clade == F_HASH_SMALL || pl.for_clade == F_HASH_BIG) && pl.for_uniform == F_NONE ? pl.for_clade : pl.for_uniform;
        if (shown == F_HASH_BIG || shown == F_HASH_SMALL) {
            const uint32_t ls = shown == F_HASH_SMALL ? RK_HASH_LOG_SLOTS - 1 : RK_HASH_LOG_SLOTS;
            char other[360] = "";
            if (pl.for_clade != pl.for_uniform || pl.for_sparse != pl.for_uniform)
                snprintf(other, sizeof(other), "; batches of 32 768 reads or more, judged on the device: clade-shaped -> %s%s%s", what(pl.for_clade),
                         pl.for_sparse != pl.for_uniform ? ", uniform and sparse-hit -> " : "", pl.for_sparse != pl.for_uniform ? what(pl.for_sparse) : "");
            snprintf(buf, sizeof(buf), "place_hash64_kernel<BITS=%u,U=%d,NPL=%d,PU=%d,LOGS=%u> %u slots, <= %u keys a read%s (+ place_packed16w_kernel for the tiles it hands over; windows=%u x %u branches)",
                     db->info.bits_per_symbol, RK_HRING, RK_HNPL, db->info.bits_per_symbol == 5 ? 2 : 3, ls, 1u << ls, hash_key_limit(ls), other, db->wp.n_win, db->wp.W);
        } else if (pl.for_uniform == F_SORTED) {
            char other[360] = "";
            if (pl.for_clade != pl.for_uniform || pl.for_sparse != pl.for_uniform)
                snprintf(other, sizeof(other), "; batches of 32 768 reads or more, judged on the device: clade-shaped -> %s, uniform and sparse-hit -> %s", what(pl.for_clade), what(pl.for_sparse));
            snprintf(buf, sizeof(buf), "place_packed16s_kernel<BITS=%u,U=8,PU=%d,WIDE=%d> windows=%u x %u branches%s (+ place_packed16w_kernel for the tiles it hands over)",
                     db->info.bits_per_symbol, db->info.bits_per_symbol == 5 ? 7 : 9, db->wp.W > 512 ? 1 : 0, db->wp.n_win, db->wp.W, other);
        }
        else
        snprintf(buf, sizeof(buf), "place_packed16w_kernel<BITS=%u,U=%d,PU=9> windows=%u x %u branches lds/wave=%zuB main=%u work=%u",
                 db->info.bits_per_symbol, RK_WRING, db->wp.n_win, db->wp.W, (size_t)16 * (db->wp.s_stride + db->wp.main_cap + db->wp.work_cap),
                 db->wp.main_cap, db->wp.work_cap);
        m->kernel_name = buf;
        return m->kernel_name.c_str();
    }
    if (choose_geometry(db, 7, g) != RK_OK) return "";
    PlaceArgs probe{};
    probe.words_per_read = 16;
    // (the tile-pipelined variant serves packed records of <= 16 words, i.e. reads of <= 256 bases / 102 residues; longer
    // records take place_packed_kernel with the same geometry)
    snprintf(buf, sizeof(buf), "%s<G=%u,BITS=%u,%s,%s,U=%d,PU=%u> lds/wave=%zuB cap=%u waves/CU<=%u (LDS; registers may allow fewer)",
             use_pipelined16(db, g, probe) ? "place_packed16_kernel" : "place_packed_kernel", g.G, db->info.bits_per_symbol, db->info.table_mode == RK_TABLE_DIRECT ? (db->compact_nib ? "DIRECT4" : "DIRECT") : (db->info.table_mode == RK_TABLE_DIRECT8 ? "DIRECT8" : "HASH"),
             db->info.rows_bytes < ROWS_FIT32_LIMIT ? "ITEM32" : "ITEM64", g.G == 64 ? RK_RING64 : RK_RING, g.pu, g.lds_per_wave, g.list_cap, g.waves_per_cu);
    m->kernel_name = buf;
    return m->kernel_name.c_str();
}

// ------------------------------------------------------------------------------------------------
// device entry points
// ------------------------------------------------------------------------------------------------
extern "C" int rk_pack_reads_device(rk_db *db, uint64_t n_reads, const uint8_t *d_seq_ascii, const uint64_t *d_seq_off,
                                    uint32_t words_per_read, uint32_t *d_packed, uint32_t *d_lens, uint32_t *d_flags,
                                    void *stream) {
    if (!db || !d_seq_ascii || !d_seq_off || !d_packed || !d_lens || !d_flags || words_per_read == 0)
        return fail(RK_ERR_INVALID, "rk_pack_reads_device: null/zero argument");
    if (n_reads == 0) return RK_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(db->info.device));
    HIP_TRY(hipMemsetAsync(d_flags, 0, n_reads * sizeof(uint32_t), s));
    const uint64_t total = n_reads * words_per_read;
    uint64_t blocks = (total + 255) / 256;
    const uint64_t maxb = (uint64_t)db->cu_count * 16;
    if (blocks > maxb) blocks = maxb;
    if (db->info.bits_per_symbol == 2)
        hipLaunchKernelGGL(pack_reads_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, d_seq_ascii, (const u64 *)d_seq_off, (u64)n_reads,
                           words_per_read, (const unsigned char *)db->d_alpha, db->info.k, d_packed, d_lens, d_flags);
    else
        hipLaunchKernelGGL(pack_reads_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, s, d_seq_ascii, (const u64 *)d_seq_off, (u64)n_reads,
                           words_per_read, (const unsigned char *)db->d_alpha, db->info.k, d_packed, d_lens, d_flags);
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

extern "C" int rk_place_packed_device(rk_db *db, const rk_params *p, uint64_t n_reads, const uint32_t *d_packed,
                                      uint32_t words_per_read, const uint32_t *d_lens, uint32_t fixed_len,
                                      const uint32_t *d_flags_in, const uint8_t *d_seq_ascii, const uint64_t *d_seq_off,
                                      const rk_result *d_out, void *stream) {
    if (!db || !d_out) return fail(RK_ERR_INVALID, "rk_place_packed_device: null argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (n_reads == 0) return RK_OK;
    if (!d_packed || words_per_read == 0) return fail(RK_ERR_INVALID, "rk_place_packed_device: null packed reads");
    if (!d_out->n_rows || !d_out->branch || !d_out->score || !d_out->lwr || !d_out->flags)
        return fail(RK_ERR_INVALID, "rk_place_packed_device: null result array");
    if (!d_lens && (uint64_t)fixed_len * db->info.bits_per_symbol > (uint64_t)words_per_read * 32)
        return fail(RK_ERR_INVALID, "rk_place_packed_device: fixed_len=%u does not fit %u words", fixed_len, words_per_read);
    const bool use_wg = db->indexed && db->lanes_per_read == 0;  // an explicit lanes_per_read forces the single-wave kernel
    const bool use_win = !use_wg && use_windowed(db, p->keep_at_most, words_per_read);
    Geometry g{};
    WgGeometry wg{};
    rc = use_wg ? choose_wg_geometry(db, wg, p->keep_at_most) : (use_win ? RK_OK : choose_geometry(db, p->keep_at_most, g));  // (the windowed launch has its own plan)
    if (rc) return rc;
    HIP_TRY(hipSetDevice(db->info.device));
    hipStream_t s = (hipStream_t)stream;
    PlaceArgs a{};
    a.db = db->view;
    a.n_reads = n_reads;
    a.packed = d_packed; a.words_per_read = words_per_read; a.lens = d_lens; a.fixed_len = fixed_len;
    a.flags_in = d_flags_in;
    const bool ascii = d_flags_in && d_seq_ascii && d_seq_off;
    a.has_ascii = ascii ? 1u : 0u;
    a.keep_at_most = p->keep_at_most; a.keep_factor = p->keep_factor; a.ns_bound = p->ns_bound;
    a.o_nrows = d_out->n_rows; a.o_branch = d_out->branch; a.o_score = d_out->score; a.o_lwr = d_out->lwr; a.o_flags = d_out->flags;
    a.s_stride = use_wg ? wg.s_stride : g.s_stride;
    a.list_cap = use_wg ? wg.list_cap : g.list_cap;
    if (use_wg) rc = launch_wg(db, wg, a, s);
    else if (use_win) rc = launch_windowed(db, a, s);
    else rc = launch_place(db, g, a, s);
    if (rc) return rc;
    if (ascii) {
        AmbArgs m{};
        m.ascii = d_seq_ascii; m.seq_off = (const u64 *)d_seq_off;
        m.char_table = db->d_alpha; m.alt_table = db->d_alpha + 256; m.alt_count = db->d_alpha + 576;
        m.amb_mode = p->amb_mode;
        m.max_amb = (uint32_t)std::floor(std::pow((double)db->info.k, 1.0 / (double)db->info.alphabet));  // AmbigSequenceKnife.java:95
        rc = launch_ascii(db, a, m, s);
        if (rc) return rc;
    }
    return RK_OK;
}

// rk_count_work_device: the work a batch asks of the database (count_work_kernel), for callers that want the reference's own
// diagnostics -- k-mers looked up, k-mers found, row entries walked -- next to the placements.  Opt-in and separate: the placement
// kernels carry no counters.
template <int BITS>
static void launch_count(const rk_db *db, const uint32_t *d_packed, uint32_t wpr, const uint32_t *d_lens, uint32_t fixed_len, const uint32_t *d_flags_in,
                         uint64_t n_reads, unsigned long long *d_out, hipStream_t s) {
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_reads + 3) / 4, (uint64_t)db->cu_count * 8);
    switch (db->info.table_mode) {
    case RK_TABLE_DIRECT: hipLaunchKernelGGL((count_work_kernel<BITS, TM_COMPACT>), dim3(blocks), dim3(256), 0, s, db->view, d_packed, wpr, d_lens, fixed_len, d_flags_in, n_reads, d_out); break;
    case RK_TABLE_DIRECT8: hipLaunchKernelGGL((count_work_kernel<BITS, TM_DIRECT8>), dim3(blocks), dim3(256), 0, s, db->view, d_packed, wpr, d_lens, fixed_len, d_flags_in, n_reads, d_out); break;
    default: hipLaunchKernelGGL((count_work_kernel<BITS, TM_HASH>), dim3(blocks), dim3(256), 0, s, db->view, d_packed, wpr, d_lens, fixed_len, d_flags_in, n_reads, d_out); break;
    }
}
extern "C" int rk_count_work_device(rk_db *db, uint64_t n_reads, const uint32_t *d_packed, uint32_t words_per_read, const uint32_t *d_lens,
                                    uint32_t fixed_len, const uint32_t *d_flags_in, rk_work *d_out, void *stream) {
    if (!db || !d_out) return fail(RK_ERR_INVALID, "rk_count_work_device: null argument");
    HIP_TRY(hipSetDevice(db->info.device));
    hipStream_t s = (hipStream_t)stream;
    static_assert(sizeof(rk_work) == 3 * sizeof(unsigned long long), "rk_work is three 64-bit counters");
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(rk_work), s));
    if (n_reads == 0) return RK_OK;
    if (!d_packed || words_per_read == 0) return fail(RK_ERR_INVALID, "rk_count_work_device: null packed reads");
    if (!d_lens && (uint64_t)fixed_len * db->info.bits_per_symbol > (uint64_t)words_per_read * 32)
        return fail(RK_ERR_INVALID, "rk_count_work_device: fixed_len=%u does not fit %u words", fixed_len, words_per_read);
    if (db->info.bits_per_symbol == 2) launch_count<2>(db, d_packed, words_per_read, d_lens, fixed_len, d_flags_in, n_reads, (unsigned long long *)d_out, s);
    else launch_count<5>(db, d_packed, words_per_read, d_lens, fixed_len, d_flags_in, n_reads, (unsigned long long *)d_out, s);
    HIP_TRY(hipGetLastError());
    return RK_OK;
}

// ------------------------------------------------------------------------------------------------
// host-buffer entry point: chunked, two workspaces on two streams so that the upload of chunk c+1 overlaps the
// kernels / download of chunk c; device buffers are kept (grow-only) in the rk_db between calls
// ------------------------------------------------------------------------------------------------
// what the host hands over: ASCII reads (packed on the device) or records already packed on the host (rk_pack_reads_host)
struct HostInput {
    const uint8_t *ascii = nullptr;   // concatenated reads; with `packed` set: only consulted for reads flagged AMBIGUOUS
    const uint64_t *off = nullptr;    // [n + 1]
    const uint32_t *packed = nullptr; // [n][wpr]
    uint32_t wpr = 0;
    const uint32_t *lens = nullptr;   // [n] or NULL (fixed_len)
    uint32_t fixed_len = 0;
    const uint32_t *flags = nullptr;  // [n] or NULL
};

static rk::PackSpec pack_spec(const Alphabet &A, uint32_t alphabet, uint32_t bits, uint32_t k, uint32_t words_per_read) {
    rk::PackSpec P;
    P.table = A.table; P.bits = bits; P.k = k; P.words_per_read = words_per_read;
    P.pad_char = alphabet == RK_ALPHABET_DNA ? 'A' : 'R';  // state 0 of either alphabet
    P.force_scalar = rk_knob("RK_PACK_SCALAR") != nullptr;  // developer / test knob
    return P;
}

static int place_host(rk_db *db, const rk_params *p, uint64_t n_reads, const HostInput &in, rk_result *out, rk_counters *counters,
                      const char *who) {
    const bool packed_in = in.packed != nullptr;
    const uint8_t *seq_ascii = in.ascii;
    const uint64_t *seq_off = in.off;
    rk_counters ct{};
    if (n_reads == 0) { if (counters) *counters = ct; return RK_OK; }
    if (!out->n_rows || !out->branch || !out->score || !out->lwr || !out->flags) return fail(RK_ERR_INVALID, "%s: null result array", who);
    if (seq_off)
        for (uint64_t r = 0; r < n_reads; r++)
            if (seq_off[r + 1] < seq_off[r]) return fail(RK_ERR_INVALID, "%s: seq_off not monotone at read %llu", who, (unsigned long long)r);
    std::lock_guard<std::mutex> lock(db->host_mutex);  // the two workspaces belong to the db: one host call at a time
    HIP_TRY(hipSetDevice(db->info.device));
    const uint32_t K = p->keep_at_most;
    // chunks of 2^18 reads: the kernel still fills the chip (2^16 tiles for 2 048 waves) and the part of a call that nothing
    // overlaps -- the first chunk's upload, the last chunk's download and drain -- stays short
    uint64_t max_chunk_reads = 1ull << 18;
    if (const char *e = rk_knob("RK_CHUNK_READS")) {  // developer knob
        const long v = atol(e);
        if (v >= 1024) max_chunk_reads = (uint64_t)v;
    }
    const uint64_t max_chunk_bytes = 128ull << 20;
    for (rk_workspace &w : db->ws)
        if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    // Caller buffers from rk_host_alloc (or otherwise page-locked) are the DMA's source / target directly; pageable ones
    // (the usual case behind JNI) go through page-locked staging with threaded copies
    auto is_pinned = [](const void *ptr) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    const bool in_pinned = packed_in ? is_pinned(in.packed) : is_pinned(seq_ascii);
    const bool out_pinned = is_pinned(out->n_rows) && is_pinned(out->branch) && is_pinned(out->score) && is_pinned(out->lwr) && is_pinned(out->flags);
    // pageable characters are packed on the host (rk_pack_host.cpp) by this call's worker threads; page-locked ones go to the
    // device as they are (no host work at all) and are packed there
    const bool host_pack = !packed_in && !in_pinned;
    Alphabet alpha;
    if (host_pack) build_alphabet(db->info.alphabet, db->convert_uo != 0, alpha);
    // host threads of this call: staging / packing on one side, result copies on the other (both only for pageable memory)
    unsigned n_stage = in_pinned ? 0u : std::max(1u, host_threads(n_reads, 0) * 5 / 8), n_drain = out_pinned ? 0u : std::max(1u, host_threads(n_reads, 0) * 3 / 8);
    if (const char *e = rk_knob("RK_STAGE_THREADS")) n_stage = (unsigned)std::max(1, atoi(e));   // developer knobs
    if (const char *e = rk_knob("RK_DRAIN_THREADS")) n_drain = (unsigned)std::max(1, atoi(e));
    const NodeCpus *node = &gpu_node_cpus(db->info.device);  // the CPUs next to the GPU: staging threads and page-locked buffers live there
    ForkJoin pool(n_stage ? n_stage - 1 : 0, node);
    auto count_flags = [&](const uint32_t *fl, uint64_t m) {  // per-batch counters, taken chunk by chunk while the flags are cache-hot
        for (uint64_t r = 0; r < m; r++) {
            const uint32_t f = fl[r];
            ct.reads++;
            if (f & RK_FLAG_PLACED) ct.placed++; else ct.unplaced++;
            if (f & RK_FLAG_BAD_CHAR) ct.bad_char++;
            if (f & RK_FLAG_TOO_SHORT) ct.too_short++;
            if (f & RK_FLAG_AMBIGUOUS) ct.ambiguous++;
        }
    };
    auto drain = [&](rk_workspace &w, ForkJoin &dpool) {  // staged results of the workspace's last chunk -> the caller's arrays
        if (!w.pending) return;
        const uint64_t a0 = w.pend_r0, m = w.pend_n;
        if (!out_pinned) {
            // the drain thread's workers, each a range of reads over the five arrays (103 bytes per read at K = 7)
            dpool.run([&](unsigned part, unsigned parts) {
                const uint64_t lo = m * part / parts, c = m * (part + 1) / parts - lo;
                if (!c) return;
                memcpy(out->n_rows + a0 + lo, w.h_nrows.as<uint8_t>() + lo, c);
                memcpy(out->branch + (a0 + lo) * K, w.h_branch.as<uint16_t>() + lo * K, c * K * 2);
                memcpy(out->score + (a0 + lo) * K, w.h_score.as<float>() + lo * K, c * K * 4);
                memcpy(out->lwr + (a0 + lo) * K, w.h_lwr.as<double>() + lo * K, c * K * 8);
                memcpy(out->flags + a0 + lo, w.h_oflags.as<uint32_t>() + lo, c * 4);
            });
        }
        count_flags(out->flags + a0, m);
        w.pending = false;
    };
    unsigned chunk_no = 0;
    int status = RK_OK;
    // developer knob: RK_HOST_TIMING=1 prints where the host thread of this call spent its time (stderr)
    const bool timing = rk_knob("RK_HOST_TIMING") != nullptr;
    double t_wait = 0, t_drain = 0, t_stage = 0, t_enq = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // Four workspaces in flight.  The call's worker threads stage chunk c + 1 (pack its characters / copy its records into
    // page-locked memory) while this thread enqueues chunk c, and a second host thread waits for the stream of the oldest chunk
    // and moves its results into the caller's arrays: staging, enqueueing, the GPU's work and the result copies of different
    // chunks overlap (one host thread doing everything by turns kept the GPU waiting: 1.6e8 reads/s on C2).
    constexpr unsigned NWS = 4;
    std::mutex qm;
    std::condition_variable qcv;
    std::deque<unsigned> submitted;   // workspace indices in submission order
    bool ws_busy[NWS] = {false, false, false, false};
    bool closing = false;
    int drain_status = RK_OK;
    std::string drain_msg;
    const int device = db->info.device;
    std::thread drainer([&]() {
        pin_this_thread(node);
        (void)hipSetDevice(device);
        std::unique_ptr<ForkJoin> dpool;
        try {
            dpool.reset(new ForkJoin(n_drain ? n_drain - 1 : 0, node));
        } catch (...) {  // no worker threads: this thread copies alone
        }
        ForkJoin none(0);
        while (true) {
            unsigned wi;
            {
                std::unique_lock<std::mutex> lk(qm);
                qcv.wait(lk, [&]() { return !submitted.empty() || closing; });
                if (submitted.empty()) return;
                wi = submitted.front();
                submitted.pop_front();
            }
            rk_workspace &w = db->ws[wi];
            const double t0 = now();
            const hipError_t he = hipStreamSynchronize(w.stream);
            const double t1 = now();
            if (he != hipSuccess) {
                std::lock_guard<std::mutex> lk(qm);
                if (drain_status == RK_OK) { drain_status = RK_ERR_HIP; drain_msg = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(he); }
                w.pending = false;
            } else {
                bool ok;
                { std::lock_guard<std::mutex> lk(qm); ok = drain_status == RK_OK; }
                if (ok) {
                    try {
                        drain(w, dpool ? *dpool : none);
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(qm);
                        if (drain_status == RK_OK) { drain_status = RK_ERR_NOMEM; drain_msg = "out of host memory while moving results"; }
                        w.pending = false;
                    }
                } else w.pending = false;
            }
            t_wait += t1 - t0; t_drain += now() - t1;
            {
                std::lock_guard<std::mutex> lk(qm);
                ws_busy[wi] = false;
            }
            qcv.notify_all();
        }
    });
    // the drainer is joined on EVERY way out of this function, an exception thrown by a container or a thread constructor included
    // (a joinable std::thread that is destroyed calls std::terminate, which would take the hosting JVM down)
    struct JoinOnExit {
        std::function<void()> fn;
        ~JoinOnExit() { if (fn) fn(); }
    } join_on_exit;
    auto finish = [&]() {  // every submitted chunk drained, the thread joined
        if (!drainer.joinable()) return;
        { std::lock_guard<std::mutex> lk(qm); closing = true; }
        qcv.notify_all();
        drainer.join();
    };
    join_on_exit.fn = finish;
    // A chunk's host side (plan + stage) runs one chunk ahead of its device side (enqueue).
    struct Plan {
        uint64_t r0 = 0, r1 = 0, n = 0;
        uint32_t wpr = 0;
        unsigned wi = 0;
        size_t pb = 0;
        bool staged_async = false;
        std::atomic<uint32_t> flags{0};   // OR of the flags the host packer set
    };
    Plan plans[2];
    auto plan_chunk = [&](Plan &c, uint64_t from, unsigned no) -> int {  // bounds, record width, workspace (waits until it is free)
        uint64_t r1 = from, max_len = 0;
        if (packed_in) {
            r1 = std::min(n_reads, from + max_chunk_reads);
        } else {
            while (r1 < n_reads && r1 - from < max_chunk_reads && (seq_off[r1 + 1] - seq_off[from] <= max_chunk_bytes || r1 == from)) {
                uint64_t L = seq_off[r1 + 1] - seq_off[r1];
                if (L > max_len) max_len = L;
                r1++;
            }
        }
        if (max_len > 0x7FFFFFFFull / 8) return fail(RK_ERR_UNSUPPORTED, "%s: read longer than 2^28 symbols", who);
        c.r0 = from; c.r1 = r1; c.n = r1 - from;
        c.wpr = packed_in ? in.wpr : rk_packed_words(db, (uint32_t)max_len);
        c.wi = no % NWS;
        c.pb = c.n * c.wpr * 4;
        c.staged_async = false;
        c.flags.store(0);
        // the workspace was last used four chunks ago: its results must have left the staging buffers before it is overwritten
        std::unique_lock<std::mutex> lk(qm);
        qcv.wait(lk, [&]() { return !ws_busy[c.wi]; });
        if (drain_status != RK_OK) return fail(drain_status, "%s", drain_msg.c_str());
        return RK_OK;
    };
    // host work of a chunk that needs no HIP call: started on the worker threads, joined with pool.wait()
    auto stage_start = [&](Plan &c) -> int {
        rk_workspace &w = db->ws[c.wi];
        const uint64_t n = c.n, c0 = c.r0;
        if (host_pack) {
            // pageable characters (the usual case behind JNI): packed HERE, by the call's worker threads, straight into the
            // page-locked staging buffer -- 48 instead of 158 bytes per 150-bp read cross the link, no copy of the characters
            int rc = w.h_packed.reserve(c.pb + 8 * n, node, device);
            if (rc) return rc;
            uint32_t *hp = w.h_packed.as<uint32_t>(), *hl = hp + n * c.wpr, *hf = hl + n;
            const rk::PackSpec P = pack_spec(alpha, db->info.alphabet, db->info.bits_per_symbol, db->info.k, c.wpr);
            Plan *pc = &c;
            pool.start([=](unsigned part, unsigned parts) {
                pc->flags.fetch_or(rk::pack_reads_range(P, seq_ascii, seq_off, c0 + n * part / parts, c0 + n * (part + 1) / parts, c0, hp, hl, hf));
            });
            c.staged_async = true;
        } else if (packed_in && !in_pinned) {
            const size_t lb = in.lens ? n * 4 : 0, fb = in.flags ? n * 4 : 0, pb = c.pb;
            int rc = w.h_packed.reserve(pb + lb + fb, node, device);
            if (rc) return rc;
            const char *src = (const char *)(in.packed + c0 * c.wpr);
            char *dst = (char *)w.h_packed.p;
            pool.start([=](unsigned part, unsigned parts) {
                const size_t a = pb * part / parts, b = pb * (part + 1) / parts;
                if (b > a) memcpy(dst + a, src + a, b - a);
            });
            c.staged_async = true;
        }
        return RK_OK;
    };
    unsigned cur = 0;
    if (n_reads) {
        status = plan_chunk(plans[0], 0, 0);
        if (status == RK_OK) status = stage_start(plans[0]);
    }
    while (status == RK_OK) {
        Plan &c = plans[cur];
        const uint64_t r0 = c.r0, r1 = c.r1, n = c.n;
        const uint32_t wpr = c.wpr;
        const unsigned wi = c.wi;
        rk_workspace &w = db->ws[wi];
        hipStream_t s = w.stream;
        double t2 = now();
        if (c.staged_async) pool.wait();
        t_stage += now() - t2;
        // the next chunk's host side starts now and runs while this chunk is enqueued
        Plan &nx = plans[cur ^ 1];
        const bool more = r1 < n_reads;
#define WS_TRY(expr) do { int rc_ = (expr); if (rc_ != RK_OK) { status = rc_; goto done; } } while (0)
#define WS_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { status = fail(RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } } while (0)
        const size_t pb = c.pb;
        // packed input: the ASCII of the chunk travels only if one of its reads carries the AMBIGUOUS flag (the ambiguity kernel
        // works on characters); otherwise 38 instead of 150 bytes per 150-bp read cross the link
        bool need_ascii = !packed_in && !host_pack;
        if (host_pack) need_ascii = (c.flags.load() & RK_FLAG_AMBIGUOUS) != 0;
        if (packed_in && in.flags && seq_ascii && seq_off)
            for (uint64_t r = r0; r < r1 && !need_ascii; r++) need_ascii = (in.flags[r] & RK_FLAG_AMBIGUOUS) != 0;
        const uint64_t nbytes = need_ascii ? seq_off[r1] - seq_off[r0] : 0;
        if (need_ascii && nbytes && !(!packed_in && in_pinned)) {  // (rare: a chunk with ambiguity codes) its characters, staged by every thread
            WS_TRY(w.h_ascii.reserve(nbytes, node, device));
            const uint8_t *src = seq_ascii + seq_off[r0];
            uint8_t *dst = w.h_ascii.as<uint8_t>();
            pool.run([&](unsigned part, unsigned parts) {
                const uint64_t a = nbytes * part / parts, b = nbytes * (part + 1) / parts;
                if (b > a) memcpy(dst + a, src + a, b - a);
            });
        }
        if (more) {
            WS_TRY(plan_chunk(nx, r1, chunk_no + 1));
            WS_TRY(stage_start(nx));
        }
        t2 = now();
        WS_TRY(w.packed.reserve(pb));
        WS_TRY(w.lens.reserve(n * 4));
        WS_TRY(w.flags.reserve(n * 4));
        WS_TRY(w.nrows.reserve(n));
        WS_TRY(w.branch.reserve(n * K * 2));
        WS_TRY(w.score.reserve(n * K * 4));
        WS_TRY(w.lwr.reserve(n * K * 8));
        WS_TRY(w.oflags.reserve(n * 4));
        if (need_ascii) {
            WS_TRY(w.ascii.reserve(nbytes));
            WS_TRY(w.off.reserve((n + 1) * 8));
            WS_TRY(w.h_off.reserve((n + 1) * 8, node, device));
            uint64_t *ho = w.h_off.as<uint64_t>();
            for (uint64_t i = 0; i <= n; i++) ho[i] = seq_off[r0 + i] - seq_off[r0];
            if (!packed_in && in_pinned) {
                if (nbytes) WS_HIP(hipMemcpyAsync(w.ascii.p, seq_ascii + seq_off[r0], nbytes, hipMemcpyHostToDevice, s));
            } else if (nbytes) {
                WS_HIP(hipMemcpyAsync(w.ascii.p, w.h_ascii.p, nbytes, hipMemcpyHostToDevice, s));
            }
            WS_HIP(hipMemcpyAsync(w.off.p, w.h_off.p, (n + 1) * 8, hipMemcpyHostToDevice, s));
        }
        if (host_pack) {
            WS_HIP(hipMemcpyAsync(w.packed.p, w.h_packed.p, pb, hipMemcpyHostToDevice, s));
            WS_HIP(hipMemcpyAsync(w.lens.p, (char *)w.h_packed.p + pb, n * 4, hipMemcpyHostToDevice, s));
            WS_HIP(hipMemcpyAsync(w.flags.p, (char *)w.h_packed.p + pb + n * 4, n * 4, hipMemcpyHostToDevice, s));
        } else if (packed_in) {
            // packed records (+ lengths, flags): straight from page-locked caller memory, else through the staging buffer
            const size_t lb = in.lens ? n * 4 : 0, fb = in.flags ? n * 4 : 0;
            if (in_pinned) WS_HIP(hipMemcpyAsync(w.packed.p, in.packed + r0 * wpr, pb, hipMemcpyHostToDevice, s));
            else WS_HIP(hipMemcpyAsync(w.packed.p, w.h_packed.p, pb, hipMemcpyHostToDevice, s));
            // (lengths and flags are small: pageable copies are fine, but they must not be read after this call returns
            //  -- the staging buffer keeps them when the caller's memory is pageable)
            if (lb) {
                if (in_pinned) WS_HIP(hipMemcpyAsync(w.lens.p, in.lens + r0, lb, hipMemcpyHostToDevice, s));
                else { memcpy((char *)w.h_packed.p + pb, in.lens + r0, lb); WS_HIP(hipMemcpyAsync(w.lens.p, (char *)w.h_packed.p + pb, lb, hipMemcpyHostToDevice, s)); }
            }
            if (fb) {
                if (in_pinned) WS_HIP(hipMemcpyAsync(w.flags.p, in.flags + r0, fb, hipMemcpyHostToDevice, s));
                else { memcpy((char *)w.h_packed.p + pb + lb, in.flags + r0, fb); WS_HIP(hipMemcpyAsync(w.flags.p, (char *)w.h_packed.p + pb + lb, fb, hipMemcpyHostToDevice, s)); }
            }
        } else {
            WS_TRY(rk_pack_reads_device(db, n, w.ascii.as<uint8_t>(), w.off.as<uint64_t>(), wpr, w.packed.as<uint32_t>(),
                                        w.lens.as<uint32_t>(), w.flags.as<uint32_t>(), s));
        }
        {
            rk_result dres{w.nrows.as<uint8_t>(), w.branch.as<uint16_t>(), w.score.as<float>(), w.lwr.as<double>(), w.oflags.as<uint32_t>()};
            const uint32_t *d_lens = (!packed_in || in.lens) ? w.lens.as<uint32_t>() : nullptr;   // (host-packed chunks carry both)
            const uint32_t *d_flags = (!packed_in || in.flags) ? w.flags.as<uint32_t>() : nullptr;
            WS_TRY(rk_place_packed_device(db, p, n, w.packed.as<uint32_t>(), wpr, d_lens, packed_in ? in.fixed_len : 0, d_flags,
                                          need_ascii ? w.ascii.as<uint8_t>() : nullptr, need_ascii ? w.off.as<uint64_t>() : nullptr, &dres, s));
        }
        if (out_pinned) {
            WS_HIP(hipMemcpyAsync(out->n_rows + r0, w.nrows.p, n, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(out->branch + r0 * K, w.branch.p, n * K * 2, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(out->score + r0 * K, w.score.p, n * K * 4, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(out->lwr + r0 * K, w.lwr.p, n * K * 8, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(out->flags + r0, w.oflags.p, n * 4, hipMemcpyDeviceToHost, s));
        } else {
            WS_TRY(w.h_nrows.reserve(n, node, device));
            WS_TRY(w.h_branch.reserve(n * K * 2, node, device));
            WS_TRY(w.h_score.reserve(n * K * 4, node, device));
            WS_TRY(w.h_lwr.reserve(n * K * 8, node, device));
            WS_TRY(w.h_oflags.reserve(n * 4, node, device));
            WS_HIP(hipMemcpyAsync(w.h_nrows.p, w.nrows.p, n, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(w.h_branch.p, w.branch.p, n * K * 2, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(w.h_score.p, w.score.p, n * K * 4, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(w.h_lwr.p, w.lwr.p, n * K * 8, hipMemcpyDeviceToHost, s));
            WS_HIP(hipMemcpyAsync(w.h_oflags.p, w.oflags.p, n * 4, hipMemcpyDeviceToHost, s));
        }
#undef WS_TRY
#undef WS_HIP
        w.pending = true; w.pend_r0 = r0; w.pend_n = n;  // (page-locked caller arrays: nothing to copy, the flags are still counted)
        {
            std::lock_guard<std::mutex> lk(qm);
            ws_busy[wi] = true;
            submitted.push_back(wi);
        }
        qcv.notify_all();
        t_enq += now() - t2;
        chunk_no++;
        if (!more) break;
        cur ^= 1;
    }
done:
    pool.wait();  // (an error may leave the next chunk's staging running: it reads the caller's arrays)
    finish();
    for (rk_workspace &w : db->ws) {  // (after an error some streams may still hold work of a half-enqueued chunk)
        if (w.stream && status != RK_OK) (void)hipStreamSynchronize(w.stream);
        w.pending = false;
    }
    if (timing)
        fprintf(stderr, "%s: GPU %d on NUMA node %d, %d of this process's CPUs there, staging threads %s; %u staging + %u drain threads\n", who, device, node->node,
                node->ok ? CPU_COUNT(&node->set) : 0, node->ok ? "kept there" : "not pinned", n_stage, n_drain);
    if (timing)
        fprintf(stderr, "%s: %u chunks; submit thread: stage input %.1f ms, stage+enqueue %.1f ms; drain thread: wait for stream %.1f ms, move results %.1f ms\n",
                who, chunk_no, t_stage * 1e3, t_enq * 1e3, t_wait * 1e3, t_drain * 1e3);
    if (status == RK_OK && drain_status != RK_OK) status = fail(drain_status, "%s", drain_msg.c_str());
    if (status != RK_OK) return status;
    if (counters) *counters = ct;
    return RK_OK;
}

// What the first rk_place_batch / rk_place_batch_packed of a handle would set up on its way -- the four workspaces' streams, device
// buffers and page-locked staging buffers for full chunks of reads of up to max_read_len symbols, the launches' scratch -- done ahead
// of time (a caller does this while it is still reading its input: ~80 ms that the first batch then does not pay).
extern "C" int rk_reserve_host_path(rk_db *db, uint32_t keep_at_most, uint32_t max_read_len) {
    if (!db) return fail(RK_ERR_INVALID, "rk_reserve_host_path: null handle");
    if (keep_at_most < 1 || keep_at_most > 16) return fail(RK_ERR_INVALID, "keep_at_most=%u outside 1..16", keep_at_most);
    RK_GUARD_BEGIN
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    {
    std::lock_guard<std::mutex> lock(db->host_mutex);
    HIP_TRY(hipSetDevice(db->info.device));
    const uint64_t n = 1ull << 18, K = keep_at_most;  // (place_host's chunk)
    const uint32_t wpr = rk_packed_words(db, max_read_len ? max_read_len : 1);
    const size_t pb = (size_t)n * wpr * 4;
    const NodeCpus *node = &gpu_node_cpus(db->info.device);
    const int device = db->info.device;
    for (rk_workspace &w : db->ws) {
        if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        int rc;
        if ((rc = w.packed.reserve(pb)) || (rc = w.lens.reserve(n * 4)) || (rc = w.flags.reserve(n * 4)) || (rc = w.nrows.reserve(n)) || (rc = w.branch.reserve(n * K * 2)) ||
            (rc = w.score.reserve(n * K * 4)) || (rc = w.lwr.reserve(n * K * 8)) || (rc = w.oflags.reserve(n * 4)) || (rc = w.h_packed.reserve(pb + 8 * n, node, device)) ||
            (rc = w.h_nrows.reserve(n, node, device)) || (rc = w.h_branch.reserve(n * K * 2, node, device)) || (rc = w.h_score.reserve(n * K * 4, node, device)) ||
            (rc = w.h_lwr.reserve(n * K * 8, node, device)) || (rc = w.h_oflags.reserve(n * 4, node, device)))
            return rc;
        (void)launch_scratch(db, w.stream, 1024 + (size_t)n / 4 + 256 + (size_t)n * 5 + 512);
    }
    }
    {
        // ... and one small batch through the whole path: the runtime loads a kernel's code to the device at its first launch
        // (tens of milliseconds for the packer, the placement kernel and the tile-order pre-pass together)
        const uint64_t m = 32768;  // (the pre-pass starts at this many reads)
        const uint32_t len = std::max<uint32_t>(db->info.k, std::min<uint32_t>(max_read_len ? max_read_len : 1u, 64u));
        std::vector<uint8_t> seq((size_t)m * len, db->info.alphabet == RK_ALPHABET_DNA ? (uint8_t)'A' : (uint8_t)'R');
        std::vector<uint64_t> off(m + 1);
        for (uint64_t i = 0; i <= m; i++) off[i] = i * len;
        std::vector<uint8_t> n_rows(m);
        std::vector<uint16_t> branch(m * keep_at_most);
        std::vector<float> score(m * keep_at_most);
        std::vector<double> lwr(m * keep_at_most);
        std::vector<uint32_t> flags(m);
        rk_result res{n_rows.data(), branch.data(), score.data(), lwr.data(), flags.data()};
        rk_params p{keep_at_most, 0.01f, RK_AMB_MEAN, -INFINITY};
        return rk_place_batch(db, &p, m, seq.data(), off.data(), &res, nullptr);
    }
    RK_GUARD_END("rk_reserve_host_path")
}

extern "C" int rk_place_batch(rk_db *db, const rk_params *p, uint64_t n_reads, const uint8_t *seq_ascii,
                              const uint64_t *seq_off, rk_result *out, rk_counters *counters) {
    if (!db || !out) return fail(RK_ERR_INVALID, "rk_place_batch: null argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (n_reads && (!seq_ascii || !seq_off)) return fail(RK_ERR_INVALID, "rk_place_batch: null reads");
    HostInput in;
    in.ascii = seq_ascii; in.off = seq_off;
    RK_GUARD_BEGIN
    return place_host(db, p, n_reads, in, out, counters, "rk_place_batch");
    RK_GUARD_END("rk_place_batch")
}

extern "C" int rk_place_batch_packed(rk_db *db, const rk_params *p, uint64_t n_reads, const uint32_t *packed, uint32_t words_per_read,
                                     const uint32_t *lens, uint32_t fixed_len, const uint32_t *flags, const uint8_t *seq_ascii,
                                     const uint64_t *seq_off, rk_result *out, rk_counters *counters) {
    if (!db || !out) return fail(RK_ERR_INVALID, "rk_place_batch_packed: null argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (n_reads && (!packed || words_per_read == 0)) return fail(RK_ERR_INVALID, "rk_place_batch_packed: null packed reads");
    if (!lens && (uint64_t)fixed_len * db->info.bits_per_symbol > (uint64_t)words_per_read * 32)
        return fail(RK_ERR_INVALID, "rk_place_batch_packed: fixed_len=%u does not fit %u words", fixed_len, words_per_read);
    if ((seq_ascii == nullptr) != (seq_off == nullptr)) return fail(RK_ERR_INVALID, "rk_place_batch_packed: seq_ascii and seq_off go together");
    HostInput in;
    in.ascii = seq_ascii; in.off = seq_off; in.packed = packed; in.wpr = words_per_read; in.lens = lens; in.fixed_len = fixed_len; in.flags = flags;
    RK_GUARD_BEGIN
    return place_host(db, p, n_reads, in, out, counters, "rk_place_batch_packed");
    RK_GUARD_END("rk_place_batch_packed")
}

// AmbigSequenceKnife.initTables' char -> state part (AmbigSequenceKnife.java:103-130) on the host, for callers that would rather
// ship 2 / 5 bits per symbol over PCIe than 8: the same records, lengths and flags pack_reads_kernel produces (rk_pack_host.cpp:
// AVX2 + BMI2 blocks of 32 symbols where the machine has them, the table-driven loop otherwise).
static int pack_reads_threads(const rk::PackSpec &P, uint64_t n_reads, const uint8_t *seq_ascii, const uint64_t *seq_off, uint32_t *packed,
                              uint32_t *lens, uint32_t *flags, uint32_t n_threads) {
    ForkJoin pool(host_threads(n_reads, n_threads) - 1);
    pool.run([&](unsigned part, unsigned parts) {
        rk::pack_reads_range(P, seq_ascii, seq_off, n_reads * part / parts, n_reads * (part + 1) / parts, 0, packed, lens, flags);
    });
    return RK_OK;
}

extern "C" int rk_pack_reads_host(const rk_db *db, uint64_t n_reads, const uint8_t *seq_ascii, const uint64_t *seq_off, uint32_t words_per_read,
                                  uint32_t *packed, uint32_t *lens, uint32_t *flags, uint32_t n_threads) {
    if (!db || !seq_off || !packed || !lens || !flags || words_per_read == 0) return fail(RK_ERR_INVALID, "rk_pack_reads_host: null/zero argument");
    if (n_reads && !seq_ascii && seq_off[n_reads]) return fail(RK_ERR_INVALID, "rk_pack_reads_host: null reads");
    RK_GUARD_BEGIN
    Alphabet A;
    build_alphabet(db->info.alphabet, db->convert_uo != 0, A);
    return pack_reads_threads(pack_spec(A, db->info.alphabet, db->info.bits_per_symbol, db->info.k, words_per_read), n_reads, seq_ascii, seq_off,
                              packed, lens, flags, n_threads);
    RK_GUARD_END("rk_pack_reads_host")
}

// The same without a database handle (and without a GPU): alphabet, --convertUO switch and k are all the packer needs.
extern "C" int rk_pack_reads(uint32_t alphabet, int convert_uo, uint32_t k, uint64_t n_reads, const uint8_t *seq_ascii, const uint64_t *seq_off,
                             uint32_t words_per_read, uint32_t *packed, uint32_t *lens, uint32_t *flags, uint32_t n_threads) {
    if (alphabet != RK_ALPHABET_DNA && alphabet != RK_ALPHABET_AA) return fail(RK_ERR_INVALID, "rk_pack_reads: alphabet must be 4 (DNA) or 20 (amino acids)");
    if (!seq_off || !packed || !lens || !flags || words_per_read == 0 || k == 0) return fail(RK_ERR_INVALID, "rk_pack_reads: null/zero argument");
    if (n_reads && !seq_ascii && seq_off[n_reads]) return fail(RK_ERR_INVALID, "rk_pack_reads: null reads");
    RK_GUARD_BEGIN
    Alphabet A;
    build_alphabet(alphabet, convert_uo != 0, A);
    return pack_reads_threads(pack_spec(A, alphabet, alphabet == RK_ALPHABET_DNA ? 2u : 5u, k, words_per_read), n_reads, seq_ascii, seq_off, packed,
                              lens, flags, n_threads);
    RK_GUARD_END("rk_pack_reads")
}

#include "rk_synth_impl.h"
#include "rk_image_impl.h"

#ifdef RK_STAMPS
// diagnostic builds only (scripts/stamps.py)
extern "C" int rk_debug_read_stamps(unsigned long long *out, int n_waves) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rk::rk_stamp_buf), (size_t)n_waves * 16 * sizeof(unsigned long long)));  // (n_waves = 8192 reads both halves)
    return RK_OK;
}
#endif
