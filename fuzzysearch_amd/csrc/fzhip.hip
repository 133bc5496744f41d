// fzhip.hip — host side of libfzhip.so: the C-ABI of include/fzhip.h over the gfx950 kernels.
//
// Data layout in HBM (per device shard):
//   [FZ_PAD_FRONT zero bytes][buf_len sequence bytes][zero bytes up to a whole tile + FZ_PAD_BACK]
// so the scan can read whole 16 KiB tiles and 8-byte halos without bounds checks; zero padding can
// never create an accepted hit because every candidate is range-checked against the global length.
//   d_out  : [1 KiB header: counters][records, 24 B each] in one allocation, so the usual result
//            comes back in ONE D2H copy (header + about as many records as the previous call had)
//              counters[0]      hits emitted to d_hits (non-fused paths)
//              counters[1]      records written
//              counters[2]      automaton work items whose candidate lists overflowed
//              counters[8..71]  confirmed n-gram hits of the fused scan (64 words: one word only
//                               sustains ~90 atomics/us)
//   d_hits : uint64 (block << 56 | global idx) hit list — only the non-fused paths use it
// The GPU produces records unordered; the host puts them in the reference's emission order by a radix
// sort on (block, idx) — the number of records is tiny next to the bytes scanned.
// Every search is synchronous: when a fz_* call returns its results are on the host and nothing is
// in flight on the sequence.  Buffers that turn out too small are grown and the search re-runs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>                             // types only: the library is dlopen-ed on first use (rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>
#include <map>
#include <mutex>
#include <vector>
#include <cassert>

#include "../../include/fzhip.h"
#include "fz_kernels.h"

namespace {

// Every process-wide switch of the library: read from the environment ONCE, on first use, into this struct.  None of them is
// needed to use the library; tests, A/B measurements and lab builds set them (INTEGRATION.md §5 describes each).  Results
// never depend on a switch — the -m gpu suite runs the same cases under the ones marked (t) and compares with the oracle.
struct Switches {
    // (t) alternative forms of the same computation
    bool no_direct = false;            // FZ_NO_DIRECT: records / counters through D2H copies, never the kernels' stores into pinned memory
    bool no_slot_and = false;          // FZ_NO_SLOT_AND: the filter's general slot form (shift + and) for every launch
    int max_blocks = 0;                // FZ_MAX_BLOCKS=n: at most n n-gram blocks per scan launch
    bool force_big_verify = false;     // FZ_FORCE_BIG_VERIFY: every stand-alone verification by fz_verify_big_kernel
    bool no_wavefront = false;         // FZ_NO_WAVEFRONT: lane-per-candidate verification where lane-per-cell would run
    bool no_wf_fuse = false;           // FZ_NO_WF_FUSE: budgets 5 .. 15 verified by the stand-alone kernel, not inside the scan
    int wf32 = -1;                     // FZ_WF32=0 / 1: pin the fused lane-per-cell form off / on for budgets 8 .. 15 (default: by density)
    bool no_bits = false;              // FZ_NO_BITS: no bit-vector verification (Levenshtein budgets >= FZ_BITS_MIN_K take round 5's forms)
    int bits_min_k = 3;                // FZ_BITS_MIN_K=k: smallest Levenshtein budget always verified by bit-vector columns (below: the register
                                       // band unless the pattern lets expect dense candidates)
    int bits_qcap = 0;                 // FZ_BITS_QCAP=n: queue entries per wave of the bit-vector form (default: by LDS)
    int bits_lds_kb = 0;               // FZ_BITS_LDS_KB: LDS per scan workgroup the bit-vector form may take (default 26)
    bool gen_legacy = false;           // FZ_GEN_LEGACY: the generic automaton as fz_lp_kernel (one wave per hit, round 3's form)
    bool gen_no_dedup = false;         // FZ_GEN_NO_DEDUP: no window table (every n-gram hit runs its automaton)
    bool gen_direct = false;           // FZ_GEN_DIRECT: automaton records stored straight into pinned host memory (round 1)
    bool gen_host_order = false;       // FZ_GEN_HOST_ORDER: the generic rows ordered by the host, not by fz_gen_order / scatter kernels
    int gh_waves = 0;                  // FZ_GH_WAVES=1 / 2 / 4: waves per hit of fz_gen_hit_kernel
    bool gh_no_bits = false;           // FZ_GH_NO_BITS: round 4's candidate step instead of the bit-parallel one
    int cand_lds_max = 0;              // FZ_CAND_LDS_MAX=n: candidate lists beyond n entries live in HBM
    bool group_best_exact = false;     // FZ_GROUP_BEST_EXACT: fz_group_best by the exact walk only
    bool no_dev_threads = false;       // FZ_NO_DEV_THREADS: a multi-device context drives every device from the calling thread
    int taper_steps = 4;               // FZ_TAPER_STEPS (0: no regions), FZ_TAPER_MIN, FZ_TAPER_WG_PER_CU: the scan grid's last round
    double taper_min = 0.25;
    int taper_wg_per_cu = 7;
    bool no_rccl = false;              // FZ_NO_RCCL: behave like an install without librccl
    std::string rccl_lib;              // FZ_RCCL_LIB=path: the collective library to dlopen first (tests: tests/libmock_rccl.so)
    std::string rocm_path = "/opt/rocm";   // ROCM_PATH
    // measurement / lab knobs (same results; they move work or change launch shapes)
    bool trace = false;                // FZ_TRACE: per-phase host timings of a search on stderr
    bool no_timing = false;            // FZ_NO_TIMING: contexts start without hipEvent timing of their kernels
    bool dual_stream = false;          // FZ_DUAL_STREAM: contexts start with fz_set_streams(2)
    bool no_ext_launch = false;        // FZ_NO_EXT_LAUNCH: events recorded behind the kernels instead of on their dispatch packets
    bool gen_hi_stream = false;        // FZ_GEN_HI_STREAM: the generic automaton on a high-priority stream of its lane
    bool no_spin = false;              // FZ_NO_SPIN: no hipDeviceScheduleSpin
    bool stream_default_priority = false;   // FZ_STREAM_DEFAULT_PRIORITY: the scan streams at default instead of lowest priority
    bool stream_nofill = false;        // FZ_STREAM_NOFILL: file streams without the reads (H2D + scan alone)
    bool stream_trace = false;         // FZ_STREAM_TRACE: host time split of a file stream on stderr
    long worker_spin_us = 1500;        // FZ_WORKER_SPIN_US: how long a device's worker thread spins before it sleeps
    long comm_timeout_ms = 60000;      // FZ_COMM_TIMEOUT_MS: deadline of every wait for a collective (then FZ_ETIMEOUT, not a hung job)
    int fused_lds_kb = 0, fused_target_kb = 0, extra_lds_kb = 0;   // FZ_FUSED_LDS_KB, FZ_FUSED_TARGET_KB, FZ_EXTRA_LDS_KB: LDS shaping of the scan
    int tiles_per_wg = 0, rounds = 0, wg_per_cu = 0;               // FZ_TILES_PER_WG, FZ_ROUNDS, FZ_WG_PER_CU: the scan grid
    int lp_grid_per_cu = 24, gh_grid_per_cu = 0;                   // FZ_LP_GRID_PER_CU, FZ_GH_GRID_PER_CU: automaton grids
};

Switches read_switches() {
        Switches v;
        auto flag = [](const char *n) { return getenv(n) != nullptr; };
        auto num = [](const char *n, int dflt) { const char *e = getenv(n); return e ? atoi(e) : dflt; };
        v.no_direct = flag("FZ_NO_DIRECT"); v.no_slot_and = flag("FZ_NO_SLOT_AND"); v.max_blocks = num("FZ_MAX_BLOCKS", 0);
        v.force_big_verify = flag("FZ_FORCE_BIG_VERIFY"); v.no_wavefront = flag("FZ_NO_WAVEFRONT"); v.no_wf_fuse = flag("FZ_NO_WF_FUSE");
        v.no_bits = flag("FZ_NO_BITS"); v.bits_min_k = num("FZ_BITS_MIN_K", 3); v.bits_qcap = num("FZ_BITS_QCAP", 0);
        v.bits_lds_kb = num("FZ_BITS_LDS_KB", 0);
        v.wf32 = num("FZ_WF32", -1); v.gen_legacy = flag("FZ_GEN_LEGACY"); v.gen_no_dedup = flag("FZ_GEN_NO_DEDUP");
        v.gen_direct = flag("FZ_GEN_DIRECT"); v.gen_host_order = flag("FZ_GEN_HOST_ORDER"); v.gh_waves = num("FZ_GH_WAVES", 0);
        v.gh_no_bits = flag("FZ_GH_NO_BITS"); v.cand_lds_max = num("FZ_CAND_LDS_MAX", 0); v.group_best_exact = flag("FZ_GROUP_BEST_EXACT");
        v.no_dev_threads = flag("FZ_NO_DEV_THREADS"); v.taper_steps = num("FZ_TAPER_STEPS", 4);
        if (const char *e = getenv("FZ_TAPER_MIN")) v.taper_min = atof(e);
        v.taper_wg_per_cu = num("FZ_TAPER_WG_PER_CU", 7); v.no_rccl = flag("FZ_NO_RCCL");
        if (const char *e = getenv("FZ_RCCL_LIB")) v.rccl_lib = e;
        if (const char *e = getenv("ROCM_PATH")) v.rocm_path = e;
        v.trace = flag("FZ_TRACE"); v.no_timing = flag("FZ_NO_TIMING"); v.dual_stream = num("FZ_DUAL_STREAM", 0) != 0;
        v.no_ext_launch = flag("FZ_NO_EXT_LAUNCH"); v.gen_hi_stream = flag("FZ_GEN_HI_STREAM"); v.no_spin = flag("FZ_NO_SPIN");
        v.stream_default_priority = flag("FZ_STREAM_DEFAULT_PRIORITY"); v.stream_nofill = flag("FZ_STREAM_NOFILL");
        v.stream_trace = flag("FZ_STREAM_TRACE");
        if (const char *e = getenv("FZ_WORKER_SPIN_US")) v.worker_spin_us = atol(e);
        if (const char *e = getenv("FZ_COMM_TIMEOUT_MS")) v.comm_timeout_ms = std::max(1L, atol(e));
        v.fused_lds_kb = num("FZ_FUSED_LDS_KB", 0); v.fused_target_kb = num("FZ_FUSED_TARGET_KB", 0); v.extra_lds_kb = num("FZ_EXTRA_LDS_KB", 0);
        v.tiles_per_wg = num("FZ_TILES_PER_WG", 0); v.rounds = num("FZ_ROUNDS", 0); v.wg_per_cu = num("FZ_WG_PER_CU", 0);
        v.lp_grid_per_cu = num("FZ_LP_GRID_PER_CU", 24); v.gh_grid_per_cu = num("FZ_GH_GRID_PER_CU", 0);
        return v;
}

Switches &switches_storage() { static Switches s = read_switches(); return s; }
const Switches &sw() { return switches_storage(); }

// FZ_TRACE=1: per-phase host timings of a search call on stderr (tuning aid)
struct Trace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    Trace() : on(sw().trace) { t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fz] %-18s %8.1f us (total %8.1f)\n", what,
                std::chrono::duration<double, std::micro>(now - last).count(),
                std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};

thread_local std::string g_err;

// FZ_NO_DIRECT=1 (test / lab knob): records and counters always come back through a D2H copy, never by the kernels' own
// stores into the pinned staging buffer
bool env_no_direct() { return sw().no_direct; }

int fail(int code, const char *fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    g_err = tmp;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? FZ_ENOMEM : FZ_EDEVICE, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

constexpr uint64_t kHeaderBytes = 1024;          // counters[0] = emitted hits, [1] = records, [8..71] = confirmed-hit tallies
constexpr uint64_t kFirstCopyRecs = 4096;        // records fetched together with the header
constexpr uint64_t kHostRecs = 16384;            // records the pinned staging buffer holds (direct mode)
static_assert(kHeaderBytes == FZ_HDR_WORDS * 8, "header layout");

struct DevState {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t *d_hits = nullptr;
    uint64_t hit_cap = 0;
    uint8_t *d_out = nullptr;                    // [header kHeaderBytes][recs]
    uint64_t rec_cap = 0;
    uint8_t *h_stage = nullptr;                  // pinned, kHeaderBytes + kHostRecs * sizeof(FzRec)
    uint8_t *h_stage_dev = nullptr;              // the same memory as the device addresses it
    // Direct mode: the kernels write records and (last workgroup) the counters straight into h_stage,
    // no D2H copy command.  Left when a search produces more than kHostRecs records, re-entered when
    // the counts are small again.
    bool direct = true;
    bool last_direct = false;                    // mode of the search being collected
    // Large record sets of the automaton kernels (10^5 .. 10^6 records): a pinned, device-mapped host
    // buffer that grows on demand; the kernel's stores cross PCIe while it runs instead of a D2H copy
    // into pageable memory afterwards.
    uint8_t *d_cand = nullptr;                   // HBM candidate lists of the automaton kernels (rare fallback)
    uint64_t cand_bytes = 0;
    // generic search ordered on the device: per-hit {first row, row count} and the finished fz_match rows
    uint8_t *d_gen_order = nullptr;
    uint8_t *d_gen_rows = nullptr;
    uint8_t *d_gen_dedup = nullptr;               // window table of the generic search (FzGenDedup, fz_device.h)
    bool dedup_zeroed = false;                   // ... zeroed behind the last search that used it
    uint64_t gen_dedup_arg = 0;                  // ... its address for the kernels of the search being launched (0: no table)
    bool dedup_used = false;                     // the search being collected ran with it (row count = counters[FZ_HDR_GEN_ROWS])
    bool gen_multi_used = false;                 // ... and its automaton as fz_gen_hit_kernel (counters[FZ_HDR_GEN_FAIL]: hits it gave up on)
    uint64_t gen_rows_cap = 0;                   // rows
    uint8_t *h_big = nullptr, *h_big_dev = nullptr;
    uint64_t big_cap = 0;                        // records
    // one recycled sequence allocation (chunked file reads upload / release 1 MiB buffers in a loop;
    // hipMalloc + hipFree per chunk would dominate)
    uint8_t *spare_alloc = nullptr;
    uint64_t spare_bytes = 0;
    uint64_t first_copy = 512;                   // records fetched with the header (tracks the last count)
    bool header_zeroed = false;                  // the counters were already zeroed after the last D2H copy
    bool verify_launched = false;                // the last enqueue ran fz_verify_kernel (ev[2] recorded)
    int scan_end_event = 1;                      // which event marks the end of the last scan (1 or 3)
    int verify_end_event = 2;                    // ... and of the verification kernel behind it (2, or 3 = the completion event)
    // staging of the last closed file stream, kept for the next one (pinning 2 x 65 MiB costs ~30 ms)
    uint8_t *stream_h[2] = {nullptr, nullptr};
    uint8_t *stream_d = nullptr;
    uint64_t stream_cap = 0, stream_d_bytes = 0;
    int n_cus = 256;
    // what the search being collected ran with (an overflow is judged against these, not against buffers
    // that a search collected in between may have grown)
    uint64_t hit_cap_used = 0, rec_cap_used = 0;
    bool fused_used = false;
    uint32_t form_used = 0;                      // FZ_FORM_* of the last enqueue
    bool wf32_candidate = false;                 // the search in this slot could have verified inside the scan with 32 lanes per candidate
    bool timed = true;                           // the search being collected recorded its start event
    double last_filter_ms = 0;                   // scan span of the search collected last on this device (fz_device_ms)
    uint64_t fold_guess = 8192, fold_copied = 0; // folded generic search: pairs fetched with the counters
    bool fold_direct = true;                     // ... or written straight into h_stage by the automaton kernel (while they fit)
    bool fold_was_direct = false;                // mode of the folded search being collected
    int lp_end_event = 2;                        // which event marks the end of the last automaton kernel (2, or 3 = the completion)
    hipStream_t stream_hi = nullptr;             // generic searches in flight: the automaton and what follows it (high priority)
    hipEvent_t ev_scan_done = nullptr;
    uint8_t *d_pat = nullptr;                    // pattern in HBM (subsequences longer than FZ_MAX_M, fz_verify_big_kernel)
    uint64_t pat_cap = 0;
    int slot_id = 0;                             // which of the two result slots is the current one
    uint32_t launches_used = 0;                  // scan launches of the last enqueue on this device
    // Two fused searches in flight (result slot 1): the younger one's scan on a stream and a counter block of its own, so
    // that it starts while the older scan drains instead of behind it (FZ_DUAL_STREAM=0: one stream, as before)
    hipStream_t stream_alt = nullptr;
    uint8_t *d_hdr_alt = nullptr;
    // RCCL (fz_comm_*): this device state is rank comm_rank of a communicator.  A search of such a context leaves
    // its counters + records in d_out; a device-to-device snapshot (d_send[slot], taken on the scan stream right
    // behind the kernels, so a younger search may reuse d_out) is what the all-gather sends.
    ncclComm_t comm = nullptr;
    int comm_rank = -1;
    hipStream_t comm_stream = nullptr;           // default priority: the collective runs next to the (low-priority) scan
    uint8_t *d_send[2] = {nullptr, nullptr};
    uint64_t send_cap = 0;                       // records a snapshot holds
    uint8_t *d_recv = nullptr, *h_recv = nullptr;
    uint64_t recv_bytes = 0;
    hipEvent_t ev_snap[2] = {nullptr, nullptr}, ev_done = nullptr;
    bool snap_taken[2] = {false, false};
    // The second result slot: fz_lev_ngrams_begin with one search already in flight launches into it, so
    // that the host orders the records of search i while search i + 1 scans (two-deep pipeline).
    struct Slot {
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        uint8_t *h_stage = nullptr, *h_stage_dev = nullptr;
        bool last_direct = false, verify_launched = false, fused_used = false, timed = true, wf32_candidate = false;
        int scan_end_event = 1, verify_end_event = 2;
        uint64_t hit_cap_used = 0, rec_cap_used = 0;
        int slot_id = 1;
    } other;
    void swap_slot() {
        std::swap(slot_id, other.slot_id);
        for (int i = 0; i < 4; ++i) std::swap(ev[i], other.ev[i]);
        std::swap(h_stage, other.h_stage);
        std::swap(h_stage_dev, other.h_stage_dev);
        std::swap(last_direct, other.last_direct);
        std::swap(verify_launched, other.verify_launched);
        std::swap(timed, other.timed);
        std::swap(fused_used, other.fused_used);
        std::swap(wf32_candidate, other.wf32_candidate);
        std::swap(scan_end_event, other.scan_end_event);
        std::swap(verify_end_event, other.verify_end_event);
        std::swap(hit_cap_used, other.hit_cap_used);
        std::swap(rec_cap_used, other.rec_cap_used);
    }
};

struct Shard {
    int dev = 0;                                 // index into ctx->devs
    uint8_t *d_alloc = nullptr;                  // allocation base
    uint8_t *d_buf = nullptr;                    // = d_alloc + FZ_PAD_FRONT
    uint64_t alloc_bytes = 0;
    FzGeom geom{};
};

}  // namespace

namespace {

// One host thread per device of a multi-device context (round 4).  A search over N shards used to be driven by the
// calling thread alone: N x ~10 us of launches one after the other, N waits, and the ordering of all N record lists
// (8 x 10^4 records at N = 8: 0.35 ms) against a 0.79 ms kernel.  Every device's worker now enqueues, waits for,
// collects AND orders its own shard; the caller only merges the ordered rows block by block.  Workers spin for a
// while after a job (the next one of a pipelined loop arrives within a millisecond), then sleep on a condition variable.
// FZ_NO_DEV_THREADS=1: no workers, the calling thread does everything in shard order (A/B knob).
struct DevWorkers {
    struct W {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        std::atomic<uint64_t> posted{0}, done{0};
        int rc = 0;
        std::string err;
        bool stop = false;
    };
    std::vector<W *> ws;
    long spin_us = 1500;

    explicit DevWorkers(size_t n) {
        spin_us = sw().worker_spin_us;
        for (size_t i = 0; i < n; ++i) {
            W *w = new W();
            ws.push_back(w);
            w->th = std::thread([this, w]() { run(w); });
        }
    }
    ~DevWorkers() {
        for (W *w : ws) {
            { std::lock_guard<std::mutex> g(w->mu); w->stop = true; }
            w->cv.notify_one();
            w->th.join();
            delete w;
        }
    }
    void run(W *w) {
        uint64_t seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (w->posted.load(std::memory_order_acquire) == seen) {
                if ((++spins & 255u) == 0 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) {
                    std::unique_lock<std::mutex> lk(w->mu);
                    w->cv.wait(lk, [&]() { return w->stop || w->posted.load(std::memory_order_acquire) != seen; });
                    if (w->stop) return;
                    break;
                }
                __builtin_ia32_pause();
            }
            seen = w->posted.load(std::memory_order_acquire);
            // a job that throws (std::bad_alloc out of a result vector's resize) must not take the process down with
            // std::terminate, nor leave wait() spinning on a worker that is gone: the error travels like any other
            try {
                w->rc = w->job();
                if (w->rc) w->err = g_err;
            } catch (const std::bad_alloc &) {
                w->rc = FZ_ENOMEM; w->err = "out of memory on a device worker thread";
            } catch (const std::exception &e) {
                w->rc = FZ_EDEVICE; w->err = std::string("exception on a device worker thread: ") + e.what();
            }
            w->done.store(seen, std::memory_order_release);
        }
    }
    void post(size_t i, std::function<int()> fn) {
        W *w = ws[i];
        // one job at a time per worker (the slot is overwritten below): callers wait for a job before they post the next; should
        // one ever not, the post waits here instead of corrupting the slot (rounds 4-5: an assert, i.e. abort() of the host process)
        while (w->done.load(std::memory_order_acquire) != w->posted.load(std::memory_order_relaxed)) std::this_thread::yield();
        w->job = std::move(fn);
        w->posted.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> g(w->mu); }
        w->cv.notify_one();
    }
    int wait(size_t i) {
        W *w = ws[i];
        const uint64_t want = w->posted.load(std::memory_order_relaxed);
        unsigned spins = 0;
        while (w->done.load(std::memory_order_acquire) != want) {
            if ((++spins & 1023u) == 0) std::this_thread::yield(); else __builtin_ia32_pause();
        }
        if (w->rc) g_err = w->err;
        return w->rc;
    }
};

// What collecting one shard of a multi-shard search leaves behind (one per shard, kept between searches).
struct ShardOut {
    std::vector<FzRec> recs;
    std::vector<uint64_t> hits;
    std::vector<fz_match> rows;                  // the shard's records in the reference's order (multi-shard searches)
    bool rerun = false;
    uint64_t nh = 0, nr = 0, bytes = 0;
    bool has_tref = false;
    DevState *td = nullptr;
    hipEvent_t tev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace

struct fz_ctx {
    std::vector<DevState> devs;
    fz_stats_t stats{};
    bool last_fused = false;
    // Candidate slots per list of the per-hit automaton kernel.  Small lists let every wave of the launch
    // be resident at once (256 slots: 8 KB of LDS per wave, 19 waves per CU; 1024: 7 per CU); a search
    // that overflows them re-runs with 4x the slots and the context remembers.
    uint32_t gen_cand_cap = 256;
    // The per-hit automaton runs as fz_gen_hit_kernel (a workgroup of 2 or 4 waves per hit).  A search in which a hit
    // outgrows a wave's share of the candidate list or its match buffer is run again with fz_lp_kernel (one wave per hit),
    // and so are the next gen_multi_skip searches of the context (1, 2, 4 .. 64 after consecutive failures; a success
    // resets the back-off): inputs that always fail (dense repeats) pay a wasted launch now and then, not every time.
    bool gen_multi = !sw().gen_legacy;
    uint32_t gen_multi_skip = 0, gen_multi_backoff = 0;
    // Single-shard searches in direct mode leave their records in the pinned staging buffer and only
    // publish a view of them (valid until the next search of this context): saves a 24 B x nr memcpy.
    const FzRec *view = nullptr;
    uint64_t view_n = 0;
    const FzGenRec *gen_view = nullptr;          // same for the per-hit automaton's records (pinned h_big)
    uint64_t gen_view_n = 0;
    const uint8_t *gen_rows_dev = nullptr;       // ... or the finished rows, ordered on the device (DevState::d_gen_rows)
    uint64_t gen_rows_n = 0;
    int gen_rows_device = 0;
    struct fz_stream *stream_inflight = nullptr;  // a file stream whose batch is on the device (other searches are refused)
    // fz_lev_ngrams_begin .. _end: up to two searches in flight, collected in launch order (pend[0] is the
    // oldest and owns the devices' current result slot, pend[1] their second slot)
    struct Pending {
        fz_seq *seq = nullptr;
        std::vector<uint8_t> pattern;
        uint32_t m = 0, k = 0;
        bool launched = false;                       // false: deferred until the older search has been collected
        uint32_t kind = FZ_MODE_LEV;                 // FZ_MODE_LEV / FZ_MODE_SUBS (result slots of lane 0) or FZ_MODE_GENERIC (lanes)
        uint32_t max_subs = 0, max_ins = 0, max_dels = 0;
        bool consolidated = false;                   // generic: deliver fz_generic_ngrams_consolidated's rows
        int lane = 0;                                // generic: which lane carries it
    } pend[2];
    int npend = 0;
    // Lanes: a second complete set of per-device state (stream, hit list, counters + records, ordering areas, staging).
    // A generic search in flight keeps its hit list and records on the device until it is collected, so the next one
    // needs buffers and a stream of its own: then its scan runs next to the older search's automaton kernel.
    std::vector<DevState> devs2;
    int lane = 0;                                // the lane the code below works on (0 except inside generic begin / end)
    // sequences still resident (fz_destroy frees what the caller did not release)
    std::vector<fz_seq *> live;
    // RCCL: number of ranks of the communicator this context joined (0: none) and whether its Levenshtein n-gram
    // searches are collective (every rank gets the merged global stream)
    bool any_found = false;                      // result of the last has_near_match_* (fz_*_any) search
    // sharded searches: where every shard's (rank's) records end in the collected vector, and the shards in ascending
    // order of the index range they own (emit_matches orders shard by shard)
    std::vector<size_t> seg_ends;
    std::vector<uint32_t> seg_order;
    // hipEvent timing of the kernels (fz_stats: filter_ms / verify_ms / device_ms).  One event record is one more packet
    // in front of the kernel and two hipEventElapsedTime calls behind it: fz_set_timing(ctx, 0) drops them.
    bool timing = !sw().no_timing;
    // The spans are read from the events only when somebody asks (fz_stats / fz_device_ms): a hipEventElapsedTime call
    // costs ~5 us of host time, two or three of them sat between the completion of every search and its result.  The
    // references die with the next launch of the context (whose enqueue re-records the events).
    struct TimingRef { DevState *d; hipEvent_t f0, f1, v0, v1, t0, t1; };
    std::vector<TimingRef> tref;
    int comm_world = 0;
    bool snapshot = false;
    bool comm_broken = false;                    // a collective ran into its deadline: the communicator is abandoned (no further
                                                 // collective is started, its streams and buffers are not waited for or freed)
    uint64_t gcap = 4096;                        // records per rank the all-gather carries (follows the counts, on all ranks alike)
    // fz_set_streams: 2 = the younger of two fused searches in flight scans on a stream of its own (FZ_DUAL_STREAM=1 presets it)
    int streams = sw().dual_stream ? 2 : 1;
    double last_gather_ms = 0;                   // host time of the last search's exchange step (all-gather + D2H + parse)
    // multi-device contexts: one host thread per device (enqueue, wait, collect and order its shard), and what the
    // shards of the search being collected left (rows_ready: every shard's rows are ordered, emit_matches only merges)
    DevWorkers *workers = nullptr;
    std::vector<ShardOut> souts;
    bool rows_ready = false;
};

struct fz_seq {
    fz_ctx *ctx = nullptr;
    uint64_t n = 0;                              // global length
    std::vector<Shard> shards;
    // collective searches: the first index every rank of the communicator owns of THIS sequence (~0: nothing), exchanged
    // on the sequence's first collective search; the ranks' segments are merged in that order
    std::vector<uint64_t> rank_lo;
};

namespace {
void comm_teardown(fz_ctx *ctx);
inline DevState &lane_dev(fz_ctx *ctx, int dev) { return ctx->lane ? ctx->devs2[dev] : ctx->devs[dev]; }

// fz_stats / fz_device_ms: the kernel spans of the search collected last, read from its events now.
void resolve_timing(fz_ctx *ctx) {
    for (const fz_ctx::TimingRef &r : ctx->tref) {
        float f = 0, v = 0, t = 0;
        if (hipSetDevice(r.d->device) != hipSuccess) continue;
        if (r.f0 && hipEventElapsedTime(&f, r.f0, r.f1) != hipSuccess) f = 0;
        if (r.v0 && hipEventElapsedTime(&v, r.v0, r.v1) != hipSuccess) v = 0;
        if (r.t0 && hipEventElapsedTime(&t, r.t0, r.t1) != hipSuccess) t = 0;
        (void)hipGetLastError();
        r.d->last_filter_ms = f;
        ctx->stats.filter_ms = std::max<double>(ctx->stats.filter_ms, f);
        ctx->stats.verify_ms = std::max<double>(ctx->stats.verify_ms, v);
        ctx->stats.device_ms = std::max<double>(ctx->stats.device_ms, t);
    }
    ctx->tref.clear();
}
}

namespace {

int ensure_hits(DevState &d, uint64_t cap) {
    if (d.hit_cap >= cap) return FZ_OK;
    HIP_TRY(hipSetDevice(d.device));
    if (d.d_hits) { HIP_TRY(hipFree(d.d_hits)); d.d_hits = nullptr; d.hit_cap = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_hits), cap * sizeof(uint64_t)));
    d.hit_cap = cap;
    return FZ_OK;
}

constexpr uint32_t kCandLdsMax = 4096;            // candidate slots per list that still live in LDS
constexpr uint32_t kCandMax = 1u << 18;           // ... and in the HBM fallback (4 MiB per workgroup)
constexpr unsigned kCandScratchGrid = 256;        // workgroups of a launch that uses the HBM lists

// LDS bytes and (if the lists do not fit LDS) the HBM scratch of one automaton launch.
int cand_lists(DevState &d, uint32_t cand_cap, size_t fixed_lds, size_t &lds, uint64_t &scratch) {
    if (fixed_lds > 150 * 1024) return fail(FZ_EUNSUPPORTED, "subsequence + window too long for the automaton kernel's LDS (%zu bytes)", fixed_lds);
    lds = fixed_lds + 2 * (size_t)cand_cap * sizeof(FzGCand);
    scratch = 0;
    const uint32_t lds_max = sw().cand_lds_max > 0 ? (uint32_t)sw().cand_lds_max : kCandLdsMax;   // (test knob: the HBM lists at small sizes)
    if (cand_cap <= lds_max && lds <= 160 * 1024) return FZ_OK;
    if (cand_cap > kCandMax) return fail(FZ_EUNSUPPORTED, "automaton candidate sets beyond %u entries", kCandMax);
    lds = fixed_lds;
    const uint64_t need = (uint64_t)kCandScratchGrid * 2 * cand_cap * sizeof(FzGCand);
    if (d.cand_bytes < need) {
        HIP_TRY(hipSetDevice(d.device));
        if (d.d_cand) { HIP_TRY(hipFree(d.d_cand)); d.d_cand = nullptr; d.cand_bytes = 0; }
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_cand), need));
        d.cand_bytes = need;
    }
    scratch = reinterpret_cast<uint64_t>(d.d_cand);
    return FZ_OK;
}

int ensure_big(DevState &d, uint64_t cap) {
    if (d.big_cap >= cap) return FZ_OK;
    HIP_TRY(hipSetDevice(d.device));
    if (d.h_big) { HIP_TRY(hipStreamSynchronize(d.stream)); HIP_TRY(hipHostFree(d.h_big)); d.h_big = nullptr; d.big_cap = 0; }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&d.h_big), cap * sizeof(FzRec), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&d.h_big_dev), d.h_big, 0));
    d.big_cap = cap;
    return FZ_OK;
}

// Ordering area and row buffer of the device-ordered generic search (rows follow the record capacity).
int ensure_gen_rows(DevState &d) {
    HIP_TRY(hipSetDevice(d.device));
    if (!d.d_gen_order) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_gen_order), (size_t)FZ_GEN_ORDER_MAX * 12));
    if (d.gen_rows_cap >= d.rec_cap) return FZ_OK;
    if (d.d_gen_rows) { HIP_TRY(hipFree(d.d_gen_rows)); d.d_gen_rows = nullptr; d.gen_rows_cap = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_gen_rows), d.rec_cap * sizeof(FzOutRow)));
    d.gen_rows_cap = d.rec_cap;
    return FZ_OK;
}

// Snapshot buffers of a communicator's device state: at least `cap` records each; contents are kept (a snapshot
// that is waiting for its all-gather survives the growth).
int ensure_send(DevState &d, uint64_t cap) {
    if (d.send_cap >= cap && d.d_send[0]) return FZ_OK;
    HIP_TRY(hipSetDevice(d.device));
    for (int i = 0; i < 2; ++i) {
        uint8_t *nb = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&nb), kHeaderBytes + cap * sizeof(FzRec)));
        if (d.d_send[i]) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(nb, d.d_send[i], kHeaderBytes + d.send_cap * sizeof(FzRec), hipMemcpyDeviceToDevice));
            HIP_TRY(hipFree(d.d_send[i]));
        } else {
            HIP_TRY(hipMemset(nb, 0, kHeaderBytes));
        }
        d.d_send[i] = nb;
    }
    d.send_cap = cap;
    return FZ_OK;
}

int ensure_recs(DevState &d, uint64_t cap) {
    if (d.rec_cap >= cap) return FZ_OK;
    HIP_TRY(hipSetDevice(d.device));
    if (d.d_out) { HIP_TRY(hipFree(d.d_out)); d.d_out = nullptr; d.rec_cap = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_out), kHeaderBytes + cap * sizeof(FzRec)));
    d.rec_cap = cap;
    d.header_zeroed = false;
    return FZ_OK;
}

uint32_t load_le32(const uint8_t *p, uint32_t avail) {
    uint32_t v = 0;
    for (uint32_t i = 0; i < 4 && i < avail; ++i) v |= (uint32_t)p[i] << (8 * i);
    return v;
}

// Describes one whole search as a list of n-gram blocks.
struct BlockPlan {
    uint32_t L = 0;
    std::vector<uint32_t> s;       // ngram_start per block (the accepted hit range follows from it: fz_block_range)
    uint64_t abs_lo = 0, abs_hi = ~0ull;   // absolute index range (exact search with start / end index)
};

using LpKernel = void (*)(const uint8_t *, const FzScanArgs, const uint64_t *, uint64_t, FzGenRec *, unsigned long long *);

LpKernel lp_kernel(uint32_t kind, bool hbm_lists) {
    switch (kind) {
    case FZ_LP_GENERIC_HIT: return hbm_lists ? fz_lp_kernel<FZ_LP_GENERIC_HIT, true> : fz_lp_kernel<FZ_LP_GENERIC_HIT, false>;
    case FZ_LP_GENERIC_SEQ: return hbm_lists ? fz_lp_kernel<FZ_LP_GENERIC_SEQ, true> : fz_lp_kernel<FZ_LP_GENERIC_SEQ, false>;
    default: return hbm_lists ? fz_lp_kernel<FZ_LP_LEV_SEQ, true> : fz_lp_kernel<FZ_LP_LEV_SEQ, false>;
    }
}

using ScanKernel = void (*)(const uint8_t *, const FzScanArgs, uint64_t, uint64_t *, FzRec *, unsigned long long *);

template <bool FUSED, bool SEG, bool SA>
ScanKernel scan_kernel_f(int nwin, int dh) {
    if (nwin == 1) return fz_scan_kernel<1, 0, FUSED, SEG, SA>;
    switch (dh) {
        case 2: return fz_scan_kernel<2, 2, FUSED, SEG, SA>;
        case 3: return fz_scan_kernel<2, 3, FUSED, SEG, SA>;
        case 4: return fz_scan_kernel<2, 4, FUSED, SEG, SA>;
        default: return fz_scan_kernel<2, 5, FUSED, SEG, SA>;
    }
}

template <bool FUSED, bool SEG>
ScanKernel scan_kernel_s(int nwin, int dh, bool sa) {
    return sa ? scan_kernel_f<FUSED, SEG, true>(nwin, dh) : scan_kernel_f<FUSED, SEG, false>(nwin, dh);
}

// The segmented variants (file API) only exist where a segment changes the outcome: Levenshtein /
// generic clamps.  Exact and substitutions-only windows fit exactly one chunk (the host assigns it).
// sa: the launch's blocks are told apart by hash bits 2..6 (slot address = one v_and).
// The fused lane-per-cell form (Levenshtein budgets 5 .. 15 of in-memory searches, fz_flush_wf): GW = 16 or 32 lanes per candidate.
template <bool SA, int GW>
ScanKernel scan_kernel_wf(int nwin, int dh) {
    if (nwin == 1) return fz_scan_kernel<1, 0, true, false, SA, GW>;
    switch (dh) {
        case 2: return fz_scan_kernel<2, 2, true, false, SA, GW>;
        case 3: return fz_scan_kernel<2, 3, true, false, SA, GW>;
        case 4: return fz_scan_kernel<2, 4, true, false, SA, GW>;
        default: return fz_scan_kernel<2, 5, true, false, SA, GW>;
    }
}

// wf_gw: 0 = register band / Hamming count, 1 / 2 = bit-vector columns on one / two 64-bit words, 4 = on one 32-bit word,
// 3 = Hamming count under the bit-vector forms' queue discipline (dense candidates), 16 / 32 = lanes per candidate
ScanKernel scan_kernel(int nwin, int dh, bool fused, bool seg, bool sa, int wf_gw = 0) {
#ifdef FZ_LAB_ONLY      // lab builds (benchmarks/lab_build.sh): only the instances of the headline and exact-search workloads
    if (wf_gw == 1 && nwin == 2 && dh == 3 && fused && !seg) return sa ? fz_scan_kernel<2, 3, true, false, true, 1> : fz_scan_kernel<2, 3, true, false, false, 1>;
    if (wf_gw == 3 && nwin == 2 && dh == 3 && fused && !seg) return sa ? fz_scan_kernel<2, 3, true, false, true, 3> : fz_scan_kernel<2, 3, true, false, false, 3>;
    if (wf_gw == 2 && nwin == 2 && dh == 3 && fused && !seg) return sa ? fz_scan_kernel<2, 3, true, false, true, 2> : fz_scan_kernel<2, 3, true, false, false, 2>;
    if (wf_gw == 4 && nwin == 2 && dh == 3 && fused && !seg) return sa ? fz_scan_kernel<2, 3, true, false, true, 4> : fz_scan_kernel<2, 3, true, false, false, 4>;
    if (wf_gw) return nullptr;
    if (nwin == 2 && dh == 3 && fused && !seg) return sa ? fz_scan_kernel<2, 3, true, false, true> : fz_scan_kernel<2, 3, true, false, false>;
    if (nwin == 2 && dh == 5 && !fused && !seg) return sa ? fz_scan_kernel<2, 5, false, false, true> : fz_scan_kernel<2, 5, false, false, false>;
    return nullptr;
#else
    if (wf_gw == 1) return seg ? nullptr : (sa ? scan_kernel_wf<true, 1>(nwin, dh) : scan_kernel_wf<false, 1>(nwin, dh));
    if (wf_gw == 2) return seg ? nullptr : (sa ? scan_kernel_wf<true, 2>(nwin, dh) : scan_kernel_wf<false, 2>(nwin, dh));
    if (wf_gw == 4) return seg ? nullptr : (sa ? scan_kernel_wf<true, 4>(nwin, dh) : scan_kernel_wf<false, 4>(nwin, dh));
    if (wf_gw == 3) return seg ? nullptr : (sa ? scan_kernel_wf<true, 3>(nwin, dh) : scan_kernel_wf<false, 3>(nwin, dh));
    if (wf_gw == 16) return seg ? nullptr : (sa ? scan_kernel_wf<true, 16>(nwin, dh) : scan_kernel_wf<false, 16>(nwin, dh));
    if (wf_gw == 32) return seg ? nullptr : (sa ? scan_kernel_wf<true, 32>(nwin, dh) : scan_kernel_wf<false, 32>(nwin, dh));
    if (seg) return fused ? scan_kernel_s<true, true>(nwin, dh, sa) : scan_kernel_s<false, true>(nwin, dh, sa);
    return fused ? scan_kernel_s<true, false>(nwin, dh, sa) : scan_kernel_s<false, false>(nwin, dh, sa);
#endif
}

// Odd multipliers tried for the window hash (24-bit ones serve v_mad_u32_u24).  One search needs a
// multiplier under which its (at most 16 per launch) distinct block hashes fall into distinct slots of
// the 32-slot table: a random one works with probability 0.91 for 3 blocks, 0.39 for 8.
const uint32_t kHashMultipliers[] = {0x9E3779u, 0x85EBCBu, 0xC2B2AFu, 0x27D4EBu, 0x165667u, 0xD3A264u | 1u, 0xFD7047u, 0xB55A4Fu,
                                     0x7FEB35u, 0x846CA7u, 0x9E6C63u, 0x3243F7u, 0x517CC1u, 0xB7E151u, 0x6A09E7u, 0xBB67AFu,
                                     // (round 6: launches carry up to 16 blocks — more tries for the rarer perfect placements)
                                     0x3C6EF3u, 0xA54FF5u, 0x510E53u, 0x9B0569u, 0x1F83D9u, 0x5BE0CDu, 0xCA62C1u, 0x8F1BBDu,
                                     0x6ED9EBu, 0x5A8279u, 0xC3D2E1u, 0x10325Fu, 0x98BADDu, 0xEFCDABu, 0x674523u, 0x2B7E15u};

// Window geometry of the filter's hash for n-gram length L (see fz_hash_windows / fz_hash_short).
struct HashGeom {
    int nwin;            // 1: L <= 4 (masked dword * K);  2: dword + 24 bits at offset dh
    int dh;
    uint32_t mask1;
    explicit HashGeom(uint32_t L)
        : nwin(L <= 4 ? 1 : 2), dh(L <= 4 ? 0 : (int)std::min<uint32_t>(L, 8) - 3),
          mask1(L >= 4 ? 0xffffffffu : ((1u << (8 * L)) - 1u)) {}
    uint32_t hash(const uint8_t *ng, uint32_t L, uint32_t k) const {
        const uint32_t a1 = load_le32(ng, L) & mask1;
        return nwin == 2 ? fz_hash_windows(a1, load_le32(ng + dh, 3), k) : fz_hash_short(a1, k);
    }
};

// Blocks [g0, g0 + n) of one scan launch, the hash multiplier and the slot bits: the longest run of
// blocks (at most 16) whose distinct hashes land in distinct slots of the 32-slot table under some
// (multiplier, shift).  One block always fits; equal n-grams (equal hashes) share a slot.
// h = yh * K + x mixes x only through the addition: which five bits tell the blocks apart depends on
// where their bytes differ, so the slot bits are a per-launch choice as well.
uint32_t choose_launch_blocks(const uint8_t *p, const uint32_t *starts, uint32_t g0, uint32_t G, uint32_t L,
                              uint32_t &hash_k, uint32_t &lut_shift) {
    const HashGeom hg(L);
    // FZ_MAX_BLOCKS=n (test knob): at most n blocks per launch, to exercise the multi-launch path
    const uint32_t max_blocks = sw().max_blocks > 0 ? std::min<uint32_t>((uint32_t)sw().max_blocks, FZ_MAX_BLOCKS_PER_LAUNCH)
                                                             : FZ_MAX_BLOCKS_PER_LAUNCH;
    const bool no_sa = sw().no_slot_and;                               // test knob: never use the low-bits form
    const uint32_t want = std::min<uint32_t>(max_blocks, G - g0);
    uint32_t nblk = 0;
    // Pass 0: hash bits 2..6 as they are (lut_shift == 2: the kernel forms the slot address with one v_and; those
    // bits only see window bytes 0 and DH) — taken only if it fits every block this launch could carry.
    // Pass 1: any aligned group of five hash bits.
    for (int pass = no_sa ? 1 : 0; pass < 2; ++pass) {
        for (uint32_t cand : kHashMultipliers) {
            const uint32_t kk = hg.nwin == 2 ? cand : (cand * 0x9E3779B1u) | 1u;
            uint32_t hb[FZ_MAX_BLOCKS_PER_LAUNCH];
            uint32_t nb = 0;
            for (; nb < want; ++nb) hb[nb] = hg.hash(p + starts[g0 + nb], L, kk);
            for (int shift = pass == 0 ? 2 : 32 - FZ_LUT_BITS; shift >= 2; shift -= FZ_LUT_BITS) {
                uint32_t slot_hash[FZ_LUT_SLOTS];
                bool used[FZ_LUT_SLOTS] = {false};
                uint32_t fit = 0;
                for (; fit < nb; ++fit) {
                    const uint32_t slot = (hb[fit] >> shift) & (FZ_LUT_SLOTS - 1u);
                    if (used[slot] && slot_hash[slot] != hb[fit]) break;
                    used[slot] = true;
                    slot_hash[slot] = hb[fit];
                }
                if (fit > nblk && (pass == 1 || fit == want)) { nblk = fit; hash_k = kk; lut_shift = (uint32_t)shift; }
            }
            if (nblk == want) break;
        }
        if (nblk == want) break;
    }
    return nblk;
}

// a connected set of matches: its hull [h0, h1) and its best match (fz_consolidate, fz_generic_ngrams_consolidated)
struct Hull { int64_t h0, h1; fz_match best; };
int consolidate_hulls(std::vector<Hull> &hulls, fz_match **out, uint64_t *n_out);

int emit_generic(fz_ctx *ctx, fz_seq *seq, const std::vector<FzGenRec> &recs_vec, uint32_t L, uint32_t k, fz_match **out,
                 uint64_t *n, uint32_t **seg_out);

struct Search {
    uint32_t mode = 0, m = 0, k = 0;
    uint32_t max_subs = 0, max_ins = 0, max_dels = 0;      // generic search only
    const uint8_t *p = nullptr;
    BlockPlan plan;
    bool collective = false;       // the context joined a communicator: every rank gets the merged stream of all ranks
    bool any = false;              // has_near_match_*: only whether a record exists
    bool fold = false;             // generic search: the device folds every hit's matches into (hull, best match) pairs
    // has_near_match_* on a long single-shard sequence: the scan of THIS call covers the buffer bytes [part_lo, part_hi) only
    // (multiples of the tile size; hits are owned by the tile their index lies in, windows reach wherever they must)
    uint64_t part_lo = 0, part_hi = ~0ull;
};

// dynamic LDS per scan workgroup when verification is fused (a function of the switches: fz_debug_reload_switches reaches it)
#define kFusedLdsBudget ((uint32_t)(sw().fused_lds_kb > 0 ? sw().fused_lds_kb : 64) * 1024u)

// Fields of FzScanArgs that every kernel of a search shares.
void fill_common_args(FzScanArgs &fa, const Shard &sh, const Search &q) {
    memset(&fa, 0, sizeof fa);
    fa.geom = sh.geom;
    fa.mode = q.mode;
    fa.m = q.m;
    fa.k = q.k;
    fa.L = q.plan.L;
    fa.max_subs = q.max_subs; fa.max_ins = q.max_ins; fa.max_dels = q.max_dels;
    fa.abs_lo = q.plan.abs_lo;
    fa.abs_hi = q.plan.abs_hi;
    if (q.m <= FZ_MAX_M) memcpy(fa.pat, q.p, q.m);
}


// The pattern in HBM, for kernels that do not (or cannot) take it from the kernel-argument block.  The copy is
// ordered on the device's stream like the kernels that read it (searches of one context run one after the other
// on that stream, so two searches in flight can share the buffer).
int stage_pattern(DevState &d, FzScanArgs &fa, const uint8_t *p, uint32_t m, bool force = false) {
    fa.pat_g = 0;
    if (m <= FZ_MAX_M && !force) return FZ_OK;
    HIP_TRY(hipSetDevice(d.device));
    if (d.pat_cap < m) {
        HIP_TRY(hipStreamSynchronize(d.stream));
        if (d.d_pat) { HIP_TRY(hipFree(d.d_pat)); d.d_pat = nullptr; d.pat_cap = 0; }
        const uint64_t cap = std::max<uint64_t>(4096, (uint64_t)m * 2);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_pat), cap));
        d.pat_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(d.d_pat, p, m, hipMemcpyHostToDevice, d.stream));
    fa.pat_g = reinterpret_cast<uint64_t>(d.d_pat);
    return FZ_OK;
}

// Lanes per candidate of the lane-per-cell verification: the 2k + 1 band cells of a row must fit.
int wavefront_group(uint32_t k) { return 2 * k + 1 <= 16 ? 16 : 2 * k + 1 <= 32 ? 32 : 64; }

// Which stand-alone verification a search takes when it is not fused into the scan: lane-per-cell with LDS windows
// (Levenshtein budgets 5 .. 31), lane-per-candidate with an LDS score ring (everything else that fits LDS), or —
// patterns beyond the argument block, budgets beyond FZ_MAX_K, windows / rings beyond LDS — one wave per hit straight
// from HBM (fz_verify_big_kernel).
struct VerifyPlan {
    bool want_wf = false, big = false;
    int gw = 16;
    size_t wf_lds = 0, ring_lds = 0;
    unsigned waves = 4;
};

VerifyPlan plan_verify(const Search &q) {
    const bool force_big = sw().force_big_verify;                               // test knob: every stand-alone verification by fz_verify_big_kernel
    const bool no_wf = sw().no_wavefront;
    VerifyPlan v;
    const uint32_t mpad = (q.m + 15u) & ~15u;
    const uint32_t win_dwords = (q.m + 2 * q.k + 6) / 4 + 1;
    const uint32_t band_w = q.mode == FZ_MODE_LEV ? 2 * q.k + 2 : 0;
    v.gw = wavefront_group(q.k);
    const uint32_t wf_per_wave = (64u / v.gw) * (win_dwords * 4u + 16u);       // one window per hit (at most 64 / gw hits per wave)
    v.wf_lds = 16 + mpad + 16 + 16 * (size_t)wf_per_wave;                       // 16 waves per workgroup
    while (v.waves > 1 && mpad + v.waves * (size_t)fz_wave_lds_bytes(win_dwords, band_w, 64, false) > 64 * 1024) v.waves >>= 1;
    v.ring_lds = mpad + (size_t)v.waves * fz_wave_lds_bytes(win_dwords, band_w, 64, false);
    v.want_wf = q.mode == FZ_MODE_LEV && q.k >= 5 && q.k <= 31 && !no_wf;
    v.big = force_big || q.m > FZ_MAX_M || q.k > FZ_MAX_K || (v.want_wf ? v.wf_lds > 64 * 1024 : v.ring_lds > 160 * 1024);
    return v;
}

// The regions of a scan grid (FzScanArgs.reg_*; nreg = 0: every workgroup strides over all tiles).  `steps` groups of
// the last `t_per_cu x n_cus` workgroups take shares shrinking to `fmin` of a full one.
void plan_scan_regions(FzScanArgs &fa, uint64_t ntiles, uint64_t grid, uint32_t n_cus, int steps, double fmin, int t_per_cu) {
    fa.nreg = 0;
    steps = std::min(steps, FZ_MAX_REGIONS - 1);
    fmin = std::min(1.0, std::max(0.02, fmin));
    const uint64_t Gw = grid, T = (uint64_t)n_cus * (uint64_t)std::max(1, t_per_cu);
    if (steps <= 0 || (uint64_t)steps > T || Gw < 2 * T || ntiles < 4 * Gw) return;
    const uint64_t per = T / steps;                            // workgroups per taper group (the last group takes the rest)
    double weight = (double)(Gw - T);
    std::vector<double> f(steps);
    std::vector<uint64_t> nw(steps);
    for (int j = 0; j < steps; ++j) {
        f[j] = 1.0 - (1.0 - fmin) * (j + 1) / steps;
        nw[j] = j + 1 < steps ? per : T - per * (steps - 1);
        weight += f[j] * (double)nw[j];
    }
    const double share = (double)ntiles / weight;              // tiles of a full workgroup
    uint64_t at = 0, wg = 0;
    auto add = [&](uint64_t nwg, uint64_t tiles) {
        fa.reg_wg0[fa.nreg] = (uint32_t)wg; fa.reg_nwg[fa.nreg] = (uint32_t)nwg;
        fa.reg_tile0[fa.nreg] = at; fa.reg_end[fa.nreg] = at + tiles;
        ++fa.nreg; wg += nwg; at += tiles;
    };
    add(Gw - T, std::min<uint64_t>(ntiles, (uint64_t)(share * (double)(Gw - T) + 0.5)));
    for (int j = 0; j < steps; ++j) {
        const uint64_t left = ntiles - at;
        add(nw[j], j + 1 < steps ? std::min<uint64_t>(left, (uint64_t)(share * f[j] * (double)nw[j] + 0.5)) : left);
    }
    // the queue codes carry a bounded per-workgroup tile iteration
    for (uint32_t r = 0; r < fa.nreg; ++r)
        if ((fa.reg_end[r] - fa.reg_tile0[r] + fa.reg_nwg[r] - 1) / fa.reg_nwg[r] >= FZ_TITER_MAX) { fa.nreg = 0; break; }
}

// ... with the process-wide settings (FZ_TAPER_STEPS, default 4; FZ_TAPER_MIN, default 0.25; FZ_TAPER_WG_PER_CU, default 7 =
// the workgroups of this kernel that are resident per CU).
void plan_scan_regions(FzScanArgs &fa, uint64_t ntiles, uint64_t grid, uint32_t n_cus) {
    const int steps = sw().taper_steps;
    const double fmin = sw().taper_min;
    const int t_per_cu = sw().taper_wg_per_cu;
    plan_scan_regions(fa, ntiles, grid, n_cus, steps, fmin, t_per_cu);
}

// Enqueue scan (+ separate verify when it cannot be fused) for one shard on its device stream.
// No host synchronisation.
int enqueue_shard(fz_ctx *ctx, const Shard &sh, const Search &q, bool with_verify, bool copy_back = true) {
    DevState &d = lane_dev(ctx, sh.dev);
    HIP_TRY(hipSetDevice(d.device));
    unsigned long long *counters = reinterpret_cast<unsigned long long *>(d.d_out);
    FzRec *recs = reinterpret_cast<FzRec *>(d.d_out + kHeaderBytes);
    const bool no_direct = env_no_direct();
    // direct mode needs a kernel to publish the counters: an empty buffer launches none
    const bool snapshot = copy_back && with_verify && q.collective;      // collective search: records stay on the device
    const bool direct = copy_back && d.direct && !no_direct && !snapshot && sh.geom.buf_len > 0 && !q.plan.s.empty();
    if (direct && with_verify) recs = reinterpret_cast<FzRec *>(d.h_stage_dev + kHeaderBytes);
    d.last_direct = direct;
    d.timed = ctx->timing;

    const uint32_t L = q.plan.L;
    const uint32_t G = (uint32_t)q.plan.s.size();
    const uint64_t ntiles_all = (sh.geom.buf_len + FZ_TILE_BYTES - 1) / FZ_TILE_BYTES;
    const uint64_t tile_lo = std::min<uint64_t>(ntiles_all, q.part_lo / FZ_TILE_BYTES);
    const uint64_t tile_hi = q.part_hi == ~0ull ? ntiles_all : std::min<uint64_t>(ntiles_all, (q.part_hi + FZ_TILE_BYTES - 1) / FZ_TILE_BYTES);
    const bool partial = tile_lo != 0 || tile_hi != ntiles_all;
    const uint64_t ntiles = tile_hi > tile_lo ? tile_hi - tile_lo : 0;          // tiles this call scans
    // Grid: every workgroup strides over ~16 tiles (256 KiB).  Measured on MI355X at 1 GiB: 6 / 8 /
    // 12 / 16 / 20 / 32 / 64 workgroups per CU -> 0.302 / 0.302 / 0.276 / 0.267 / 0.265 / 0.280 /
    // 0.333 ms: several rounds of short workgroups overlap one workgroup's end-of-life verification
    // (latency-bound) with the others' streaming; too many pay the per-workgroup fixed cost.  At least
    // 6 per CU (the co-resident count at this kernel's SGPR use) so small inputs still fill the chip.
    // (Round 2 measured the alternative — a persistent grid whose waves draw 4 KiB chunks from ticket
    // counters so that all finish together: same time without candidates, 0.04 ms slower on DNA,
    // because every wave then runs its verification at the same moment, at the end.)
    // (re-measured with the software-pipelined loop of round 2, 1 GiB: 8 / 10 / 12 / 14 / 16 / 20 / 24 / 32 tiles ->
    //  DNA k = 2: 0.232 / 0.223 / 0.222 / 0.222 / 0.224 / 0.240 / 0.244 / 0.260 ms; exact search: 0.191 / 0.182 /
    //  0.187 / 0.193 / 0.199 / 0.201 / 0.196 / 0.205 ms)
    // Round 3 (two-level finish tickets in place, `profiles/r03_lab_ab.txt`): what matters below ~10 rounds of resident
    // workgroups is that the grid is a WHOLE number of rounds of 6 workgroups per CU (round 4's device stamps show SEVEN
    // resident per CU — amdgpu_waves_per_eu(7, 7) — so these are not rounds of residents; the multiples of 6 per CU
    // stay because they measured best, with and without the tapered last round below; a last round that is
    // 56 % full costs 1 GiB 8 us) of ~9.5 tiles per workgroup: 1 GiB, 4 rounds (10.7 tiles) 0.2046 ms, 5 rounds 0.2048,
    // 3 rounds 0.2062, 6 rounds 0.2074, 8 rounds 0.2104 against 0.2121-0.2146 for 12 tiles (3.56 rounds); 2 GiB, 9 rounds
    // 0.3887 against 0.3989; 512 MiB, 2 rounds 0.1165 against 0.1182.  Long inputs keep 12 tiles per workgroup (4 GiB:
    // 0.7743 against 0.7791-0.7855 for 14-18 whole rounds and 0.802 for 8 tiles: the per-workgroup cost wins there).
    const int tiles_per_wg_env = sw().tiles_per_wg > 0 ? sw().tiles_per_wg : 0;
    const int tiles_per_wg = tiles_per_wg_env ? tiles_per_wg_env : 12;
    const uint64_t resident = (uint64_t)d.n_cus * 6;
    uint64_t max_grid = std::max<uint64_t>(resident, ntiles / tiles_per_wg);
    if (!tiles_per_wg_env && ntiles < resident * 120) {
        const uint64_t rounds = std::max<uint64_t>(1, (2 * ntiles + resident * 19 / 2) / (resident * 19));   // round(ntiles / (9.5 resident))
        max_grid = resident * rounds;
    }
    // (has_near_match_*, measured in round 5 on 4 GiB with a match in the first MiB: 0.21 ms against 0.84 ms for the full
    //  scan.  What is left is NOT the running workgroups finishing their tiles but the ~21 800 workgroups that start after
    //  the first record, skip their tiles and still take their finish tickets: ~100 agent-scope atomics per microsecond on
    //  the sixteen shard words, which share one cache line.  Tried and dropped: a check of the record counter per tile in
    //  the scan's loop (cannot shorten what the tickets bound); workgroups of 6 tiles (0.31 ms: twice the tickets); the shard
    //  words spread over four lines of the header (every scan slower — 1 GiB exact search 0.188-0.193 -> 0.197 ms, same
    //  box: those lines also hold the statistics words).)
    {   // lab knobs: FZ_ROUNDS=r -> a grid of r x (FZ_WG_PER_CU workgroups per CU): whole rounds of resident workgroups
        const int rounds = sw().rounds;
        const int per_cu = sw().wg_per_cu > 0 ? sw().wg_per_cu : 6;
        if (rounds > 0) max_grid = (uint64_t)d.n_cus * per_cu * rounds;
    }
    // the queue codes carry a bounded per-workgroup tile iteration
    const uint64_t min_grid = (ntiles + FZ_TITER_MAX - 1) / FZ_TITER_MAX;
    dim3 grid((unsigned)std::max<uint64_t>(std::max<uint64_t>(1, min_grid), std::min<uint64_t>(ntiles, max_grid)));

    FzScanArgs fa;
    fill_common_args(fa, sh, q);
    // Tapered last round: workgroups start in blockIdx order, so the last resident round of a launch starts while the
    // machine is still full and — with equal shares — ends one workgroup life (~60 us) after the grid ran dry, the chip
    // draining all the while (device stamps of every workgroup, benchmarks/lab_scan_phases.py: residency falls linearly from
    // 1 792 to 0 over the last 60 us of a 1 GiB launch).  The last `resident` workgroups therefore take shrinking shares
    // (kTaperSteps groups, down to kTaperMin of a full share) of their own tile range at the end of the buffer, the others
    // correspondingly more.
    plan_scan_regions(fa, ntiles, grid.x, (uint32_t)d.n_cus);
    if (partial) {                                               // one region: the call's tiles, no taper
        fa.nreg = 1;
        fa.reg_wg0[0] = 0; fa.reg_nwg[0] = grid.x; fa.reg_tile0[0] = tile_lo; fa.reg_end[0] = tile_hi;
    }
    if (q.mode == FZ_MODE_GENERIC && !with_verify) fa.gen_dedup = d.gen_dedup_arg;      // the scan fills the window table (run_generic)
    const bool force_big = sw().force_big_verify;
    const VerifyPlan vp = plan_verify(q);
    {
        int rc = stage_pattern(d, fa, q.p, q.m, with_verify && vp.big);
        if (rc) return rc;
    }
    const HashGeom hgeom(L);
    const int nwin = hgeom.nwin, dh = hgeom.dh;
    fa.d2 = nwin == 2 ? std::min<uint32_t>(L, 8) - 4 : 0;
    fa.mask1 = hgeom.mask1;
    fa.mask2 = 0xffffffffu;
    fa.band_w = q.mode == FZ_MODE_LEV ? 2 * q.k + 2 : 0;
    fa.win_dwords = (q.m + 2 * q.k + 6) / 4 + 1;
    fa.hit_cap = d.hit_cap;
    fa.rec_cap = direct ? kHostRecs : d.rec_cap;
    const uint32_t mpad = (q.m + 15u) & ~15u;
    // Lanes that verify at once: all 64 while the staged windows stay small; fewer for long patterns
    // so that the scan keeps ~8 workgroups per CU resident (measured at m = 64, k = 5 on 1 GiB of text:
    // 64 lanes -> 31.6 KB LDS, 5 workgroups/CU, scan 0.540 ms; candidates are rare there anyway).
    const uint32_t target = (uint32_t)(sw().fused_target_kb > 0 ? sw().fused_target_kb : 18) * 1024u;
    fa.vlanes = 64;
    uint32_t fused_lds;
    if (sh.geom.seg_stride == 0) {
        // in-memory search: the windows of queued candidates are prefetched by LDS-DMA, 16-byte pieces, one slot
        // per queue entry (fz_prefetch_windows); the queue shrinks for long patterns (a tile that outgrows it is
        // re-scanned by enumeration, 64 entries at a time: 64 is the floor)
        fa.win_pieces = (q.m + 2 * q.k + 3 + 15) / 16;           // window <= m + 2k bytes + 3 of dword alignment
        fa.qcap = 128;
        while (fa.qcap > 64 && mpad + FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(fa.qcap, fa.win_pieces) > target + 4096)
            fa.qcap >>= 1;
        fused_lds = mpad + FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(fa.qcap, fa.win_pieces);
        if (fa.win_pieces * 16u + 16u > FZ_PAD_BACK) fused_lds = ~0u;   // the last pieces may lie past the sequence: inside the padding only
    } else {
        while (fa.vlanes > 16 && mpad + FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_bytes(fa.win_dwords, fa.band_w, fa.vlanes, true) > target)
            fa.vlanes >>= 1;
        fused_lds = mpad + FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_bytes(fa.win_dwords, fa.band_w, fa.vlanes, true);
    }
    // k > 4 (register band too wide for the scan kernel's VGPR budget) -> lane-per-cell: inside the scan as well while 16
    // or 32 lanes hold the band (budgets 5 .. 15: the candidates of a wave's queue, four or two at a time), in a kernel of
    // its own beyond that
    fa.fused = (with_verify && !force_big && q.m <= FZ_MAX_M && q.k <= FZ_MAX_K && fused_lds <= kFusedLdsBudget &&
                (q.mode != FZ_MODE_LEV || q.k <= 4)) ? 1u : 0u;
    // Levenshtein budgets from FZ_BITS_MIN_K on, patterns up to 128 characters, in-memory: bit-vector columns, one candidate
    // per lane, inside the scan (fz_verify_lev_bits) — a column costs the same whatever the budget, so neither the band's
    // width nor the candidates' density decides the form.  The queue takes what LDS allows (full passes: fz_bits_flush).
    int bits_nw = 0;
    // Candidates a wave is expected to find in one tile (4 KiB of offsets per wave) if the sequence is uniform over the
    // symbols the PATTERN uses — G blocks of L characters over sigma symbols: 4096 G / sigma^L (DNA, m = 54, k = 8: 9;
    // m = 20, k = 4: 80; text patterns: next to nothing).  A function of the search's arguments, not of earlier calls; an
    // estimate that is off costs time, never rows.
    double per_tile = 0;
    {
        bool seen[256] = {false};
        uint32_t sigma = 0;
        for (uint32_t i = 0; i < q.m; ++i)
            if (!seen[q.p[i]]) { seen[q.p[i]] = true; ++sigma; }
        per_tile = 4096.0 * std::min<uint32_t>(G, FZ_MAX_BLOCKS_PER_LAUNCH);
        for (uint32_t i = 0; i < L && per_tile > 0.01; ++i) per_tile /= (double)std::max(2u, sigma);
    }
    // Which budgets: from FZ_BITS_MIN_K (3) on always — at k = 3 on DNA it is 20 % ahead of the register band, at k >= 5 it
    // replaces the lane-per-cell forms; below that only where the expected density is beyond what the register-band form's
    // queue discipline takes (more than ~16 candidates per tile and wave: short DNA patterns), the headline workload
    // (k = 2, 3 per tile) keeps the band.
    const bool bits_budget = q.k >= (uint32_t)sw().bits_min_k || per_tile > 16.0;
    if (q.mode == FZ_MODE_LEV && with_verify && !force_big && !sw().no_bits && sh.geom.seg_stride == 0 && bits_budget &&
        q.m <= FZ_BITS_MAX_M(2) && q.k <= FZ_MAX_K && fa.win_pieces * 16u + 16u <= FZ_PAD_BACK) {
        bits_nw = q.m <= FZ_BITS_MAX_M(4) ? 4 : q.m <= FZ_BITS_MAX_M(1) ? 1 : 2;      // 32-, 64-, 128-bit columns
        // (26 KB: six workgroups per CU.  Measured on 1 GiB of DNA, m = 54, k = 8, 2.4e6 candidates: 64 / 96 / 128 / 160 entries
        //  per wave = 26 / 37 / 47 / 58 KB -> 0.463 / 0.517 / 0.656 / 0.830 ms: fuller passes do not pay for the lost waves)
        const uint32_t budget = (uint32_t)(sw().bits_lds_kb > 0 ? sw().bits_lds_kb : 26) * 1024u;
        const uint32_t fixed = mpad + FZ_TABLE_BYTES + FZ_PEQ_BYTES(bits_nw);
        // Queue entries per wave: one full pass (64) + twice the expected candidates per tile, within the LDS budget.
        uint32_t qc = sw().bits_qcap > 0 ? (uint32_t)sw().bits_qcap
                                         : (uint32_t)std::min(512.0, 64.0 + 32.0 * std::ceil(2.0 * per_tile / 32.0));
        while (qc > 64u && fixed + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(qc, fa.win_pieces) > budget) qc -= 32u;
        fa.qcap = qc;
        fused_lds = fixed + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(fa.qcap, fa.win_pieces);
        if (fused_lds > kFusedLdsBudget) bits_nw = 0;
        else fa.fused = 1u;
    }
    // Substitutions-only searches whose pattern lets expect dense candidates: the same Hamming count under the bit-vector forms' queue discipline — full 64-candidate passes, tiles denser than the queue
    // taken in block-range passes — instead of round 1's (half-full queues; a tile that overflows is enumerated: DNA, m = 20,
    // 4 substitutions: 14 ms per GiB; m = 12, 3: 45 ms).
    bool adapt_plain = false;
    if (!bits_nw && fa.fused && sh.geom.seg_stride == 0 && with_verify && !sw().no_bits && per_tile > 16.0 && q.mode == FZ_MODE_SUBS) {
        const uint32_t fixed = mpad + FZ_TABLE_BYTES;
        uint32_t qc = sw().bits_qcap > 0 ? (uint32_t)sw().bits_qcap : (uint32_t)std::min(512.0, 64.0 + 32.0 * std::ceil(2.0 * per_tile / 32.0));
        while (qc > 64u && fixed + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(qc, fa.win_pieces) > target + 4096) qc -= 32u;
        if (fixed + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(qc, fa.win_pieces) <= kFusedLdsBudget) {
            adapt_plain = true;
            fa.qcap = qc;
            fused_lds = fixed + FZ_WAVES_PER_BLOCK * fz_wave_lds_pref_bytes(qc, fa.win_pieces);
        }
    }
    const bool no_wf_fuse = sw().no_wf_fuse;                                    // test / measurement knob: the stand-alone kernel
    const uint32_t wf_fused_lds = mpad + FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_bytes(fz_wf_fused_dwords(fa.win_dwords, (uint32_t)vp.gw), 0, 1, true);
    // (in-memory searches only: the segmented instances of this form spill registers — the file API keeps the kernel of its own)
    // 32 lanes per candidate (budgets 8 .. 15): the fused form is 7-14 % behind scan + stand-alone kernel where candidates
    // are rare (1 GiB of text, m = 64, k = 8 / 12: 0.546 / 0.654 against 0.509 / 0.597 ms) and 1.25 .. 7.7 x ahead where they
    // are not (DNA, m = 100, k = 10, 4.5e4 candidates: 0.551 against 0.687 ms; m = 54, k = 8, 2.4e6: 2.70 against 20.7 ms —
    // the hit list outgrows its buffer and the search runs twice).  FZ_WF32=0 / 1 pins the choice.
    const int wf32_env = sw().wf32;
    const bool wf_candidate = !bits_nw && !fa.fused && sh.geom.seg_stride == 0 && with_verify && !force_big && !no_wf_fuse && vp.want_wf && !vp.big && vp.gw <= 32 &&
                              q.m <= FZ_MAX_M && wf_fused_lds <= target + 4096;
    d.wf32_candidate = wf_candidate && vp.gw == 32;
    // (round 6: patterns up to 128 characters never get here — bit-vector columns; for the longer ones the choice follows the
    //  expected density per_tile above, a function of the pattern — the measured crossover, 1.5e4 .. 4.5e4 candidates per GiB, is
    //  ~0.1 per tile and wave — not, as in rounds 4 and 5, what the context's previous search saw)
    const bool wf_fused = wf_candidate && (vp.gw == 16 || (wf32_env >= 0 ? wf32_env != 0 : per_tile > 0.1));
    if (wf_fused) { fa.fused = 1u; fused_lds = wf_fused_lds; fa.vlanes = 64; }
    // (the hit-emitting form keeps no pattern in LDS: fz_confirm reads it from the argument block / HBM)
    const uint32_t scan_lds = fa.fused ? fused_lds : FZ_TABLE_BYTES + FZ_WAVES_PER_BLOCK * fz_wave_lds_bytes(0, 0, 64, true);
    // the stream and the counter block of this search (see DevState::stream_alt)
    const bool alt = ctx->streams == 2 && d.slot_id == 1 && fa.fused && direct && copy_back && !fa.pat_g && sh.geom.seg_stride == 0;
    const hipStream_t st = alt ? d.stream_alt : d.stream;
    if (alt) counters = reinterpret_cast<unsigned long long *>(d.d_hdr_alt);
    else {
        if (!d.header_zeroed) HIP_TRY(hipMemsetAsync(d.d_out, 0, kHeaderBytes, d.stream));
        d.header_zeroed = false;
    }

    // When the scan's last launch is also the search's last kernel (fused verification, results written straight to
    // the host), the start / completion events ride on the kernels' own dispatch packets (hipExtLaunchKernelGGL)
    // instead of two extra packets around them: fewer packets on the critical path of a synchronous call, and
    // filter_ms becomes the kernels' own span.
    const bool no_ext = sw().no_ext_launch;
    const bool ext_events = !no_ext && copy_back && direct && !(with_verify && !fa.fused) && ntiles > 0 && G > 0;
    // ... and whenever the scan launches a kernel at all, its start / end events (ev[0], ev[1]: fz_stats' filter_ms) ride
    // on the first / last launch as well instead of two packets of their own in front of and behind the scan
    const bool attach = !no_ext && ntiles > 0 && G > 0;
    if (ctx->timing && !attach) HIP_TRY(hipEventRecord(d.ev[0], st));
    uint32_t launches = 0;
    for (uint32_t g0 = 0; g0 < G && ntiles > 0;) {
        // Blocks [g0, g0 + nblk) of this launch and the hash multiplier: the longest run of blocks (at
        // most 8) whose distinct hashes land in distinct table slots under some multiplier.  One block
        // always fits; equal n-grams (equal hashes) share a slot.
        uint32_t hash_k = 0, lut_shift = 0;
        const uint32_t nblk = choose_launch_blocks(q.p, q.plan.s.data(), g0, G, L, hash_k, lut_shift);
        fa.hash_k = hash_k;
        fa.lut_shift = lut_shift;
        fa.nblk = nblk;
        fa.g0 = g0;
        for (uint32_t b = 0; b < nblk; ++b) {
            const uint8_t *ng = q.p + q.plan.s[g0 + b];
            fa.A[b] = load_le32(ng, L) & fa.mask1;
            fa.B[b] = nwin == 2 ? load_le32(ng + fa.d2, 4) : 0;
            fa.H[b] = nwin == 2 ? fz_hash_windows(fa.A[b], load_le32(ng + dh, 3), fa.hash_k) : fz_hash_short(fa.A[b], fa.hash_k);
            fa.s[b] = q.plan.s[g0 + b];
            fz_block_range(q.mode, q.m, q.k, L, fa.s[b], fa.lo_rel[b], fa.hi_sub[b]);
        }
        // the final kernel of the search publishes the counters to the host (direct mode)
        const bool verify_follows = with_verify && !fa.fused;
        fa.host_hdr = (direct && !verify_follows && g0 + nblk >= G) ? reinterpret_cast<uint64_t>(d.h_stage_dev) : 0;
        // equal n-grams (equal hashes) share a table slot: the rare path then compares with every block
        fa.flags = q.any ? FZ_FLAG_ANY : 0u;
        for (uint32_t b = 1; b < nblk; ++b)
            for (uint32_t c = 0; c < b; ++c)
                if (fa.H[b] == fa.H[c]) fa.flags |= FZ_FLAG_DUP_HASHES;
        ScanKernel kern = scan_kernel(nwin, dh, fa.fused != 0, sh.geom.seg_stride != 0, fa.lut_shift == 2, bits_nw ? bits_nw : adapt_plain ? 3 : wf_fused ? vp.gw : 0);
        if (!kern) return fail(FZ_EUNSUPPORTED, "this (lab) build carries no scan kernel for nwin=%d dh=%d", nwin, dh);
        const uint32_t extra_lds = (uint32_t)sw().extra_lds_kb * 1024u;
        hipEvent_t ev_start = (attach && ctx->timing && g0 == 0) ? d.ev[0] : nullptr;
        hipEvent_t ev_stop = g0 + nblk >= G ? (ext_events ? d.ev[3] : (attach && ctx->timing) ? d.ev[1] : nullptr) : nullptr;
        if (ev_start || ev_stop)
            hipExtLaunchKernelGGL(kern, grid, dim3(FZ_FILTER_THREADS), scan_lds + extra_lds, st, ev_start, ev_stop, 0u, sh.d_buf, fa,
                                  ntiles, d.d_hits, recs, counters);
        else
            hipLaunchKernelGGL(kern, grid, dim3(FZ_FILTER_THREADS), scan_lds + extra_lds, st, sh.d_buf, fa, ntiles, d.d_hits, recs,
                               counters);
        HIP_TRY(hipGetLastError());
        ++launches;
        g0 += nblk;
    }
    // ev[1] = end of the scan.  When the results need no copy and no verify kernel follows, ev[3] is
    // recorded at the same point of the stream: one event packet less on the critical path.
    d.scan_end_event = (copy_back && direct && !(with_verify && !fa.fused)) ? 3 : 1;
    if (d.scan_end_event == 1 && ctx->timing && !attach) HIP_TRY(hipEventRecord(d.ev[1], st));
    d.verify_launched = false;
    d.verify_end_event = 2;
    // the verification kernel is the search's last one when its records go straight to the host: then the completion
    // event rides on its launch (no ev[2] / ev[3] packets behind it; the verify span is ev[1] .. ev[3])
    hipEvent_t v_stop = (!no_ext && copy_back && direct && !snapshot) ? d.ev[3] : nullptr;
#define FZ_LAUNCH_VERIFY(kernel, grid_, block_, lds_)                                                                      \
    do {                                                                                                                   \
        if (v_stop) hipExtLaunchKernelGGL(kernel, grid_, block_, lds_, d.stream, nullptr, v_stop, 0u, sh.d_buf, fa, d.d_hits, recs, counters); \
        else hipLaunchKernelGGL(kernel, grid_, block_, lds_, d.stream, sh.d_buf, fa, d.d_hits, recs, counters);            \
    } while (0)
    if (with_verify && !fa.fused && ntiles > 0 && G > 0) {
        d.verify_launched = true;
        if (v_stop) d.verify_end_event = 3;
        fa.nblk = 0;
        fa.g0 = 0;
        fa.host_hdr = direct ? reinterpret_cast<uint64_t>(d.h_stage_dev) : 0;
        const int gw = vp.gw;
        const size_t wf_lds = vp.wf_lds, ring_lds = vp.ring_lds;
        const unsigned waves = vp.waves;
        const bool want_wf = vp.want_wf, big = vp.big;
        if (big) {
            if (!fa.pat_g) return fail(FZ_EDEVICE, "internal: the pattern was not staged for the big verification");
            const uint32_t cells = q.mode == FZ_MODE_LEV ? 2 * q.k + 1 : 1;
            const dim3 bgrid(d.n_cus * 32), bblock(64);
            if (cells <= 64) FZ_LAUNCH_VERIFY(fz_verify_big_kernel<1>, bgrid, bblock, 0);
            else if (cells <= 128) FZ_LAUNCH_VERIFY(fz_verify_big_kernel<2>, bgrid, bblock, 0);
            else if (cells <= 256) FZ_LAUNCH_VERIFY(fz_verify_big_kernel<4>, bgrid, bblock, 0);
            else if (cells <= 512) FZ_LAUNCH_VERIFY(fz_verify_big_kernel<8>, bgrid, bblock, 0);
            else if (cells <= 1024) FZ_LAUNCH_VERIFY(fz_verify_big_kernel<16>, bgrid, bblock, 0);
            else FZ_LAUNCH_VERIFY(fz_verify_big_kernel<32>, bgrid, bblock, 0);
        } else if (want_wf) {
            // lane-per-cell: 64 / gw candidates per wave, one contiguous byte window per candidate
            fa.gw = (uint32_t)gw;
            // 16 waves per workgroup: few workgroups = few finish tickets (every ticket is an atomic on one word)
            const dim3 vgrid(d.n_cus * 2), vblock(1024);
            if (gw == 16) FZ_LAUNCH_VERIFY(fz_verify_wf_kernel<16>, vgrid, vblock, wf_lds);
            else if (gw == 32) FZ_LAUNCH_VERIFY(fz_verify_wf_kernel<32>, vgrid, vblock, wf_lds);
            else FZ_LAUNCH_VERIFY(fz_verify_wf_kernel<64>, vgrid, vblock, wf_lds);
        } else {
            // LDS: pattern + per-wave window and score ring; the block was shrunk until it fits.
            fa.vlanes = 64;
            if (ring_lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fz_verify_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring_lds));
            FZ_LAUNCH_VERIFY(fz_verify_kernel, dim3(d.n_cus * 4), dim3(64 * waves), ring_lds);
        }
        HIP_TRY(hipGetLastError());
    }
#undef FZ_LAUNCH_VERIFY
    const bool completion_attached = ext_events || (d.verify_launched && v_stop);
    if (copy_back) {
        if (d.verify_launched && ctx->timing && d.verify_end_event == 2) HIP_TRY(hipEventRecord(d.ev[2], d.stream));
        if (snapshot) {
            // counters to the host (overflow checks, statistics); counters + records to this slot's snapshot
            HIP_TRY(hipMemcpyAsync(d.h_stage, d.d_out, kHeaderBytes, hipMemcpyDeviceToHost, d.stream));
            int rc = ensure_send(d, d.rec_cap);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(d.d_send[d.slot_id], d.d_out, kHeaderBytes + d.rec_cap * sizeof(FzRec), hipMemcpyDeviceToDevice, d.stream));
            HIP_TRY(hipEventRecord(d.ev_snap[d.slot_id], d.stream));
            d.snap_taken[d.slot_id] = true;
        } else if (!direct) {
            d.first_copy = std::min<uint64_t>(std::min<uint64_t>(d.first_copy, kFirstCopyRecs), d.rec_cap);
            HIP_TRY(hipMemcpyAsync(d.h_stage, d.d_out, kHeaderBytes + d.first_copy * sizeof(FzRec), hipMemcpyDeviceToHost,
                                   d.stream));
        }
        if (!completion_attached) HIP_TRY(hipEventRecord(d.ev[3], st));
        // the counters are zeroed for the NEXT search now, off the critical path of that call: by the
        // publishing workgroup itself in direct mode, by a memset behind the copy otherwise
        if (!direct) HIP_TRY(hipMemsetAsync(d.d_out, 0, kHeaderBytes, d.stream));
        if (!alt) d.header_zeroed = true;
    }
    d.launches_used = launches;                  // (summed by search_enqueue: this may run on the device's worker thread)
    d.fused_used = fa.fused != 0;
    d.form_used = !with_verify ? FZ_FORM_NONE : bits_nw == 4 ? FZ_FORM_FUSED_BITS32 : bits_nw == 1 ? FZ_FORM_FUSED_BITS1 : bits_nw == 2 ? FZ_FORM_FUSED_BITS2
                  : wf_fused ? FZ_FORM_FUSED_CELLS : fa.fused ? FZ_FORM_FUSED_BAND : FZ_FORM_KERNEL;
    d.hit_cap_used = d.hit_cap;
    d.rec_cap_used = d.rec_cap;
    return FZ_OK;
}

// Wait for a shard, handle overflow (so.rerun = capacities grown, the caller must re-run), collect into `so`.
// May run on the device's worker thread: touches the device state, `so` and — only for single-shard searches
// (view_ok), which never run on a worker — the context's record view.
int collect_shard(fz_ctx *ctx, const Shard &sh, bool with_verify, bool view_ok, ShardOut &so, bool collective = false) {
    DevState &d = ctx->devs[sh.dev];
    HIP_TRY(hipSetDevice(d.device));
    Trace tr;
    so.recs.clear();
    so.hits.clear();
    so.rerun = false;
    so.has_tref = false;
    so.nh = so.nr = so.bytes = 0;
    // wait for the copy only: the memset that pre-zeroes the header for the next search runs behind it
    HIP_TRY(hipEventSynchronize(d.ev[3]));
    tr.mark("  sync");
    const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(d.h_stage);
    uint64_t nh = cnt[0];
    const uint64_t nr = cnt[1];
    const bool fused = with_verify && d.fused_used;
    if (fused) { nh = 0; for (int i = 0; i < 64; ++i) nh += cnt[8 + i]; }
    bool rerun = false;
    if (nh > d.hit_cap_used && !fused) {
        HIP_TRY(hipStreamSynchronize(d.stream));          // a second search in flight still uses the old buffers
        HIP_TRY(hipStreamSynchronize(d.stream_alt));
        int rc = ensure_hits(d, nh + nh / 8 + 1024);
        if (rc) return rc;
        rerun = true;
    }
    if (d.last_direct && with_verify && nr > kHostRecs) d.direct = false;   // too many for the staging buffer
    if (nr > (d.last_direct ? kHostRecs : d.rec_cap_used)) {
        HIP_TRY(hipStreamSynchronize(d.stream));
        HIP_TRY(hipStreamSynchronize(d.stream_alt));
        int rc = ensure_recs(d, nr + nr / 8 + 1024);
        if (rc) return rc;
        rerun = true;
    }
    if (!d.last_direct && nr * 4 < kHostRecs) d.direct = true;
    so.rerun = rerun;
    if (rerun) return FZ_OK;
    d.last_filter_ms = 0;
    if (d.timed) {
        so.has_tref = true;
        so.td = &d;
        so.tev[0] = d.ev[0]; so.tev[1] = d.ev[d.scan_end_event]; so.tev[2] = d.verify_launched ? d.ev[1] : nullptr;
        so.tev[3] = d.ev[d.verify_end_event]; so.tev[4] = d.ev[0]; so.tev[5] = d.ev[3];
    }
    so.bytes = sh.geom.buf_len;
    so.nh = nh;
    if (with_verify && collective) {
        // collective search: the records travel in the all-gather (gather_records), not through this host
    } else if (with_verify && view_ok && d.last_direct) {
        so.nr = nr;
        ctx->view = reinterpret_cast<const FzRec *>(d.h_stage + kHeaderBytes);
        ctx->view_n = nr;
    } else if (with_verify) {
        so.nr = nr;
        so.recs.resize(nr);
        const uint64_t first = d.last_direct ? nr : std::min<uint64_t>(nr, d.first_copy);
        if (first) memcpy(so.recs.data(), d.h_stage + kHeaderBytes, first * sizeof(FzRec));
        if (nr > first)
            HIP_TRY(hipMemcpy(so.recs.data() + first, d.d_out + kHeaderBytes + first * sizeof(FzRec),
                              (nr - first) * sizeof(FzRec), hipMemcpyDeviceToHost));
        d.first_copy = std::max<uint64_t>(512, nr + nr / 4 + 64);      // next call: fetch about this many
    } else {
        so.hits.resize(nh);
        if (nh) HIP_TRY(hipMemcpy(so.hits.data(), d.d_hits, nh * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return FZ_OK;
}

int check_halo(const fz_seq *seq, uint64_t need) {
    for (const Shard &sh : seq->shards) {
        const FzGeom &g = sh.geom;
        const uint64_t want_lo = g.own_lo > need ? g.own_lo - need : 0;
        const uint64_t want_hi = std::min<uint64_t>(g.n, g.own_hi + need);
        if (g.own_hi > g.own_lo && (g.buf_off > want_lo || g.buf_off + g.buf_len < want_hi))
            return fail(FZ_EHALO, "shard halo too small: need %llu bytes around [%llu, %llu)",
                        (unsigned long long)need, (unsigned long long)g.own_lo, (unsigned long long)g.own_hi);
    }
    return FZ_OK;
}

// fn(si) for every shard of the sequence: on the shard's device worker when the context has workers and the sequence
// several shards (one shard per device), else on the calling thread in shard order.  -> first error.
template <class F>
int for_each_shard(fz_ctx *ctx, fz_seq *seq, F fn) {
    const size_t ns = seq->shards.size();
    if (!ctx->workers || ns < 2) {
        for (size_t si = 0; si < ns; ++si) { int rc = fn(si); if (rc) return rc; }
        return FZ_OK;
    }
    for (size_t si = 0; si < ns; ++si) ctx->workers->post((size_t)seq->shards[si].dev, [fn, si]() { return fn(si); });
    int rc = FZ_OK;
    std::string err;
    for (size_t si = 0; si < ns; ++si) {
        const int r = ctx->workers->wait((size_t)seq->shards[si].dev);
        if (r && !rc) { rc = r; err = g_err; }
    }
    if (rc) g_err = err;
    return rc;
}

int comm_rank_lows(fz_ctx *ctx, fz_seq *seq);
int comm_gather_host(fz_ctx *ctx, const void *data, uint64_t bytes, std::vector<uint8_t> &all);
int comm_or(fz_ctx *ctx, bool &flag);
bool comm_multi_process(const fz_ctx *ctx);

// Host half of the exchange step: `blocks` = world blocks of `bytes_per_rank` bytes, each [1 KiB of counters][records];
// counters[1] = records the rank produced (it stored min(count, cap) of them).  -> *top = the largest count; if it is
// above `cap` nothing else happens (the caller re-gathers with a larger capacity: every rank sees the same headers and
// decides alike).  Else `recs` = the ranks' records rank by rank, seg_ends[r] = where rank r's end, seg_order = the
// ranks in ascending order of the index range they own (own_lo[r]; a rank that owns nothing sorts by its number).
void parse_gathered(const uint8_t *blocks, int world, uint64_t bytes_per_rank, uint64_t cap, const uint64_t *own_lo,
                    std::vector<FzRec> &recs, std::vector<size_t> &seg_ends, std::vector<uint32_t> &seg_order, uint64_t *top_out,
                    uint64_t *total_out) {
    uint64_t top = 0, total = 0;
    for (int r = 0; r < world; ++r) {
        const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(blocks + (uint64_t)r * bytes_per_rank);
        top = std::max<uint64_t>(top, cnt[1]);
        total += cnt[1];
    }
    *top_out = top;
    *total_out = total;
    recs.clear();
    seg_ends.clear();
    seg_order.clear();
    if (top > cap) return;
    recs.resize(total);
    uint64_t o = 0;
    for (int r = 0; r < world; ++r) {
        const uint8_t *blk = blocks + (uint64_t)r * bytes_per_rank;
        const uint64_t c = reinterpret_cast<const unsigned long long *>(blk)[1];
        if (c) memcpy(recs.data() + o, blk + kHeaderBytes, c * sizeof(FzRec));
        o += c;
        seg_ends.push_back(o);
        seg_order.push_back((uint32_t)r);
    }
    if (own_lo)
        std::stable_sort(seg_order.begin(), seg_order.end(), [&](uint32_t x, uint32_t y) { return own_lo[x] < own_lo[y]; });
}

// RCCL is loaded on first use (dlopen), not linked: a single-GPU install without librccl — or with ROCM_PATH pointing
// somewhere that lacks it — still loads libfzhip.so and searches; only fz_comm_* then return FZ_EUNSUPPORTED.
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
    bool ok = false;
    bool stand_in = false;                       // tests/mock_rccl.cpp (FZ_RCCL_LIB): N ranks on ONE device are possible
};

const RcclApi *rccl_api() {
    static const RcclApi api = []() {
        RcclApi a;
        if (sw().no_rccl) { a.error = "disabled by FZ_NO_RCCL"; return a; }      // test knob: an install without librccl
        std::vector<std::string> names;
        if (!sw().rccl_lib.empty()) names.push_back(sw().rccl_lib);
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back(sw().rocm_path + "/lib/librccl.so.1");
        void *h = nullptr;
        for (const std::string &n : names) {
            h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
            const char *de = dlerror();
            a.error += (a.error.empty() ? "" : "; ") + std::string(de ? de : n.c_str());
        }
        if (!h) return a;
        bool all = true;
        auto sym = [&](const char *name) { void *p = dlsym(h, name); if (!p) { all = false; a.error = std::string("librccl lacks ") + name; } return p; };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = all;
        a.stand_in = dlsym(h, "fzmock_rccl") != nullptr;
        return a;
    }();
    return &api;
}

#define RCCL_NEED()                                                                                \
    do {                                                                                           \
        if (!rccl_api()->ok) return fail(FZ_EUNSUPPORTED, "RCCL is not available (%s)", rccl_api()->error.c_str()); \
    } while (0)

#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(FZ_EDEVICE, "%s failed: %s", #expr, rccl_api()->GetErrorString(r_)); \
    } while (0)

// Every wait for a collective has a deadline (FZ_COMM_TIMEOUT_MS, default 60 s): a rank that never arrives — a peer process
// that died, a link that does not come up — is an error the caller can act on (bench.py prints its line with
// `collective_error` and the host-merged value), not a job that hangs until somebody kills it.  Polling (hipStreamQuery)
// instead of hipStreamSynchronize: spinning at first — a gather takes tens of microseconds —, yielding after 200 us.
int comm_wait(hipStream_t st, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    const long limit_ms = sw().comm_timeout_ms;
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return FZ_OK;
        (void)hipGetLastError();                           // (hipErrorNotReady is an answer, not a failure of a later launch)
        if (e != hipErrorNotReady) return fail(FZ_EDEVICE, "%s: %s", what, hipGetErrorString(e));
        const auto waited = std::chrono::steady_clock::now() - t0;
        if (waited > std::chrono::milliseconds(limit_ms))
            return fail(FZ_ETIMEOUT, "%s did not complete within %ld ms (FZ_COMM_TIMEOUT_MS): a rank of the communicator never arrived",
                        what, limit_ms);
        if (waited > std::chrono::microseconds(200)) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

// The exchange step of a sharded search: every rank of the communicator contributes the
// snapshot of its counters + records (FzRec, unordered), ONE ncclAllGather of kHeaderBytes + gcap records per
// rank (a group call over the device states of this process), one D2H copy of the gathered block, and every rank
// holds all records of the global search; the caller orders them by (block, index) as it does for one shard.
// gcap follows the counts: the gathered headers show every rank the same numbers, so all ranks grow / shrink
// alike; an overflow re-gathers from the same snapshots (they hold everything the search produced).
// The ranks' segments are ordered by the index ranges they own (exchanged once per sequence: comm_rank_lows).
int gather_records(fz_ctx *ctx, fz_seq *seq, std::vector<FzRec> &recs) {
    const int world = ctx->comm_world;
    if (world <= 0) return fail(FZ_EINVAL, "the context has not joined a communicator");
    if (ctx->comm_broken) return fail(FZ_ETIMEOUT, "the communicator was abandoned after a collective ran into its deadline: "
                                                   "fz_comm_set_collective(ctx, 0) searches without it");
    recs.clear();
    const auto t_start = std::chrono::steady_clock::now();
    int rcl = comm_rank_lows(ctx, seq);
    if (rcl) return rcl;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const uint64_t bytes = kHeaderBytes + ctx->gcap * sizeof(FzRec);
        for (DevState &d : ctx->devs) {
            HIP_TRY(hipSetDevice(d.device));
            int rc = ensure_send(d, std::max<uint64_t>(d.rec_cap, ctx->gcap));
            if (rc) return rc;
            if (d.recv_bytes < (uint64_t)world * bytes) {
                HIP_TRY(hipStreamSynchronize(d.comm_stream));
                if (d.d_recv) HIP_TRY(hipFree(d.d_recv));
                if (d.h_recv) HIP_TRY(hipHostFree(d.h_recv));
                d.d_recv = nullptr; d.h_recv = nullptr; d.recv_bytes = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_recv), (uint64_t)world * bytes));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&d.h_recv), (uint64_t)world * bytes, hipHostMallocDefault));
                d.recv_bytes = (uint64_t)world * bytes;
            }
            bool has_shard = false;
            for (const Shard &sh : seq->shards) has_shard |= &ctx->devs[sh.dev] == &d;
            if (has_shard && d.snap_taken[d.slot_id]) HIP_TRY(hipStreamWaitEvent(d.comm_stream, d.ev_snap[d.slot_id], 0));
            else HIP_TRY(hipMemsetAsync(d.d_send[d.slot_id], 0, kHeaderBytes, d.comm_stream));   // this rank holds nothing of the sequence
        }
        NCCL_TRY(rccl_api()->GroupStart());
        for (DevState &d : ctx->devs)
            NCCL_TRY(rccl_api()->AllGather(d.d_send[d.slot_id], d.d_recv, bytes, ncclChar, d.comm, d.comm_stream));
        NCCL_TRY(rccl_api()->GroupEnd());
        DevState &d0 = ctx->devs[0];
        HIP_TRY(hipSetDevice(d0.device));
        HIP_TRY(hipMemcpyAsync(d0.h_recv, d0.d_recv, (uint64_t)world * bytes, hipMemcpyDeviceToHost, d0.comm_stream));
        HIP_TRY(hipEventRecord(d0.ev_done, d0.comm_stream));
        {
            int rcw = comm_wait(d0.comm_stream, "the all-gather of the ranks' records");
            if (rcw) { if (rcw == FZ_ETIMEOUT) ctx->comm_broken = true; return rcw; }
        }
        for (size_t i = 1; i < ctx->devs.size(); ++i) {      // the other device states of this process: only wait
            HIP_TRY(hipSetDevice(ctx->devs[i].device));
            int rcw = comm_wait(ctx->devs[i].comm_stream, "the all-gather of the ranks' records");
            if (rcw) { if (rcw == FZ_ETIMEOUT) ctx->comm_broken = true; return rcw; }
        }
        uint64_t top = 0, total = 0;
        parse_gathered(d0.h_recv, world, bytes, ctx->gcap, seq->rank_lo.data(), recs, ctx->seg_ends, ctx->seg_order, &top, &total);
        if (top > ctx->gcap) {                               // identical decision on every rank
            ctx->gcap = (top + top / 4 + 1023) / 1024 * 1024;
            continue;
        }
        ctx->stats.raw_matches = total;
        const uint64_t want = std::max<uint64_t>(1024, (top + top / 4 + 1023) / 1024 * 1024);
        if (want * 2 <= ctx->gcap) ctx->gcap = want;         // follow the counts down as well (hysteresis: a factor of two)
        ctx->last_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        return FZ_OK;
    }
    return fail(FZ_EDEVICE, "all-gather capacity kept overflowing");
}

// Launch a search on every shard (no host synchronisation).
int search_enqueue(fz_ctx *ctx, fz_seq *seq, const Search &q, bool with_verify) {
    ctx->view = nullptr;
    ctx->view_n = 0;
    ctx->rows_ready = false;
    ctx->stats.filter_launches = 0;
    ctx->stats.bytes_scanned = ctx->stats.ngram_hits = ctx->stats.raw_matches = 0;
    ctx->stats.filter_ms = ctx->stats.verify_ms = ctx->stats.device_ms = 0;
    ctx->tref.clear();
    int rc = for_each_shard(ctx, seq, [ctx, seq, &q, with_verify](size_t si) { return enqueue_shard(ctx, seq->shards[si], q, with_verify); });
    if (rc) return rc;
    for (const Shard &sh : seq->shards) {
        const DevState &d = lane_dev(ctx, sh.dev);
        ctx->stats.filter_launches += d.launches_used;
        ctx->stats.verify_form = d.form_used;
        ctx->last_fused = d.fused_used;
    }
    return FZ_OK;
}

int emit_matches(const FzRec *recs, size_t cnt, uint32_t L, fz_match **out, uint64_t *n, uint64_t idx_bound, uint32_t blk_bound,
                 bool may_have_empty, std::vector<fz_match> *into);

// Wait for the search launched by search_enqueue and collect it; on an overflow the buffers have
// been grown and the search is launched again (deterministic, at most three times).
// One shard: the records are in `recs` (or the context's view of the staging buffer), the hits in `hits`.
// Several shards: every shard's records are collected AND ordered on its own (ctx->souts[si].rows, by the device's
// worker when the context has workers) and emit_matches only merges the rows (ctx->rows_ready); hits are concatenated.
int search_collect(fz_ctx *ctx, fz_seq *seq, const Search &q, bool with_verify, std::vector<FzRec> &recs,
                   std::vector<uint64_t> &hits) {
    const size_t ns = seq->shards.size();
    if (ctx->souts.size() < ns) ctx->souts.resize(ns);
    const bool order_rows = with_verify && !q.collective && ns > 1 && !q.any && seq->shards[0].geom.seg_stride == 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        recs.clear();
        hits.clear();
        ctx->rows_ready = false;
        // the statistics describe the search being collected (a younger one may have been launched meanwhile)
        ctx->stats.bytes_scanned = ctx->stats.ngram_hits = ctx->stats.raw_matches = 0;
        ctx->stats.filter_ms = ctx->stats.verify_ms = ctx->stats.device_ms = 0;
        ctx->tref.clear();
        Trace tr;
        ctx->seg_ends.clear();
        ctx->seg_order.clear();
        const uint32_t L = q.plan.L, nblk = (uint32_t)q.plan.s.size();
        const uint64_t nglob = seq->n;
        const bool collective = q.collective;
        int rc = for_each_shard(ctx, seq, [ctx, seq, with_verify, ns, collective, order_rows, L, nblk, nglob](size_t si) {
            ShardOut &so = ctx->souts[si];
            int r = collect_shard(ctx, seq->shards[si], with_verify, ns == 1, so, collective);
            if (r || so.rerun || !order_rows) return r;
            const bool dense = ctx->devs[seq->shards[si].dev].fused_used;
            return emit_matches(so.recs.data(), so.recs.size(), L, nullptr, nullptr, nglob, nblk, !dense, &so.rows);
        });
        if (rc) return rc;
        bool any_rerun = false;
        for (size_t si = 0; si < ns; ++si) {
            ShardOut &so = ctx->souts[si];
            any_rerun |= so.rerun;
            if (so.rerun) continue;
            ctx->stats.bytes_scanned += so.bytes;
            ctx->stats.ngram_hits += so.nh;
            ctx->stats.raw_matches += so.nr;
            if (so.has_tref) ctx->tref.push_back({so.td, so.tev[0], so.tev[1], so.tev[2], so.tev[3], so.tev[4], so.tev[5]});
        }
        tr.mark(" collect");
        if (!any_rerun) {
            if (with_verify && q.collective) {
                int rc2 = gather_records(ctx, seq, recs);
                if (rc2) return rc2;
                tr.mark(" all-gather");
                return FZ_OK;
            }
            if (ns == 1) {
                recs.swap(ctx->souts[0].recs);
                hits.swap(ctx->souts[0].hits);
            } else if (order_rows) {
                ctx->rows_ready = true;
                for (uint32_t si = 0; si < ns; ++si) ctx->seg_order.push_back(si);
                std::sort(ctx->seg_order.begin(), ctx->seg_order.end(), [&](uint32_t x, uint32_t y) {
                    return seq->shards[x].geom.own_lo < seq->shards[y].geom.own_lo;
                });
            } else {
                for (size_t si = 0; si < ns; ++si) {
                    recs.insert(recs.end(), ctx->souts[si].recs.begin(), ctx->souts[si].recs.end());
                    hits.insert(hits.end(), ctx->souts[si].hits.begin(), ctx->souts[si].hits.end());
                }
            }
            return FZ_OK;
        }
        rc = search_enqueue(ctx, seq, q, with_verify);
        if (rc) return rc;
    }
    return fail(FZ_EDEVICE, "result buffers kept overflowing");
}

int run_search(fz_ctx *ctx, fz_seq *seq, const Search &q, bool with_verify, std::vector<FzRec> &recs,
               std::vector<uint64_t> &hits) {
    if (ctx->npend) return fail(FZ_EINVAL, "a search started with fz_lev_ngrams_begin is still in flight");
    memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    Trace tr;
    int rc = search_enqueue(ctx, seq, q, with_verify);
    if (rc) return rc;
    tr.mark(" enqueue");
    return search_collect(ctx, seq, q, with_verify, recs, hits);
}

// Generic search: scan (emit exact hits) -> fz_generic_kernel (one wave per hit) -> records.
// Re-runs with larger buffers on overflow: hit list, record list, then candidate lists.
// `phase`: 0 = the whole search; 1 = launch only (fz_generic_ngrams_begin: the kernels of the first attempt are enqueued on
// the current lane and the call returns); 2 = collect what phase 1 launched (fz_search_end), then carry on as phase 0 if
// a buffer turned out too small.
int run_generic(fz_ctx *ctx, fz_seq *seq, const Search &q, std::vector<FzGenRec> &recs_out, int phase = 0) {
    memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    static_assert(sizeof(FzGenRec) == sizeof(FzRec), "record buffers are shared");
    uint32_t cand_cap = ctx->gen_cand_cap;
    bool legacy_rest = false;                                  // the remaining attempts of this search use fz_lp_kernel
    for (int attempt = 0; attempt < 10; ++attempt) {
        recs_out.clear();
        ctx->any_found = false;
        ctx->gen_view = nullptr;
        ctx->gen_view_n = 0;
        ctx->gen_rows_dev = nullptr;
        ctx->gen_rows_n = 0;
        bool rerun = false;
        bool multi_failed = false, multi_ok = false;
        const bool legacy_now = legacy_rest || (attempt == 0 && phase != 2 && ctx->gen_multi_skip > 0);
        if (attempt == 0 && phase != 2 && ctx->gen_multi_skip > 0) { --ctx->gen_multi_skip; legacy_rest = true; }
        ctx->stats.filter_launches = 0;
        ctx->stats.bytes_scanned = ctx->stats.ngram_hits = ctx->stats.raw_matches = 0;
        ctx->stats.filter_ms = ctx->stats.verify_ms = ctx->stats.device_ms = 0;
        ctx->tref.clear();
        const uint32_t mpad = (q.m + 15u) & ~15u, wpad = (q.m + 2 * q.k + 15u) & ~15u;
        for (const Shard &sh : seq->shards) {
            if (phase == 2 && attempt == 0) break;            // launched by fz_generic_ngrams_begin
            DevState &d = lane_dev(ctx, sh.dev);
            // Window table: the hits that the n-gram blocks of one occurrence produce share their window, the automaton runs
            // once per window (fz_device.h: FzGenDedup; the scan fills the table as it lists the hits).  Where the rows are
            // finished on the device, where only (hull, best) pairs leave it, and for the flag-only search;
            // FZ_GEN_NO_DEDUP=1: every hit on its own (A/B, tests).
            const bool no_dedup = sw().gen_no_dedup;
            const bool gen_direct0 = sw().gen_direct, host_order0 = sw().gen_host_order;
            const bool dev_order0 = !gen_direct0 && !host_order0 && !q.any && !q.fold && seq->shards.size() == 1 && sh.geom.seg_stride == 0 &&
                                    !comm_multi_process(ctx);
            const bool dedup = !no_dedup && sh.geom.seg_stride == 0 && (dev_order0 || q.fold || q.any);
            d.dedup_used = dedup;
            d.gen_dedup_arg = 0;
            if (dedup) {
                HIP_TRY(hipSetDevice(d.device));
                if (!d.d_gen_dedup) {
                    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_gen_dedup), FZ_GEN_DEDUP_BYTES));
                    d.dedup_zeroed = false;
                }
                if (!d.dedup_zeroed) HIP_TRY(hipMemsetAsync(d.d_gen_dedup, 0, FZ_GEN_DEDUP_ZERO_BYTES, d.stream));
                d.dedup_zeroed = false;
                d.gen_dedup_arg = reinterpret_cast<uint64_t>(d.d_gen_dedup);
            }
            int rc = enqueue_shard(ctx, sh, q, /*with_verify=*/false, /*copy_back=*/false);
            if (rc) return rc;
            size_t lds = 0;
            uint64_t scratch = 0;
            rc = cand_lists(d, cand_cap, mpad + wpad + FZ_GEN_MCAP * 8, lds, scratch);
            if (rc) return rc;
            FzScanArgs fa;
            fill_common_args(fa, sh, q);
            rc = stage_pattern(d, fa, q.p, q.m);
            if (rc) return rc;
            fa.flags = (q.any ? FZ_FLAG_ANY : 0u) | (q.fold ? FZ_FLAG_FOLD : 0u);
            fa.cand_cap = cand_cap;
            fa.cand_scratch = scratch;
            fa.lp_kind = FZ_LP_GENERIC_HIT;
            fa.lp_starts = 0;
            rc = ensure_big(d, 1u << 16);
            if (rc) return rc;
            fa.hit_cap = d.hit_cap;
            // Records of the automaton: into the device buffer, fetched with ONE copy once the count is known.
            // (Round 1 let the kernel store them straight into pinned host memory: 2.1e5 scattered 24-byte stores
            // cross PCIe at ~10 GB/s and the kernel cannot finish before they have drained — 0.49 ms for a kernel
            // whose work takes a fraction of that; FZ_GEN_DIRECT=1 restores that path.)
            const bool gen_direct = sw().gen_direct;
            // One shard, no segments: the rows are ordered and finished on the device (fz_gen_order_kernel,
            // fz_gen_scatter_kernel) and cross PCIe once, straight into the caller's buffer; FZ_GEN_HOST_ORDER=1
            // keeps the host's run ordering (emit_generic), which also serves searches with more than
            // FZ_GEN_ORDER_MAX hits, several shards and the file API's segments.
            const bool host_order = sw().gen_host_order;
            const bool dev_order = !gen_direct && !host_order && !q.any && !q.fold && seq->shards.size() == 1 && sh.geom.seg_stride == 0 && !comm_multi_process(ctx);
            if (dev_order) {
                rc = ensure_gen_rows(d);
                if (rc) return rc;
                fa.gen_order = reinterpret_cast<uint64_t>(d.d_gen_order);
                fa.rows_cap = d.gen_rows_cap;
            }
            fa.gen_dedup = d.gen_dedup_arg;
            // Folded search (a few thousand pairs): the automaton kernel writes them straight into the pinned staging
            // buffer and its last workgroup publishes the counters there, as the fused scan does for its records — no
            // copy command between the kernel and the host (FZ_NO_DIRECT=1, or more pairs than the buffer holds: the copy).
            const bool no_direct = env_no_direct();
            const bool fold_direct = q.fold && !gen_direct && !no_direct && d.fold_direct && seq->shards.size() == 1;
            d.fold_was_direct = fold_direct;
            fa.rec_cap = gen_direct ? d.big_cap : fold_direct ? kHostRecs : d.rec_cap;
            if (fold_direct) fa.host_hdr = reinterpret_cast<uint64_t>(d.h_stage_dev);
            unsigned long long *counters = reinterpret_cast<unsigned long long *>(d.d_out);
            FzGenRec *recs = gen_direct ? reinterpret_cast<FzGenRec *>(d.h_big_dev)
                           : fold_direct ? reinterpret_cast<FzGenRec *>(d.h_stage_dev + kHeaderBytes)
                                         : reinterpret_cast<FzGenRec *>(d.d_out + kHeaderBytes);
            static_assert(sizeof(FzGenRec) == sizeof(FzRec), "the generic records share the record buffer");
            // fz_gen_hit_kernel: four waves per hit, every wave a quarter of the list (in-memory searches with the lists in LDS)
            const uint32_t gh_waves_env = (uint32_t)sw().gh_waves;
            const bool gh_no_bits = sw().gh_no_bits;
            // the bit-parallel form (fz_gen_hit_kernel<W, true>: 64-bit equality words, flags as words, unconditional stores):
            // patterns of at most 64 characters and budgets of at most 32.  FZ_GH_NO_BITS=1: the round-4 step (A/B, tests);
            // FZ_GH_WAVES=1 / 2 / 4: waves per hit.  Default with the bit-parallel form: ONE — measured in round 5 (profiles/
            // r05_generic_kernels.txt): a hit's time is ~1 500 cycles per window character that has candidates, whatever the
            // number of waves or of slices per trip (the big tree of a true occurrence belongs to one start, and the per-character
            // cost is the dependent chain of one trip), so more waves per hit only cost residency; 2 otherwise (round 4)
            // fz_gen_hit_kernel<W, true>'s implicit bounds, held HERE: one 64-bit equality word per window character (m <= 64),
            // the equality words of a window in two register pairs (m + 2k <= 128), skip counts in 32 bits (k <= 32), counters
            // below 2^31 (limits <= FZ_MAX_K = 255); anything else takes round 4's step
            static_assert(FZ_MAX_K < (1u << 31), "fz_b_lt / fz_b_eq operands");
            const bool gh_bits = q.m <= 64u && q.k <= 32u && q.m >= 1u && q.m + 2u * q.k <= 128u &&
                                 std::max(std::max(q.max_subs, q.max_ins), std::max(q.max_dels, q.k)) <= FZ_MAX_K && !gh_no_bits;
            const uint32_t gh_waves = gh_waves_env == 4u || gh_waves_env == 2u || (gh_waves_env == 1u && gh_bits) ? gh_waves_env : gh_bits ? 1u : 2u;
            const uint32_t capw = gh_waves == 1u ? std::max<uint32_t>(64u, cand_cap) : std::max<uint32_t>(64u, cand_cap / 2u);   // slots per list of one wave
            const size_t lds_multi = (size_t)mpad + wpad + FZ_GH_CTL_BYTES + (size_t)gh_waves * 2u * capw * sizeof(FzGCand) +
                                     (size_t)gh_waves * FZ_GH_MCAP_W(gh_waves) * 8u +
                                     (gh_bits ? (size_t)FZ_GH_PT_BYTES + (size_t)wpad * 8u + (size_t)64u * gh_waves * 8u : 0u);
            using GhKernel = void (*)(const uint8_t *, const FzScanArgs, const uint64_t *, FzGenRec *, unsigned long long *);
            const GhKernel gh = gh_bits ? (gh_waves == 1u ? fz_gen_hit_kernel<1, true> : gh_waves == 2u ? fz_gen_hit_kernel<2, true> : fz_gen_hit_kernel<4, true>)
                                        : (gh_waves == 4u ? fz_gen_hit_kernel<4, false> : fz_gen_hit_kernel<2, false>);
            const bool multi = ctx->gen_multi && !legacy_now && scratch == 0 && sh.geom.seg_stride == 0 && lds_multi <= 160 * 1024;
            d.gen_multi_used = multi;
            if (multi) {
                fa.cand_cap = capw;
                if (lds_multi > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(gh), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi));
            }
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lp_kernel(FZ_LP_GENERIC_HIT, scratch != 0)),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned grid_per_cu = (unsigned)sw().lp_grid_per_cu;   // lab knob
            // (measured on configs[3b], 6144 hits: 16 / 24 / 32 workgroups per CU with 512-entry match buffers 0.325 /
            //  0.310 / 0.310 ms, with 128-entry ones 0.325 / 0.303 / 0.304 ms — once every hit is resident the kernel
            //  takes as long as its slowest hit)
            // (Lab knob FZ_GEN_HI_STREAM=1, measured and NOT the default: for a search in flight next to another one,
            // everything behind the scan on a HIGH-priority stream of the lane, so that the automaton's waves take the
            // slots the other lane's scan frees instead of waiting behind its grid.  configs[3b], two in flight: 0.364 ms
            // per search against 0.310 ms with the automaton on the lane's own low-priority stream — the automaton's
            // 4 171 waves then hold their CUs' registers for 160 us and the scan, which is what bounds the pair, starves.)
            const bool use_hi = sw().gen_hi_stream;
            hipStream_t st2 = d.stream;
            if (phase == 1 && use_hi) {
                if (!d.stream_hi) {
                    int prio_least = 0, prio_greatest = 0;
                    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
                    HIP_TRY(hipStreamCreateWithPriority(&d.stream_hi, hipStreamNonBlocking, prio_greatest));
                    HIP_TRY(hipEventCreateWithFlags(&d.ev_scan_done, hipEventDisableTiming));
                }
                HIP_TRY(hipEventRecord(d.ev_scan_done, d.stream));
                HIP_TRY(hipStreamWaitEvent(d.stream_hi, d.ev_scan_done, 0));
                st2 = d.stream_hi;
            }
            // the automaton's end event rides on its own launch; for a folded search in direct mode that event is also
            // the search's completion (ev[3]): no event packet of its own anywhere in such a search
            const bool no_ext = sw().no_ext_launch;
            hipEvent_t lp_stop = no_ext ? nullptr : fold_direct ? d.ev[3] : ctx->timing ? d.ev[2] : nullptr;
            d.lp_end_event = fold_direct ? 3 : 2;
            const unsigned multi_per_cu_env = (unsigned)sw().gh_grid_per_cu;
            const unsigned multi_per_cu = multi_per_cu_env ? multi_per_cu_env : 32u;   // lab knob (configs[3b]: 16 -> 0.130 ms, 32 -> 0.104: every hit of the search needs a workgroup of its own)
            if (multi && lp_stop)
                hipExtLaunchKernelGGL(gh, dim3(d.n_cus * multi_per_cu), dim3(64 * gh_waves), lds_multi, st2, nullptr, lp_stop, 0u,
                                      sh.d_buf, fa, d.d_hits, recs, counters);
            else if (multi)
                hipLaunchKernelGGL(gh, dim3(d.n_cus * multi_per_cu), dim3(64 * gh_waves), lds_multi, st2, sh.d_buf, fa, d.d_hits,
                                   recs, counters);
            else if (lp_stop)
                hipExtLaunchKernelGGL(lp_kernel(FZ_LP_GENERIC_HIT, scratch != 0), dim3(scratch ? kCandScratchGrid : d.n_cus * grid_per_cu),
                                      dim3(64), lds, st2, nullptr, lp_stop, 0u, sh.d_buf, fa, d.d_hits, (uint64_t)0, recs, counters);
            else
                hipLaunchKernelGGL(lp_kernel(FZ_LP_GENERIC_HIT, scratch != 0), dim3(scratch ? kCandScratchGrid : d.n_cus * grid_per_cu), dim3(64),
                                   lds, st2, sh.d_buf, fa, d.d_hits, (uint64_t)0, recs, counters);
            HIP_TRY(hipGetLastError());
            if (ctx->timing && !lp_stop && !fold_direct) HIP_TRY(hipEventRecord(d.ev[2], st2));
            if (dev_order) {
                hipLaunchKernelGGL(fz_gen_order_kernel, dim3(d.n_cus * 8), dim3(256), 0, st2, d.d_hits, fa, counters);
                hipLaunchKernelGGL(fz_gen_scatter_kernel, dim3(d.n_cus * 8), dim3(256), 0, st2, d.d_hits, fa, recs,
                                   reinterpret_cast<FzOutRow *>(d.d_gen_rows), counters);
                HIP_TRY(hipGetLastError());
            }
            // folded search: the (few) pairs come back with the counters in ONE copy — as many as the previous search had
            d.fold_copied = fold_direct ? kHostRecs : q.fold ? std::min<uint64_t>(std::min<uint64_t>(d.fold_guess, kHostRecs), d.rec_cap) : 0;
            if (!fold_direct) {
                HIP_TRY(hipMemcpyAsync(d.h_stage, d.d_out, kHeaderBytes + d.fold_copied * sizeof(FzGenRec), hipMemcpyDeviceToHost, st2));
                HIP_TRY(hipEventRecord(d.ev[3], st2));
                // the counters are zeroed for the next search now, behind the copy (not in front of that search's scan)
                if (st2 == d.stream) {
                    HIP_TRY(hipMemsetAsync(d.d_out, 0, kHeaderBytes, st2));
                    d.header_zeroed = true;
                }
            } else {
                if (!lp_stop) HIP_TRY(hipEventRecord(d.ev[3], st2));
                d.header_zeroed = true;                       // by the publishing workgroup
            }
            if (d.dedup_used && st2 == d.stream) {            // the window table is cleared for the next search, behind this one
                HIP_TRY(hipMemsetAsync(d.d_gen_dedup, 0, FZ_GEN_DEDUP_ZERO_BYTES, d.stream));
                d.dedup_zeroed = true;
            }
        }
        if (phase == 1) return FZ_OK;
        bool lists_overflowed = false;
        Trace trg;
        trg.mark(" generic enqueue");
        for (const Shard &sh : seq->shards) {
            DevState &d = lane_dev(ctx, sh.dev);
            HIP_TRY(hipSetDevice(d.device));
            HIP_TRY(hipEventSynchronize(d.ev[3]));            // (recorded behind the last command of the search, on whichever stream)
            trg.mark(" generic sync");
            const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(d.h_stage);
            const uint64_t nh = cnt[0], nr = cnt[1];
            uint64_t novf = cnt[2];
            if (d.gen_multi_used && cnt[FZ_HDR_GEN_FAIL]) {   // a hit outgrew a wave's share of the list / its match buffer: one wave per hit
                multi_failed = true;
                if (q.any && nr > 0) { ctx->any_found = true; continue; }
                rerun = true;
                continue;
            }
            if (d.gen_multi_used) multi_ok = true;
            // rows of the ordered search: with the window table every hit takes its window's matches, computed once
            const uint64_t nrows = d.dedup_used && !q.fold && !q.any && nh <= FZ_GEN_ORDER_MAX ? cnt[FZ_HDR_GEN_ROWS] : nr;
            if (q.any) {                                      // has_near_match_generic_ngrams: a record anywhere settles it
                ctx->any_found |= nr > 0;
                if (nr > 0) continue;
                if (nh > d.hit_cap) { int rc = ensure_hits(d, nh + nh / 8 + 1024); if (rc) return rc; rerun = true; }
                if (novf) { lists_overflowed = true; rerun = true; }
                ctx->stats.bytes_scanned += sh.geom.buf_len;
                ctx->stats.ngram_hits += nh;
                continue;
            }
            const bool gen_direct2 = sw().gen_direct;
            if (nh > d.hit_cap) { int rc = ensure_hits(d, nh + nh / 8 + 1024); if (rc) return rc; rerun = true; }
            if (nr > d.big_cap) { int rc = ensure_big(d, nr + nr / 8 + 1024); if (rc) return rc; if (gen_direct2) rerun = true; }
            if (d.fold_was_direct) {                          // pairs beyond the staging buffer were dropped: again, through d_out
                if (nr > kHostRecs) { d.fold_direct = false; rerun = true; }
            } else if (q.fold && nr * 4 < kHostRecs) d.fold_direct = true;
            if (!gen_direct2 && !d.fold_was_direct && std::max(nr, nrows) > d.rec_cap) { int rc = ensure_recs(d, std::max(nr, nrows) + nrows / 8 + 1024); if (rc) return rc; rerun = true; }
            const bool host_order2 = sw().gen_host_order;
            const bool rows_ready = !gen_direct2 && !host_order2 && !q.fold && seq->shards.size() == 1 && sh.geom.seg_stride == 0 &&
                                    nh <= FZ_GEN_ORDER_MAX && nrows <= d.gen_rows_cap && !comm_multi_process(ctx);
            // Window table + device ordering: only the windows' LEADERS left automaton records, the members' rows exist on the
            // device-ordered path alone.  If that path cannot be taken for this attempt (the row buffer is smaller than the row
            // count: the record buffers were regrown between the launch and this collection, e.g. by the other lane's search),
            // the host-ordered fallback below would silently miss the members' rows — run again, the launch sizes the row buffer
            // by the record capacity (ensure_gen_rows).  Not reachable with today's capacity bookkeeping (rows <= rec_cap is
            // checked above); kept as the invariant's enforcement.
            if (d.dedup_used && !q.fold && !q.any && !rows_ready && !rerun && nh <= FZ_GEN_ORDER_MAX && nrows > d.gen_rows_cap) {
                if (d.rec_cap < nrows) { int rc = ensure_recs(d, nrows + nrows / 8 + 1024); if (rc) return rc; }
                rerun = true;
            }
            const bool in_stage = q.fold && nr <= d.fold_copied && seq->shards.size() == 1;   // the pairs are in h_stage already
            if (q.fold) d.fold_guess = std::max<uint64_t>(4096, nr + nr / 4 + 256);
            if (!gen_direct2 && !rerun && !novf && nr && !rows_ready && !in_stage)
                HIP_TRY(hipMemcpy(d.h_big, d.d_out + kHeaderBytes, nr * sizeof(FzGenRec), hipMemcpyDeviceToHost));
            if (novf) { lists_overflowed = true; rerun = true; }
            if (rerun) continue;
            if (d.timed) ctx->tref.push_back({&d, d.ev[0], d.ev[1], d.ev[1], d.ev[d.lp_end_event], d.ev[0], d.ev[3]});
            ctx->stats.bytes_scanned += sh.geom.buf_len;
            ctx->stats.ngram_hits += nh;
            if (rows_ready) {                                 // finished rows, fetched by emit_generic
                ctx->gen_rows_dev = d.d_gen_rows;
                ctx->gen_rows_n = nrows;
                ctx->gen_rows_device = d.device;
            } else if (seq->shards.size() == 1) {             // read in place (valid until the next search)
                ctx->gen_view = reinterpret_cast<const FzGenRec *>(in_stage ? d.h_stage + kHeaderBytes : d.h_big);
                ctx->gen_view_n = nr;
            } else {
                const size_t base = recs_out.size();
                recs_out.resize(base + nr);
                if (nr) memcpy(recs_out.data() + base, d.h_big, nr * sizeof(FzGenRec));
            }
        }
        if (multi_failed) {
            legacy_rest = true;
            ctx->gen_multi_backoff = ctx->gen_multi_backoff ? std::min<uint32_t>(64u, ctx->gen_multi_backoff * 2u) : 1u;
            ctx->gen_multi_skip = ctx->gen_multi_backoff;
        } else if (multi_ok && !rerun) {
            ctx->gen_multi_backoff = 0;
        }
        if (lists_overflowed) cand_cap *= 4;
        if (q.any && ctx->any_found) return FZ_OK;
        if (!rerun) {
            ctx->gen_cand_cap = std::min<uint32_t>(cand_cap, kCandLdsMax);   // remember what worked, never the HBM fallback
            return FZ_OK;
        }
    }
    return fail(FZ_EUNSUPPORTED, "generic search: candidate lists / result buffers kept overflowing");
}

// Result buffers handed to the caller (released with fz_free).  Large ones (a generic search over 1 GiB of text
// returns 2.1e5 rows = 5 MB) are recycled: glibc serves them with mmap and gives them back with munmap, so every
// call would fault in ~1200 fresh pages while it writes the rows (~0.3 ms of a 1.4 ms call).  A header in front of
// the payload carries the capacity; fz_free parks up to two big blocks, alloc_out takes a parked block that fits.
struct OutHdr { uint64_t bytes; uint64_t magic; };
constexpr uint64_t kOutMagic = 0x667a6f7574627566ull;
constexpr uint64_t kOutRecycleMin = 32u << 10;
std::mutex g_out_lock;
OutHdr *g_out_parked[2] = {nullptr, nullptr};

int alloc_out(uint64_t n, size_t elem, void **out) {
    const uint64_t need = std::max<uint64_t>(1, n * elem);
    OutHdr *h = nullptr;
    if (need >= kOutRecycleMin) {
        std::lock_guard<std::mutex> g(g_out_lock);
        for (OutHdr *&slot : g_out_parked)
            if (slot && slot->bytes >= need && slot->bytes <= 4 * need) { h = slot; slot = nullptr; break; }
    }
    if (!h) {
        h = static_cast<OutHdr *>(malloc(sizeof(OutHdr) + need));
        if (!h) { *out = nullptr; return fail(FZ_ENOMEM, "out of host memory for %llu results", (unsigned long long)n); }
        h->bytes = need;
        h->magic = kOutMagic;
    }
    *out = h + 1;
    return FZ_OK;
}

void release_out(void *p) {
    if (!p) return;
    OutHdr *h = static_cast<OutHdr *>(p) - 1;
    if (h->magic != kOutMagic) { fprintf(stderr, "fz_free: not a libfzhip result buffer\n"); abort(); }
    if (h->bytes >= kOutRecycleMin) {
        std::lock_guard<std::mutex> g(g_out_lock);
        for (OutHdr *&slot : g_out_parked)
            if (!slot) { slot = h; return; }
    }
    h->magic = 0;
    free(h);
}

int validate(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, bool in_pipeline = false) {
    if (!ctx || !seq || seq->ctx != ctx) return fail(FZ_EINVAL, "bad ctx/seq handle");
    if (ctx->npend && !in_pipeline) return fail(FZ_EINVAL, "a search started with fz_lev_ngrams_begin is still in flight");
    if (ctx->stream_inflight) return fail(FZ_EINVAL, "a file stream of this context has a batch in flight (finish or close it first)");
    if (!p || m == 0) return fail(FZ_EINVAL, "subsequence must not be empty");
    if (m > FZ_MAX_M_ANY) return fail(FZ_EUNSUPPORTED, "subsequence longer than %u bytes", FZ_MAX_M_ANY);
    return FZ_OK;
}

// Order records by key = (block, idx): LSD radix sort over the key bytes that actually vary
// (O(M) — a comparison sort of the 24-byte records costs ~40 ns per record on the host).
void sort_recs(std::vector<FzRec> &recs) {
    const size_t n = recs.size();
    if (n < 2) return;
    if (n < 64) {
        std::sort(recs.begin(), recs.end(), [](const FzRec &a, const FzRec &b) { return a.key < b.key; });
        return;
    }
    uint64_t diff = 0;
    for (size_t i = 1; i < n; ++i) diff |= recs[i].key ^ recs[0].key;
    std::vector<FzRec> tmp(n);
    FzRec *src = recs.data(), *dst = tmp.data();
    for (int byte = 0; byte < 8; ++byte) {
        if (((diff >> (8 * byte)) & 0xff) == 0) continue;
        size_t count[257] = {0};
        for (size_t i = 0; i < n; ++i) ++count[((src[i].key >> (8 * byte)) & 0xff) + 1];
        for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < n; ++i) dst[count[(src[i].key >> (8 * byte)) & 0xff]++] = src[i];
        std::swap(src, dst);
    }
    if (src != recs.data()) memcpy(recs.data(), src, n * sizeof(FzRec));
}

// Records -> fz_match rows in (block, index) order == the reference's emission order.
// The keys of one search differ in few bits (block number + an index range), so they are first
// squeezed next to the record's position into one 64-bit word and those words are sorted with 11-bit
// LSD radix passes over the key bits only (8-byte elements instead of 24-byte records); the records
// are then read once, in order.
// Slot-per-hit kernels leave FZ_REC_NONE in the slots of hits that did not verify.
bool drop_empty_slots(const FzRec *recs, size_t cnt, std::vector<FzRec> &kept) {
    size_t empty = 0;
    for (size_t i = 0; i < cnt; ++i) empty += recs[i].dist == FZ_REC_NONE;
    if (!empty) return false;
    kept.clear();
    kept.reserve(cnt - empty);
    for (size_t i = 0; i < cnt; ++i) if (recs[i].dist != FZ_REC_NONE) kept.push_back(recs[i]);
    return true;
}

// idx_bound / blk_bound (0: unknown): every hit index is below idx_bound and every block number below blk_bound
// (sequence length and block count of the search) — saves the pass that finds the key ranges; may_have_empty: some
// records may be empty slots (the slot-per-hit verification).
// `into` (optional): the rows go into this vector instead of a result buffer (per-shard ordering of a sharded search).
int emit_matches(const FzRec *recs, size_t cnt, uint32_t L, fz_match **out, uint64_t *n, uint64_t idx_bound = 0,
                 uint32_t blk_bound = 0, bool may_have_empty = true, std::vector<fz_match> *into = nullptr) {
    size_t nv = 0;
    uint64_t imin = ~0ull, imax = 0;
    uint32_t gmax = 0;
    // key ranges known from the search (sequence length, number of blocks): no pass over the records for them — the
    // records sit in pinned memory the device has just written, and the first pass over them is the cold one; empty
    // slots (slot-per-hit kernels) are then dropped while the key words are built
    const bool bounds_known = idx_bound && blk_bound;
    if (bounds_known) {
        nv = cnt; imin = 0; imax = idx_bound - 1; gmax = blk_bound - 1;
    } else {
        // one pass for the empty slots (slot-per-hit kernels) and the key ranges
        for (size_t i = 0; i < cnt; ++i) {
            if (recs[i].dist == FZ_REC_NONE) continue;
            const uint64_t idx = fz_hit_index(recs[i].key);
            imin = std::min(imin, idx); imax = std::max(imax, idx);
            gmax = std::max(gmax, fz_hit_block(recs[i].key));
            ++nv;
        }
    }
    fz_match *mo = nullptr;
    auto make_room = [&]() -> int {                            // nv is final
        if (into) {
            into->resize(nv);
            mo = into->data();
            return FZ_OK;
        }
        void *mem = nullptr;
        int rc = alloc_out(nv, sizeof(fz_match), &mem);
        if (rc) return rc;
        mo = static_cast<fz_match *>(mem);
        *out = mo;
        *n = nv;
        return FZ_OK;
    };
    const bool count_later = bounds_known && may_have_empty && cnt > 0;   // nv = cnt is an upper bound until the key pass
    if (!count_later) { int rc = make_room(); if (rc) return rc; }
    auto put = [&](size_t i, const FzRec &r) {
        const uint64_t idx = fz_hit_index(r.key);
        mo[i].start = (int64_t)(idx - r.l);
        mo[i].end = (int64_t)(idx + L + r.r);
        mo[i].dist = (int32_t)r.dist;
        mo[i].block = (int32_t)fz_hit_block(r.key);
    };
    if (nv == 0) return count_later ? make_room() : FZ_OK;
    int ibits = 0, gbits = 0, pbits = 0;                       // index range, block number, record position
    while (ibits < FZ_IDX_BITS && ((imax - imin) >> ibits)) ++ibits;
    while ((gmax >> gbits)) ++gbits;
    while (((cnt - 1) >> pbits)) ++pbits;
    const int kbits = ibits + gbits;
    if (kbits + pbits <= 64) {
        // One 64-bit word per record: (block, index - imin) above the record's position.  ONE distribution pass on the
        // top digit of the key (about two buckets per record), then the buckets — a record or two each unless the
        // matches cluster — are put in order: sorting the whole words orders equal keys by position, and a search
        // never has two records with one key.  (2409 records: 14 -> ~7 us against three LSD passes with their own
        // range pass and two freshly allocated word arrays.)
        static thread_local std::vector<uint64_t> wa, wb;
        static thread_local std::vector<uint32_t> hist;
        if (wa.size() < nv) { wa.resize(nv + nv / 2 + 64); wb.resize(wa.size()); }
        int dbits = 1;                                         // about two records per bucket: the prefix sum and the bucket walk cost per bucket
        while (dbits < 16 && (2ull << dbits) < nv) ++dbits;
        dbits = std::max(1, std::min(dbits, std::max(1, kbits)));
        const int shift = pbits + kbits - dbits;
        const size_t nb = (size_t)1 << dbits;
        hist.assign(nb + 1, 0);
        uint64_t *a = wa.data(), *b = wb.data();
        size_t w = 0;
        for (size_t i = 0; i < cnt; ++i) {
            if (recs[i].dist == FZ_REC_NONE) continue;
            const uint64_t key = ((uint64_t)fz_hit_block(recs[i].key) << ibits) | (fz_hit_index(recs[i].key) - imin);
            const uint64_t word = (pbits < 64 ? key << pbits : 0) | (uint64_t)i;
            a[w++] = word;
            ++hist[(shift >= 0 ? (size_t)(word >> shift) : 0) + 1];
        }
        if (count_later) { nv = w; int rc = make_room(); if (rc) return rc; if (nv == 0) return FZ_OK; }
        for (size_t d = 0; d < nb; ++d) hist[d + 1] += hist[d];
        for (size_t i = 0; i < nv; ++i) b[hist[shift >= 0 ? (size_t)(a[i] >> shift) : 0]++] = a[i];
        // hist[d] is now the END of bucket d
        size_t s0 = 0;
        for (size_t d = 0; d < nb; ++d) {
            const size_t e0 = hist[d];
            const size_t len = e0 - s0;
            if (len > 24) std::sort(b + s0, b + e0);
            else if (len > 1) {
                for (size_t i = s0 + 1; i < e0; ++i) {
                    const uint64_t v = b[i];
                    size_t j = i;
                    while (j > s0 && b[j - 1] > v) { b[j] = b[j - 1]; --j; }
                    b[j] = v;
                }
            }
            s0 = e0;
        }
        const uint64_t pmask = pbits ? ((1ull << pbits) - 1) : 0;
        for (size_t i = 0; i < nv; ++i) put(i, recs[b[i] & pmask]);
        return FZ_OK;
    }
    std::vector<FzRec> copy;
    copy.reserve(nv);
    for (size_t i = 0; i < cnt; ++i) if (recs[i].dist != FZ_REC_NONE) copy.push_back(recs[i]);
    if (count_later) { nv = copy.size(); int rc = make_room(); if (rc) return rc; }
    sort_recs(copy);
    for (size_t i = 0; i < nv; ++i) put(i, copy[i]);
    return FZ_OK;
}

// Segmented searches (file API): reference order = segment (chunk) major, then (block, index).
int emit_matches_seg(const FzRec *recs, size_t cnt, uint32_t L, fz_match **out, uint64_t *n, uint32_t **seg_out) {
    std::vector<FzRec> kept;
    if (drop_empty_slots(recs, cnt, kept)) return emit_matches_seg(kept.data(), kept.size(), L, out, n, seg_out);
    void *mem = nullptr, *smem_ = nullptr;
    int rc = alloc_out(cnt, sizeof(fz_match), &mem);
    if (rc) return rc;
    rc = alloc_out(cnt, sizeof(uint32_t), &smem_);
    if (rc) { release_out(mem); return rc; }
    fz_match *mo = static_cast<fz_match *>(mem);
    uint32_t *so = static_cast<uint32_t *>(smem_);
    std::vector<uint32_t> order(cnt);
    for (size_t i = 0; i < cnt; ++i) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [recs](uint32_t x, uint32_t y) {
        if (recs[x].aux != recs[y].aux) return recs[x].aux < recs[y].aux;
        return recs[x].key < recs[y].key;
    });
    for (size_t i = 0; i < cnt; ++i) {
        const FzRec &r = recs[order[i]];
        const uint64_t idx = fz_hit_index(r.key);
        mo[i].start = (int64_t)(idx - r.l);
        mo[i].end = (int64_t)(idx + L + r.r);
        mo[i].dist = (int32_t)r.dist;
        mo[i].block = (int32_t)fz_hit_block(r.key);
        so[i] = r.aux;
    }
    *out = mo; *n = cnt; *seg_out = so;
    return FZ_OK;
}

// A sharded search (several devices, or the ranks of an all-gather): every shard's records are ordered on their own —
// 10^4 records stay in the host's L2, one 8 x 10^4-record ordering does not (0.8 ms on 77 000 records) — and, shards
// owning ascending index ranges (seg_order lists them that way), the reference's order is, block by block, the shards'
// runs of that block one after the other.  merge_rows does the second half on rows that are already ordered.
int merge_rows(const std::vector<const std::vector<fz_match> *> &rows, const std::vector<uint32_t> &seg_order, fz_match **out, uint64_t *n) {
    const size_t ns = rows.size();
    size_t total = 0;
    for (size_t si = 0; si < ns; ++si) total += rows[si]->size();
    void *mem = nullptr;
    int rc = alloc_out(total, sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *mo = static_cast<fz_match *>(mem);
    std::vector<size_t> pos(ns, 0);
    size_t o = 0;
    while (o < total) {
        int32_t g = INT32_MAX;                             // the smallest block any shard still holds
        for (size_t k = 0; k < ns; ++k) {
            const size_t si = seg_order[k];
            if (pos[si] < rows[si]->size()) g = std::min(g, (*rows[si])[pos[si]].block);
        }
        for (size_t k = 0; k < ns; ++k) {
            const size_t si = seg_order[k];
            const std::vector<fz_match> &r = *rows[si];
            size_t q = pos[si];
            // (a run of one block: gallop, then binary search — a shard holds thousands of rows per block)
            if (q < r.size() && r[q].block == g) {
                size_t step = 1, hi = q + 1;
                while (hi < r.size() && r[hi].block == g) { q = hi; hi = std::min(r.size(), hi + step); step <<= 1; }
                // r[q].block == g, and (hi == r.size() or r[hi].block != g): the run ends in (q, hi]
                size_t lo2 = q + 1, hi2 = hi;
                while (lo2 < hi2) { const size_t mid = (lo2 + hi2) / 2; if (r[mid].block == g) lo2 = mid + 1; else hi2 = mid; }
                q = lo2;
            }
            if (q > pos[si]) memcpy(mo + o, r.data() + pos[si], (q - pos[si]) * sizeof(fz_match));
            o += q - pos[si];
            pos[si] = q;
        }
    }
    *out = mo;
    *n = total;
    return FZ_OK;
}

// seg_ends[i] = end of shard i's records in `recs`.  With workers (ctx) the segments are ordered in parallel.
int emit_matches_segments(fz_ctx *ctx, const FzRec *recs, const std::vector<size_t> &seg_ends, const std::vector<uint32_t> &seg_order, uint32_t L,
                          fz_match **out, uint64_t *n, uint64_t idx_bound, uint32_t blk_bound, bool may_have_empty) {
    const size_t ns = seg_ends.size();
    static thread_local std::vector<std::vector<fz_match>> rows;
    if (rows.size() < ns) rows.resize(ns);
    std::vector<fz_match> *rowp = rows.data();
    auto order_one = [recs, &seg_ends, L, idx_bound, blk_bound, may_have_empty, rowp](size_t si) {
        const size_t b = si ? seg_ends[si - 1] : 0, e = seg_ends[si];
        return emit_matches(recs + b, e - b, L, nullptr, nullptr, idx_bound, blk_bound, may_have_empty, &rowp[si]);
    };
    if (ctx && ctx->workers && ns > 1 && ns <= ctx->workers->ws.size() && seg_ends.back() >= 4096) {
        for (size_t si = 0; si < ns; ++si) ctx->workers->post(si, [order_one, si]() { return order_one(si); });
        int rc = FZ_OK;
        for (size_t si = 0; si < ns; ++si) { const int r = ctx->workers->wait(si); if (r && !rc) rc = r; }
        if (rc) return rc;
    } else {
        for (size_t si = 0; si < ns; ++si) { int rc = order_one(si); if (rc) return rc; }
    }
    std::vector<const std::vector<fz_match> *> ptrs(ns);
    for (size_t si = 0; si < ns; ++si) ptrs[si] = &rows[si];
    return merge_rows(ptrs, seg_order, out, n);
}

// the records of the search that just ran: the staging-buffer view, the shards' ordered rows or the collected vector
int emit_matches(fz_ctx *ctx, const std::vector<FzRec> &recs, uint32_t L, fz_match **out, uint64_t *n,
                 uint64_t idx_bound = 0, uint32_t blk_bound = 0) {
    if (ctx->rows_ready) {                                 // several shards: ordered by search_collect, merged here
        std::vector<const std::vector<fz_match> *> ptrs(ctx->seg_order.size());
        for (size_t si = 0; si < ptrs.size(); ++si) ptrs[si] = &ctx->souts[si].rows;
        return merge_rows(ptrs, ctx->seg_order, out, n);
    }
    // (the fused scan appends real records only; the stand-alone verifications may leave empty slots)
    bool dense = true;                                     // (of the search being collected: its slot is the current one)
    for (const DevState &d : ctx->devs) dense = dense && d.fused_used;
    if (!ctx->view && ctx->seg_ends.size() > 1 && ctx->seg_ends.back() == recs.size())
        return emit_matches_segments(ctx, recs.data(), ctx->seg_ends, ctx->seg_order, L, out, n, idx_bound, blk_bound, !dense);
    return ctx->view ? emit_matches(ctx->view, (size_t)ctx->view_n, L, out, n, idx_bound, blk_bound, !dense)
                     : emit_matches(recs.data(), recs.size(), L, out, n, idx_bound, blk_bound, !dense);
}

}  // namespace

extern "C" {

int fz_abi_version(void) { return FZ_ABI_VERSION; }

const char *fz_last_error(void) { return g_err.c_str(); }

int fz_device_count(int *n) {
    if (!n) return fail(FZ_EINVAL, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(FZ_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return FZ_OK;
}

static int devstate_init(DevState &d) {
    HIP_TRY(hipSetDevice(d.device));
    // a search call is ~0.3 ms: spin instead of sleeping on the completion interrupt — on EVERY device of the context
    // (the flag is per device; refused with hipErrorSetOnActiveProcess once a device is in use: then it stays as it is)
    if (!sw().no_spin) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void)hipGetLastError();
    // Lowest priority: a different hardware queue than the default-priority streams of the rest of
    // the process, and the dispatcher prefers their workgroups.  Measured with RCCL on torch's
    // stream: at default priority an all_gather launched while a scan was running only started
    // after it (250 us); the scan alone is not slower at low priority.
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const bool default_prio = sw().stream_default_priority;
    HIP_TRY(hipStreamCreateWithPriority(&d.stream, hipStreamNonBlocking, default_prio ? 0 : prio_least));
    HIP_TRY(hipStreamCreateWithPriority(&d.stream_alt, hipStreamNonBlocking, default_prio ? 0 : prio_least));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.d_hdr_alt), kHeaderBytes));
    HIP_TRY(hipMemset(d.d_hdr_alt, 0, kHeaderBytes));
    for (auto &ev : d.ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&d.h_stage), kHeaderBytes + kHostRecs * sizeof(FzRec), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&d.h_stage_dev), d.h_stage, 0));
    for (auto &ev : d.other.ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&d.other.h_stage), kHeaderBytes + kHostRecs * sizeof(FzRec), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&d.other.h_stage_dev), d.other.h_stage, 0));
    int rc = ensure_hits(d, 1u << 20);
    if (rc == FZ_OK) rc = ensure_recs(d, 1u << 16);
    return rc;
}

static void devstate_destroy(DevState &d) {
    (void)hipSetDevice(d.device);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    if (d.stream_alt) { (void)hipStreamSynchronize(d.stream_alt); (void)hipStreamDestroy(d.stream_alt); }
    if (d.d_hdr_alt) (void)hipFree(d.d_hdr_alt);
    if (d.stream_hi) (void)hipStreamSynchronize(d.stream_hi);
    if (d.d_hits) (void)hipFree(d.d_hits);
    if (d.d_out) (void)hipFree(d.d_out);
    if (d.spare_alloc) (void)hipFree(d.spare_alloc);
    if (d.h_stage) (void)hipHostFree(d.h_stage);
    if (d.other.h_stage) (void)hipHostFree(d.other.h_stage);
    for (auto &ev : d.other.ev) if (ev) (void)hipEventDestroy(ev);
    if (d.h_big) (void)hipHostFree(d.h_big);
    if (d.d_cand) (void)hipFree(d.d_cand);
    if (d.d_pat) (void)hipFree(d.d_pat);
    if (d.d_gen_order) (void)hipFree(d.d_gen_order);
    if (d.d_gen_rows) (void)hipFree(d.d_gen_rows);
    if (d.d_gen_dedup) (void)hipFree(d.d_gen_dedup);
    for (int i = 0; i < 2; ++i) if (d.stream_h[i]) (void)hipHostFree(d.stream_h[i]);
    if (d.stream_d) (void)hipFree(d.stream_d);
    for (auto &ev : d.ev) if (ev) (void)hipEventDestroy(ev);
    if (d.ev_scan_done) (void)hipEventDestroy(d.ev_scan_done);
    if (d.stream_hi) { (void)hipStreamSynchronize(d.stream_hi); (void)hipStreamDestroy(d.stream_hi); }
    if (d.stream) (void)hipStreamDestroy(d.stream);
}

// the second lane (generic searches in flight): created on first use, hit lists as large as lane 0's
static int ensure_lane2(fz_ctx *ctx) {
    if (ctx->devs2.size() != ctx->devs.size()) {
        // built aside and only installed once EVERY device initialised: a half-built lane (out of memory part-way) must
        // not be mistaken for a usable one by the next call
        std::vector<DevState> lane(ctx->devs.size());
        for (size_t i = 0; i < ctx->devs.size(); ++i) {
            lane[i].device = ctx->devs[i].device;
            lane[i].n_cus = ctx->devs[i].n_cus;
            int rc = devstate_init(lane[i]);
            if (rc) {
                const std::string keep = g_err;
                for (size_t j = 0; j <= i; ++j) devstate_destroy(lane[j]);
                g_err = keep;
                return rc;
            }
        }
        ctx->devs2.swap(lane);
    }
    for (size_t i = 0; i < ctx->devs.size(); ++i) {
        int rc = ensure_hits(ctx->devs2[i], ctx->devs[i].hit_cap);
        if (rc) return rc;
    }
    return FZ_OK;
}

int fz_create(const int *device_ids, int n_devices, fz_ctx **out) {
    if (!out) return fail(FZ_EINVAL, "null argument");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(FZ_EDEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    // a search call is ~0.3 ms: spin instead of sleeping on the completion interrupt (ignored if the
    // runtime was already initialised with other flags, e.g. by torch in a distributed job)
    if (!sw().no_spin) (void)hipSetDeviceFlags(hipDeviceScheduleSpin);
    (void)hipGetLastError();
    std::vector<int> ids;
    if (!device_ids || n_devices <= 0) ids.push_back(0);
    else ids.assign(device_ids, device_ids + n_devices);
    fz_ctx *ctx = new (std::nothrow) fz_ctx();
    if (!ctx) return fail(FZ_ENOMEM, "out of memory");
    for (int id : ids) {
        if (id < 0 || id >= count) { fz_destroy(ctx); return fail(FZ_EINVAL, "device id %d out of range (have %d)", id, count); }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, id) != hipSuccess) { fz_destroy(ctx); return fail(FZ_EDEVICE, "hipGetDeviceProperties failed"); }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fz_destroy(ctx);
            return fail(FZ_EDEVICE, "device %d is %s; libfzhip is built for gfx950 (MI355X) only", id, prop.gcnArchName);
        }
        ctx->devs.emplace_back();
        DevState &d = ctx->devs.back();
        d.device = id;
        d.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        int rc = devstate_init(d);
        if (rc) { fz_destroy(ctx); return rc; }
    }
    if (ctx->devs.size() > 1 && !sw().no_dev_threads) {
        ctx->workers = new (std::nothrow) DevWorkers(ctx->devs.size());
        if (!ctx->workers) { fz_destroy(ctx); return fail(FZ_ENOMEM, "out of memory"); }
    }
    *out = ctx;
    return FZ_OK;
}

void fz_destroy(fz_ctx *ctx) {
    if (!ctx) return;
    delete ctx->workers;
    ctx->workers = nullptr;
    // sequences the caller never released: their device memory goes with the context (their handles die with it)
    while (!ctx->live.empty()) {
        for (DevState &d : ctx->devs) { (void)hipSetDevice(d.device); if (d.stream) (void)hipStreamSynchronize(d.stream); }
        fz_seq_release(ctx->live.back());
    }
    comm_teardown(ctx);
    for (DevState &d : ctx->devs) devstate_destroy(d);
    for (DevState &d : ctx->devs2) devstate_destroy(d);
    delete ctx;
}

static int upload_one(fz_ctx *ctx, int dev_index, const uint8_t *host_buf, const FzGeom &geom, Shard &sh) {
    DevState &d = ctx->devs[dev_index];
    HIP_TRY(hipSetDevice(d.device));
    const uint64_t tiles = (geom.buf_len + FZ_TILE_BYTES - 1) / FZ_TILE_BYTES;
    const uint64_t body = std::max<uint64_t>(1, tiles) * FZ_TILE_BYTES;
    sh.dev = dev_index;
    sh.alloc_bytes = FZ_PAD_FRONT + body + FZ_PAD_BACK;
    sh.geom = geom;
    if (d.spare_alloc && d.spare_bytes >= sh.alloc_bytes && d.spare_bytes <= 4 * sh.alloc_bytes + (1u << 20)) {
        sh.d_alloc = d.spare_alloc;               // recycle (the stream orders the reuse)
        sh.alloc_bytes = d.spare_bytes;
        d.spare_alloc = nullptr;
        d.spare_bytes = 0;
    } else {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sh.d_alloc), sh.alloc_bytes));
    }
    sh.d_buf = sh.d_alloc + FZ_PAD_FRONT;
    HIP_TRY(hipMemsetAsync(sh.d_alloc, 0, FZ_PAD_FRONT, d.stream));
    HIP_TRY(hipMemsetAsync(sh.d_buf + geom.buf_len, 0, sh.alloc_bytes - FZ_PAD_FRONT - geom.buf_len, d.stream));
    if (geom.buf_len) HIP_TRY(hipMemcpyAsync(sh.d_buf, host_buf, geom.buf_len, hipMemcpyHostToDevice, d.stream));
    // grow the hit list with the sequence: n-gram hits on 4-letter text run at ~G * n / 4^L
    int rc = ensure_hits(d, std::max<uint64_t>(1u << 20, geom.buf_len / 64));
    if (rc) return rc;
    return FZ_OK;
}

int fz_seq_upload(fz_ctx *ctx, const uint8_t *host, uint64_t n, fz_seq **out) {
    if (!ctx || !out || (!host && n)) return fail(FZ_EINVAL, "null argument");
    if (n >= (1ull << FZ_IDX_BITS)) return fail(FZ_EUNSUPPORTED, "sequence too long");
    *out = nullptr;
    fz_seq *seq = new (std::nothrow) fz_seq();
    if (!seq) return fail(FZ_ENOMEM, "out of memory");
    seq->ctx = ctx;
    seq->n = n;
    const uint64_t R = ctx->devs.size();
    const uint64_t halo = 2 * (uint64_t)FZ_MAX_M_ANY + 2 * FZ_MAX_K_ANY + 64;   // >= m + k for every supported query
    for (uint64_t r = 0; r < R; ++r) {
        FzGeom g{};
        g.n = n;
        g.own_lo = n / R * r + std::min<uint64_t>(r, n % R);
        g.own_hi = n / R * (r + 1) + std::min<uint64_t>(r + 1, n % R);
        const uint64_t lo = g.own_lo > halo ? g.own_lo - halo : 0;
        const uint64_t hi = std::min<uint64_t>(n, g.own_hi + halo);
        g.buf_off = lo;
        g.buf_len = hi - lo;
        if (R > 1 && g.own_hi == g.own_lo) continue;   // nothing to own
        seq->shards.emplace_back();
        int rc = upload_one(ctx, (int)r, host + lo, g, seq->shards.back());
        if (rc) { fz_seq_release(seq); return rc; }
    }
    for (Shard &sh : seq->shards) {
        (void)hipSetDevice(ctx->devs[sh.dev].device);
        hipError_t e = hipStreamSynchronize(ctx->devs[sh.dev].stream);
        if (e != hipSuccess) { fz_seq_release(seq); return fail(FZ_EDEVICE, "upload failed: %s", hipGetErrorString(e)); }
    }
    ctx->live.push_back(seq);
    *out = seq;
    return FZ_OK;
}

int fz_seq_upload_shard(fz_ctx *ctx, const uint8_t *host_buf, uint64_t buf_len, uint64_t buf_global_off,
                        uint64_t own_lo, uint64_t own_hi, uint64_t global_n, fz_seq **out) {
    if (!ctx || !out || (!host_buf && buf_len)) return fail(FZ_EINVAL, "null argument");
    if (global_n >= (1ull << FZ_IDX_BITS)) return fail(FZ_EUNSUPPORTED, "sequence too long");
    if (buf_global_off + buf_len > global_n || own_lo > own_hi || own_hi > global_n)
        return fail(FZ_EINVAL, "shard ranges outside the global sequence");
    if (own_hi > own_lo && (own_lo < buf_global_off || own_hi > buf_global_off + buf_len))
        return fail(FZ_EINVAL, "owned range not inside the shard buffer");
    *out = nullptr;
    fz_seq *seq = new (std::nothrow) fz_seq();
    if (!seq) return fail(FZ_ENOMEM, "out of memory");
    seq->ctx = ctx;
    seq->n = global_n;
    FzGeom g{};
    g.n = global_n;
    g.buf_off = buf_global_off;
    g.buf_len = buf_len;
    g.own_lo = own_lo;
    g.own_hi = own_hi;
    seq->shards.emplace_back();
    int rc = upload_one(ctx, 0, host_buf, g, seq->shards.back());
    if (rc == FZ_OK) {
        hipError_t e = hipStreamSynchronize(ctx->devs[0].stream);
        if (e != hipSuccess) rc = fail(FZ_EDEVICE, "upload failed: %s", hipGetErrorString(e));
    }
    if (rc) { fz_seq_release(seq); return rc; }
    ctx->live.push_back(seq);
    *out = seq;
    return FZ_OK;
}

int fz_seq_new(fz_ctx *ctx, uint64_t global_n, fz_seq **out) {
    if (!ctx || !out) return fail(FZ_EINVAL, "null argument");
    *out = nullptr;
    if (global_n >= (1ull << FZ_IDX_BITS)) return fail(FZ_EUNSUPPORTED, "sequence too long");
    fz_seq *seq = new (std::nothrow) fz_seq();
    if (!seq) return fail(FZ_ENOMEM, "out of memory");
    seq->ctx = ctx;
    seq->n = global_n;
    ctx->live.push_back(seq);
    *out = seq;
    return FZ_OK;
}

int fz_seq_add_shard(fz_seq *seq, int dev_index, const uint8_t *host_buf, uint64_t buf_len, uint64_t buf_global_off,
                     uint64_t own_lo, uint64_t own_hi) {
    if (!seq || !seq->ctx || (!host_buf && buf_len)) return fail(FZ_EINVAL, "null argument");
    fz_ctx *ctx = seq->ctx;
    if (ctx->npend || ctx->stream_inflight) return fail(FZ_EINVAL, "a search of this context is in flight");
    if (dev_index < 0 || dev_index >= (int)ctx->devs.size()) return fail(FZ_EINVAL, "device index %d outside the context (%zu devices)", dev_index, ctx->devs.size());
    if (buf_global_off + buf_len > seq->n || own_lo > own_hi || own_hi > seq->n)
        return fail(FZ_EINVAL, "shard ranges outside the global sequence");
    if (own_hi > own_lo && (own_lo < buf_global_off || own_hi > buf_global_off + buf_len))
        return fail(FZ_EINVAL, "owned range not inside the shard buffer");
    for (const Shard &o : seq->shards) {
        if (o.dev == dev_index) return fail(FZ_EINVAL, "device %d already holds a shard of this sequence", dev_index);
        if (own_lo < o.geom.own_hi && o.geom.own_lo < own_hi) return fail(FZ_EINVAL, "owned ranges of two shards overlap");
    }
    FzGeom g{};
    g.n = seq->n;
    g.buf_off = buf_global_off;
    g.buf_len = buf_len;
    g.own_lo = own_lo;
    g.own_hi = own_hi;
    seq->shards.emplace_back();
    int rc = upload_one(ctx, dev_index, host_buf, g, seq->shards.back());
    if (rc == FZ_OK) {
        hipError_t e = hipStreamSynchronize(ctx->devs[dev_index].stream);     // host_buf is only borrowed for this call
        if (e != hipSuccess) rc = fail(FZ_EDEVICE, "upload failed: %s", hipGetErrorString(e));
    }
    if (rc) {
        Shard &sh = seq->shards.back();
        if (sh.d_alloc) { (void)hipSetDevice(ctx->devs[dev_index].device); (void)hipFree(sh.d_alloc); }
        seq->shards.pop_back();
    }
    return rc;
}

int fz_device_ms(fz_ctx *ctx, double *filter_ms, int cap) {
    if (!ctx || (!filter_ms && cap > 0)) return fail(FZ_EINVAL, "null argument");
    resolve_timing(ctx);
    for (int i = 0; i < cap && i < (int)ctx->devs.size(); ++i) filter_ms[i] = ctx->devs[i].last_filter_ms;
    return (int)ctx->devs.size();
}

uint64_t fz_seq_len(const fz_seq *seq) { return seq ? seq->n : 0; }

void fz_seq_release(fz_seq *seq) {
    if (!seq) return;
    if (seq->ctx) {
        auto &live = seq->ctx->live;
        live.erase(std::remove(live.begin(), live.end(), seq), live.end());
    }
    for (Shard &sh : seq->shards) {
        if (sh.d_alloc) {
            DevState &d = seq->ctx->devs[sh.dev];
            (void)hipSetDevice(d.device);
            if (sh.alloc_bytes <= (64u << 20) && (!d.spare_alloc || d.spare_bytes < sh.alloc_bytes)) {
                if (d.spare_alloc) (void)hipFree(d.spare_alloc);
                d.spare_alloc = sh.d_alloc;       // keep small buffers for the next upload
                d.spare_bytes = sh.alloc_bytes;
            } else {
                (void)hipFree(sh.d_alloc);
            }
        }
    }
    delete seq;
}

int fz_search_exact(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint64_t lo, uint64_t hi,
                    uint64_t **idx, uint64_t *n) {
    if (!idx || !n) return fail(FZ_EINVAL, "null argument");
    *idx = nullptr; *n = 0;
    int rc = validate(ctx, seq, p, m);
    if (rc) return rc;
    rc = check_halo(seq, m);
    if (rc) return rc;
    const uint64_t N = seq->n;
    lo = std::min(lo, N);                                  // search_exact.py:70-71
    hi = std::max(lo, std::min(hi, N));
    Search q;
    q.mode = FZ_MODE_EXACT; q.m = m; q.k = 0; q.p = p;
    q.plan.L = m;
    q.plan.s = {0};
    q.plan.abs_lo = lo;
    q.plan.abs_hi = hi;
    std::vector<FzRec> recs;
    std::vector<uint64_t> hits;
    rc = run_search(ctx, seq, q, /*with_verify=*/false, recs, hits);
    if (rc) return rc;
    if (comm_multi_process(ctx)) {                         // one rank of several: every rank gets every rank's hits
        std::vector<uint8_t> all;
        rc = comm_gather_host(ctx, hits.data(), hits.size() * sizeof(uint64_t), all);
        if (rc) return rc;
        hits.resize(all.size() / sizeof(uint64_t));
        if (!all.empty()) memcpy(hits.data(), all.data(), all.size());
    }
    std::sort(hits.begin(), hits.end());
    void *mem = nullptr;
    rc = alloc_out(hits.size(), sizeof(uint64_t), &mem);
    if (rc) return rc;
    uint64_t *o = static_cast<uint64_t *>(mem);
    for (size_t i = 0; i < hits.size(); ++i) o[i] = fz_hit_index(hits[i]);
    *idx = o; *n = hits.size();
    ctx->stats.raw_matches = hits.size();
    return FZ_OK;
}

// Argument checks and the block plan of find_near_matches_levenshtein_ngrams (levenshtein_ngram.py:159-176).
static int lev_plan(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, Search &q, bool in_pipeline = false) {
    int rc = validate(ctx, seq, p, m, in_pipeline);
    if (rc) return rc;
    const uint32_t L = m / (k + 1);
    if (L == 0) return fail(FZ_EINVAL, "the subsequence length must be greater than max_l_dist");
    if (k > FZ_MAX_K_ANY) return fail(FZ_EUNSUPPORTED, "max_l_dist above %u is not supported by the verify kernels", FZ_MAX_K_ANY);
    rc = check_halo(seq, (uint64_t)m + k);
    if (rc) return rc;
    q.mode = FZ_MODE_LEV; q.m = m; q.k = k; q.p = p;
    q.collective = ctx->snapshot;
    q.plan.L = L;
    for (uint32_t s = 0; s + L <= m; s += L) q.plan.s.push_back(s);   // levenshtein_ngram.py:171-176 (ranges: fz_block_range)
    if (q.plan.s.size() > FZ_MAX_BLOCKS) return fail(FZ_EUNSUPPORTED, "more than %u n-gram blocks", FZ_MAX_BLOCKS);
    return FZ_OK;
}

int fz_lev_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    Search q;
    int rc = lev_plan(ctx, seq, p, m, k, q);
    if (rc) return rc;
    std::vector<FzRec> recs;
    std::vector<uint64_t> hits;
    Trace tr;
    rc = run_search(ctx, seq, q, true, recs, hits);
    if (rc) return rc;
    tr.mark("run_search");
    rc = emit_matches(ctx, recs, q.plan.L, out, n, seq->n, (uint32_t)q.plan.s.size());
    tr.mark("sort+emit");
    if (rc == FZ_OK) ctx->stats.raw_matches = *n;
    return rc;
}

// find_near_matches_levenshtein_ngrams + consolidate_overlapping_matches in one call (what LevenshteinSearch.search followed
// by consolidate_matches computes): no second call, no array round trip through the caller in between.
int fz_lev_ngrams_consolidated(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    fz_match *raw = nullptr;
    uint64_t nraw = 0;
    int rc = fz_lev_ngrams(ctx, seq, p, m, k, &raw, &nraw);
    if (rc) return rc;
    rc = fz_consolidate(raw, nraw, out, n);
    release_out(raw);
    return rc;
}

// Argument checks and the block plan of the substitutions-only n-gram search (template :92-101).
static int subs_plan(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, Search &q, bool in_pipeline = false) {
    int rc = validate(ctx, seq, p, m, in_pipeline);
    if (rc) return rc;
    const uint32_t L = m / (k + 1);
    if (L == 0) return fail(FZ_EUNSUPPORTED, "max_substitutions >= len(subsequence): every window matches; not a GPU path");
    rc = check_halo(seq, m);
    if (rc) return rc;
    q.mode = FZ_MODE_SUBS; q.m = m; q.k = k; q.p = p;
    q.collective = ctx->snapshot;
    q.plan.L = L;
    for (uint32_t s = 0; s + L <= m; s += L) q.plan.s.push_back(s);   // template :92-101 (ranges: fz_block_range)
    if (q.plan.s.size() > FZ_MAX_BLOCKS) return fail(FZ_EUNSUPPORTED, "more than %u n-gram blocks", FZ_MAX_BLOCKS);
    return FZ_OK;
}

// ... and of find_near_matches_generic_ngrams (generic_search.py:198-228).
static int generic_plan(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins, uint32_t max_dels,
                        uint32_t max_l, Search &q, bool in_pipeline = false) {
    int rc = validate(ctx, seq, p, m, in_pipeline);
    if (rc) return rc;
    const uint32_t k = max_l;
    if (k > FZ_MAX_K) return fail(FZ_EUNSUPPORTED, "max_l_dist above %d is not supported by the verify kernels", FZ_MAX_K);
    const uint32_t L = m / (k + 1);
    if (L == 0) return fail(FZ_EINVAL, "the subsequence length must be greater than max_l_dist");
    // the automaton's records carry window-relative start / end in 16 bits each: the window (m + 2k bytes) must fit
    if ((uint64_t)m + 2ull * k > 65535ull)
        return fail(FZ_EUNSUPPORTED, "generic search: len(subsequence) + 2 * max_l_dist = %llu exceeds 65535 (16-bit window coordinates)",
                    (unsigned long long)m + 2ull * k);
    rc = check_halo(seq, (uint64_t)m + k);
    if (rc) return rc;
    q.mode = FZ_MODE_GENERIC; q.m = m; q.k = k; q.p = p;
    q.max_subs = std::min(max_subs, 255u); q.max_ins = std::min(max_ins, 255u); q.max_dels = std::min(max_dels, 255u);
    q.plan.L = L;
    for (uint32_t s = 0; s + L <= m; s += L) q.plan.s.push_back(s);   // generic_search.py:221-228 (ranges: fz_block_range)
    if (q.plan.s.size() > FZ_MAX_BLOCKS) return fail(FZ_EUNSUPPORTED, "more than %u n-gram blocks", FZ_MAX_BLOCKS);
    return FZ_OK;
}

static int emit_generic_result(fz_ctx *ctx, fz_seq *seq, const Search &q, const std::vector<FzGenRec> &recs_vec, bool consolidated,
                               fz_match **out, uint64_t *n);

// Exchange step of a generic search in a one-process-per-GPU job: the automaton records (or folded pairs) of every rank,
// back to back; emit_generic / consolidate_hulls order them by their (block, index) keys, which are global.
static int generic_exchange(fz_ctx *ctx, std::vector<FzGenRec> &recs_vec) {
    if (!comm_multi_process(ctx)) return FZ_OK;
    const FzGenRec *mine = ctx->gen_view ? ctx->gen_view : recs_vec.data();
    const size_t nmine = ctx->gen_view ? (size_t)ctx->gen_view_n : recs_vec.size();
    std::vector<uint8_t> all;
    int rc = comm_gather_host(ctx, mine, nmine * sizeof(FzGenRec), all);
    if (rc) return rc;
    ctx->gen_view = nullptr;
    ctx->gen_view_n = 0;
    recs_vec.resize(all.size() / sizeof(FzGenRec));
    if (!all.empty()) memcpy(recs_vec.data(), all.data(), all.size());
    return FZ_OK;
}

static int pending_plan(fz_ctx *ctx, fz_ctx::Pending &pd, Search &q) {
    const uint8_t *p = pd.pattern.data();
    if (pd.kind == FZ_MODE_LEV) return lev_plan(ctx, pd.seq, p, pd.m, pd.k, q, true);
    if (pd.kind == FZ_MODE_SUBS) return subs_plan(ctx, pd.seq, p, pd.m, pd.k, q, true);
    int rc = generic_plan(ctx, pd.seq, p, pd.m, pd.max_subs, pd.max_ins, pd.max_dels, pd.k, q, true);
    q.fold = pd.consolidated;
    return rc;
}

// Launch a search and return (fz_lev_ngrams_begin, fz_subs_ngrams_begin, fz_generic_ngrams_begin); fz_search_end collects.
static int pending_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t kind, uint32_t k, uint32_t max_subs,
                         uint32_t max_ins, uint32_t max_dels, bool consolidated) {
    if (ctx && ctx->npend >= 2) return fail(FZ_EINVAL, "two searches are already in flight on this context");
    const bool generic = kind == FZ_MODE_GENERIC;
    if (ctx && ctx->npend == 1 && (ctx->pend[0].kind == FZ_MODE_GENERIC) != generic)
        return fail(FZ_EINVAL, "generic searches and Levenshtein / substitutions-only searches cannot be in flight together");
    if (!ctx || !p || m == 0) return fail(FZ_EINVAL, "bad ctx / empty subsequence");
    fz_ctx::Pending &pd = ctx->pend[ctx->npend];
    pd.pattern.assign(p, p + m);                             // the caller's buffer is only borrowed for this call
    pd.seq = seq;
    pd.m = m;
    pd.k = k;
    pd.kind = kind;
    pd.max_subs = max_subs; pd.max_ins = max_ins; pd.max_dels = max_dels;
    pd.consolidated = consolidated;
    pd.lane = 0;
    Search q;
    int rc = pending_plan(ctx, pd, q);
    if (rc) return rc;
    memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    const bool second = ctx->npend == 1;
    pd.launched = true;
    if (generic) {
        // the lane the older search does not use; its scan then runs next to the older search's automaton kernel
        pd.lane = second ? 1 - ctx->pend[0].lane : 0;
        if (pd.lane == 1) { rc = ensure_lane2(ctx); if (rc) return rc; }
        std::vector<FzGenRec> none;
        ctx->lane = pd.lane;
        rc = run_generic(ctx, seq, q, none, /*phase=*/1);
        ctx->lane = 0;
        if (rc) return rc;
    } else {
        // with a search in flight this one goes to the devices' second result slot — unless a device is out of
        // direct mode (large record sets are fetched from the shared device buffer after the kernel: a second
        // search would overwrite it), then the launch waits for fz_search_end of the older search
        if (second && !ctx->snapshot) for (const DevState &d : ctx->devs) if (!d.direct || env_no_direct()) pd.launched = false;
        if (pd.launched) {
            if (second) for (DevState &d : ctx->devs) d.swap_slot();
            rc = search_enqueue(ctx, seq, q, true);
            if (second) for (DevState &d : ctx->devs) d.swap_slot();
            if (rc) return rc;
        }
    }
    ++ctx->npend;
    return FZ_OK;
}

int fz_lev_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k) {
    return pending_begin(ctx, seq, p, m, FZ_MODE_LEV, k, 0, 0, 0, false);
}

int fz_subs_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k) {
    return pending_begin(ctx, seq, p, m, FZ_MODE_SUBS, k, 0, 0, 0, false);
}

int fz_generic_ngrams_begin(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                            uint32_t max_dels, uint32_t max_l, int consolidated) {
    return pending_begin(ctx, seq, p, m, FZ_MODE_GENERIC, max_l, max_subs, max_ins, max_dels, consolidated != 0);
}

int fz_search_end(fz_ctx *ctx, fz_match **out, uint64_t *n) {
    if (!ctx || !out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    if (!ctx->npend) return fail(FZ_EINVAL, "no search in flight");
    fz_ctx::Pending &pd = ctx->pend[0];
    Search q;
    int rc = pending_plan(ctx, pd, q);
    if (pd.kind == FZ_MODE_GENERIC) {
        std::vector<FzGenRec> recs_vec;
        ctx->lane = pd.lane;
        if (rc == FZ_OK) rc = run_generic(ctx, pd.seq, q, recs_vec, /*phase=*/2);
        if (rc == FZ_OK) rc = generic_exchange(ctx, recs_vec);
        if (rc == FZ_OK) rc = emit_generic_result(ctx, pd.seq, q, recs_vec, pd.consolidated, out, n);
        ctx->lane = 0;
        if (--ctx->npend == 1) std::swap(ctx->pend[0], ctx->pend[1]);
        if (rc != FZ_OK && *out) { release_out(*out); *out = nullptr; *n = 0; }
        return rc;
    }
    std::vector<FzRec> recs;
    std::vector<uint64_t> hits;
    if (rc == FZ_OK) rc = search_collect(ctx, pd.seq, q, true, recs, hits);
    if (rc == FZ_OK) rc = emit_matches(ctx, recs, q.plan.L, out, n, pd.seq->n, (uint32_t)q.plan.s.size());
    // the younger search (if any) becomes the oldest: its slot becomes the current one
    if (--ctx->npend == 1) {
        std::swap(ctx->pend[0], ctx->pend[1]);
        if (ctx->pend[0].launched) {
            for (DevState &d : ctx->devs) d.swap_slot();
        } else {                                             // deferred: launch it now, into the slot that just became free
            Search q2;
            int rc2 = pending_plan(ctx, ctx->pend[0], q2);
            if (rc2 == FZ_OK) rc2 = search_enqueue(ctx, ctx->pend[0].seq, q2, true);
            if (rc2 != FZ_OK) { ctx->npend = 0; if (rc == FZ_OK) rc = rc2; }
            ctx->pend[0].launched = true;
        }
    }
    if (rc != FZ_OK && *out) { release_out(*out); *out = nullptr; *n = 0; }   // an error never hands a buffer over
    return rc;
}

int fz_lev_ngrams_end(fz_ctx *ctx, fz_match **out, uint64_t *n) { return fz_search_end(ctx, out, n); }

// has_near_match_* stops at the first match (substitutions_only.py:218-233, generic_search.py:240-253).  On a long
// single-shard sequence the flag-only searches therefore scan the buffer in growing pieces — 64 MiB, then 4 x as much
// each time — and stop behind the first piece that produced a match: a match in the first MiB of 4 GiB costs one 64 MiB
// search (rounds 4 / 5: one launch whose not-yet-started workgroups skipped their tiles — a quarter of a full scan, bounded
// by the skipped workgroups' finish tickets; the generic form: a full scan + the automaton launch).  Without a match the
// pieces cost three more launches than one scan.  -> the byte ranges, or one whole-buffer range.
static std::vector<std::pair<uint64_t, uint64_t>> any_pieces(fz_ctx *ctx, fz_seq *seq) {
    std::vector<std::pair<uint64_t, uint64_t>> out;
    const uint64_t first = 64ull << 20;
    if (seq->shards.size() != 1 || comm_multi_process(ctx) || seq->shards[0].geom.seg_stride != 0 ||
        seq->shards[0].geom.buf_len < 8 * first) {
        out.emplace_back(0, ~0ull);
        return out;
    }
    const uint64_t len = seq->shards[0].geom.buf_len;
    uint64_t at = 0, size = first;
    while (at < len) {
        const uint64_t end = (len - at <= size + size / 2) ? len : at + size;      // (no sliver at the end)
        out.emplace_back(at, end >= len ? ~0ull : end);
        at = end;
        size *= 4;
    }
    return out;
}

static int subs_ngrams_impl(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n, int *found) {
    Search q;
    int rc = subs_plan(ctx, seq, p, m, k, q);
    if (rc) return rc;
    if (seq->n < m) return FZ_OK;                              // template :66-68
    const uint32_t L = q.plan.L;
    q.any = found != nullptr;
    if (q.any) q.collective = false;                           // a flag, not a stream: exchanged below
    std::vector<FzRec> recs;
    std::vector<uint64_t> hits;
    if (found) {
        bool any = false;
        for (const auto &piece : any_pieces(ctx, seq)) {
            q.part_lo = piece.first; q.part_hi = piece.second;
            rc = run_search(ctx, seq, q, true, recs, hits);
            if (rc) return rc;
            any = (ctx->view ? ctx->view_n : (uint64_t)recs.size()) > 0;
            if (any) break;
        }
        if (comm_multi_process(ctx)) { rc = comm_or(ctx, any); if (rc) return rc; }
        *found = any ? 1 : 0;
        return FZ_OK;
    }
    rc = run_search(ctx, seq, q, true, recs, hits);
    if (rc) return rc;
    return emit_matches(ctx, recs, L, out, n, seq->n, (uint32_t)q.plan.s.size());
}

int fz_subs_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    return subs_ngrams_impl(ctx, seq, p, m, k, out, n, nullptr);
}

// The substitutions-only n-gram search as the reference's wrapper returns it for bytes-like input (substitutions_only.py:
// 258-282): the best match of every overlap group in group-list order — fz_group_best of fz_subs_ngrams in one call.
int fz_subs_ngrams_best(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    fz_match *raw = nullptr;
    uint64_t nraw = 0;
    int rc = subs_ngrams_impl(ctx, seq, p, m, k, &raw, &nraw, nullptr);
    if (rc) return rc;
    rc = fz_group_best(raw, nraw, out, n);
    release_out(raw);
    return rc;
}

int fz_subs_ngrams_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, int *found) {
    if (!found) return fail(FZ_EINVAL, "null argument");
    *found = 0;
    return subs_ngrams_impl(ctx, seq, p, m, k, nullptr, nullptr, found);
}

static int generic_ngrams_impl(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                               uint32_t max_dels, uint32_t max_l, fz_match **out, uint64_t *n, int *found, bool consolidated = false);

int fz_generic_ngrams_consolidated(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                                   uint32_t max_dels, uint32_t max_l, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    return generic_ngrams_impl(ctx, seq, p, m, max_subs, max_ins, max_dels, max_l, out, n, nullptr, true);
}

int fz_generic_ngrams(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                      uint32_t max_dels, uint32_t max_l, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    return generic_ngrams_impl(ctx, seq, p, m, max_subs, max_ins, max_dels, max_l, out, n, nullptr);
}

int fz_generic_ngrams_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                          uint32_t max_dels, uint32_t max_l, int *found) {
    if (!found) return fail(FZ_EINVAL, "null argument");
    *found = 0;
    return generic_ngrams_impl(ctx, seq, p, m, max_subs, max_ins, max_dels, max_l, nullptr, nullptr, found);
}

// What a finished generic search hands over: the raw stream in the reference's order, or — consolidated — the device's
// (hull, best match) pairs, one or a few per n-gram hit, through the second stage of the consolidation.
static int emit_generic_result(fz_ctx *ctx, fz_seq *seq, const Search &q, const std::vector<FzGenRec> &recs_vec, bool consolidated,
                               fz_match **out, uint64_t *n) {
    const uint32_t L = q.plan.L, k = q.k;
    if (!consolidated) return emit_generic(ctx, seq, recs_vec, L, k, out, n, nullptr);
    const FzGenRec *prs = ctx->gen_view ? ctx->gen_view : recs_vec.data();
    const size_t npr = ctx->gen_view ? (size_t)ctx->gen_view_n : recs_vec.size();
    static thread_local std::vector<Hull> hulls;
    hulls.clear();
    hulls.reserve(npr);
    for (size_t i = 0; i < npr; ++i) {
        const FzOutRow best = fz_gen_row(prs[i].key, L, k, 0, prs[i].se, prs[i].dist);
        const FzOutRow hull = fz_gen_row(prs[i].key, L, k, 0, prs[i].win, 0);
        hulls.push_back(Hull{hull.start, hull.end, fz_match{best.start, best.end, best.dist, best.block}});
    }
    ctx->stats.raw_matches = npr;
    return consolidate_hulls(hulls, out, n);
}

static int generic_ngrams_impl(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                               uint32_t max_dels, uint32_t max_l, fz_match **out, uint64_t *n, int *found, bool consolidated) {
    Search q;
    int rc = generic_plan(ctx, seq, p, m, max_subs, max_ins, max_dels, max_l, q);
    if (rc) return rc;
    q.any = found != nullptr;
    q.fold = consolidated;
    std::vector<FzGenRec> recs_vec;
    Trace tr;
    if (found) {
        bool any = false;
        for (const auto &piece : any_pieces(ctx, seq)) {       // (the scan of a piece lists its hits, the automaton runs on them)
            q.part_lo = piece.first; q.part_hi = piece.second;
            rc = run_generic(ctx, seq, q, recs_vec);
            if (rc) return rc;
            any = ctx->any_found;
            if (any) break;
        }
        if (comm_multi_process(ctx)) { rc = comm_or(ctx, any); if (rc) return rc; }
        *found = any ? 1 : 0;
        return FZ_OK;
    }
    rc = run_generic(ctx, seq, q, recs_vec);
    if (rc) return rc;
    tr.mark("generic: kernels");
    rc = generic_exchange(ctx, recs_vec);
    if (rc) return rc;
    rc = emit_generic_result(ctx, seq, q, recs_vec, consolidated, out, n);
    tr.mark("generic: rows");
    return rc;
}

}  // extern "C"

namespace {

// Reference order: (segment,) hits in (block, idx) order, each hit's matches in automaton emission order.
// A wave writes the matches of its hit in contiguous runs (one bulk append per <= 512 matches) that
// are already in emission order, so only the runs are ordered, by (segment, key, first emission number),
// and the 24-byte records are read once: 2.1e5 records order in ~0.4 ms instead of ~2 ms.
int emit_generic(fz_ctx *ctx, fz_seq *seq, const std::vector<FzGenRec> &recs_vec, uint32_t L, uint32_t k, fz_match **out,
                 uint64_t *n, uint32_t **seg_out) {
    if (ctx->gen_rows_dev) {                                   // ordered and finished on the device: one copy
        static_assert(sizeof(FzOutRow) == sizeof(fz_match) && offsetof(FzOutRow, dist) == offsetof(fz_match, dist) &&
                      offsetof(FzOutRow, block) == offsetof(fz_match, block), "FzOutRow is fz_match");
        const uint64_t nrows = ctx->gen_rows_n;
        HIP_TRY(hipSetDevice(ctx->gen_rows_device));
        void *mem = nullptr;
        int rc = alloc_out(nrows, sizeof(fz_match), &mem);
        if (rc) return rc;
        hipError_t e = hipMemcpy(mem, ctx->gen_rows_dev, nrows * sizeof(fz_match), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { release_out(mem); return fail(FZ_EDEVICE, "hipMemcpy of the ordered rows: %s", hipGetErrorString(e)); }
        if (seg_out) *seg_out = nullptr;
        *out = static_cast<fz_match *>(mem);
        *n = nrows;
        ctx->stats.raw_matches = nrows;
        return FZ_OK;
    }
    struct Run { uint64_t key; uint32_t seg; uint32_t seq0; uint32_t len; size_t first; };
    std::vector<Run> runs;
    const FzGeom &geom = seq->shards.empty() ? FzGeom{} : seq->shards[0].geom;
    const bool segs = geom.seg_stride != 0;                    // without segments `win` is the hit's slot, not a segment
    const FzGenRec *recs = ctx->gen_view ? ctx->gen_view : recs_vec.data();
    const size_t nrecs = ctx->gen_view ? (size_t)ctx->gen_view_n : recs_vec.size();
    for (size_t i = 0; i < nrecs;) {
        size_t j = i + 1;
        while (j < nrecs && recs[j].key == recs[i].key && recs[j].win == recs[i].win && recs[j].seq == recs[j - 1].seq + 1) ++j;
        runs.push_back(Run{recs[i].key, segs ? recs[i].win : 0u, recs[i].seq, (uint32_t)(j - i), i});
        i = j;
    }
    std::sort(runs.begin(), runs.end(), [](const Run &x, const Run &y) {
        if (x.seg != y.seg) return x.seg < y.seg;
        return x.key != y.key ? x.key < y.key : x.seq0 < y.seq0;
    });
    void *mem = nullptr;
    int rc = alloc_out(nrecs, sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *mo = static_cast<fz_match *>(mem);
    uint32_t *so = nullptr;
    if (seg_out) {
        void *smem_ = nullptr;
        rc = alloc_out(nrecs, sizeof(uint32_t), &smem_);
        if (rc) { release_out(mem); return rc; }
        so = static_cast<uint32_t *>(smem_);
    }
    size_t o = 0;
    for (const Run &run : runs) {
        uint64_t sa = 0;
        if (geom.seg_stride) {                                 // start of segment run.seg (fz_segment)
            const uint64_t core = geom.seg_org + (uint64_t)run.seg * geom.seg_stride;
            sa = core - geom.seg_org >= geom.seg_pre ? core - geom.seg_pre : geom.seg_org;
        }
        for (size_t i = run.first; i < run.first + run.len; ++i, ++o) {
            const FzOutRow row = fz_gen_row(run.key, L, k, sa, recs[i].se, recs[i].dist);
            mo[o].start = row.start; mo[o].end = row.end; mo[o].dist = row.dist; mo[o].block = row.block;
            if (so) so[o] = run.seg;
        }
    }
    if (seg_out) *seg_out = so;
    *out = mo;
    *n = nrecs;
    ctx->stats.raw_matches = nrecs;
    return FZ_OK;
}

}  // namespace

// ---- RCCL without PyTorch ------------------------------------------------
namespace {

int comm_setup_dev(DevState &d) {
    HIP_TRY(hipSetDevice(d.device));
    if (!d.comm_stream) HIP_TRY(hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking));
    for (auto &ev : d.ev_snap) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (!d.ev_done) HIP_TRY(hipEventCreateWithFlags(&d.ev_done, hipEventDisableTiming));
    return ensure_send(d, d.rec_cap);
}

void comm_teardown(fz_ctx *ctx) {
    if (ctx->comm_broken) {
        // a collective never completed: waiting for its stream or freeing what it may still write would hang as well —
        // the communicator, its streams and buffers are left behind (the process is about to report the failure)
        for (DevState &d : ctx->devs) {
            d.comm = nullptr; d.comm_stream = nullptr; d.d_recv = nullptr; d.h_recv = nullptr; d.recv_bytes = 0; d.send_cap = 0;
            for (auto &b : d.d_send) b = nullptr;
            for (auto &ev : d.ev_snap) ev = nullptr;
            d.ev_done = nullptr; d.comm_rank = -1; d.snap_taken[0] = d.snap_taken[1] = false;
        }
        ctx->comm_world = 0; ctx->snapshot = false; ctx->comm_broken = false;
        return;
    }
    for (DevState &d : ctx->devs) {
        (void)hipSetDevice(d.device);
        if (d.comm_stream) (void)hipStreamSynchronize(d.comm_stream);
        if (d.comm) { if (rccl_api()->ok) (void)rccl_api()->CommDestroy(d.comm); d.comm = nullptr; }
        for (auto &b : d.d_send) if (b) { (void)hipFree(b); b = nullptr; }
        d.send_cap = 0;
        if (d.d_recv) { (void)hipFree(d.d_recv); d.d_recv = nullptr; }
        if (d.h_recv) { (void)hipHostFree(d.h_recv); d.h_recv = nullptr; }
        d.recv_bytes = 0;
        for (auto &ev : d.ev_snap) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        if (d.ev_done) { (void)hipEventDestroy(d.ev_done); d.ev_done = nullptr; }
        if (d.comm_stream) { (void)hipStreamDestroy(d.comm_stream); d.comm_stream = nullptr; }
        d.comm_rank = -1;
        d.snap_taken[0] = d.snap_taken[1] = false;
    }
    ctx->comm_world = 0;
    ctx->snapshot = false;
}

int comm_busy(fz_ctx *ctx) {
    if (!ctx) return fail(FZ_EINVAL, "null argument");
    if (ctx->npend || ctx->stream_inflight) return fail(FZ_EINVAL, "a search of this context is in flight");
    return FZ_OK;
}

// One-process-per-GPU jobs: is this context one rank of several (its own results are only a part of the answer)?
bool comm_multi_process(const fz_ctx *ctx) { return ctx->snapshot && ctx->comm_world > 0 && ctx->devs.size() == 1; }

// nbytes from every rank of a one-process-per-GPU communicator, rank order, through device buffers (blocking).
int comm_allgather_fixed(fz_ctx *ctx, const void *send, uint64_t nbytes, void *recv) {
    if (ctx->comm_broken) return fail(FZ_ETIMEOUT, "the communicator was abandoned after a collective ran into its deadline");
    DevState &d = ctx->devs[0];
    HIP_TRY(hipSetDevice(d.device));
    const uint64_t world = (uint64_t)ctx->comm_world;
    uint8_t *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (world + 1) * nbytes));
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(tmp, send, nbytes, hipMemcpyHostToDevice, d.comm_stream));
        NCCL_TRY(rccl_api()->AllGather(tmp, tmp + nbytes, nbytes, ncclChar, d.comm, d.comm_stream));
        HIP_TRY(hipMemcpyAsync(recv, tmp + nbytes, world * nbytes, hipMemcpyDeviceToHost, d.comm_stream));
        return comm_wait(d.comm_stream, "an all-gather over the job's ranks");
    };
    int rc = body();
    if (rc == FZ_ETIMEOUT) ctx->comm_broken = true;        // (and the buffer stays: the stream may still write to it)
    else (void)hipFree(tmp);
    return rc;
}

// The exchange step of the searches whose results reach the host before they are exchanged (exact search: hit
// indices; generic search: automaton records / folded pairs; the linear-programming routes): every rank contributes
// `bytes` bytes of `elem`-byte items, `all` receives the ranks' contributions back to back in rank order.  Two
// all-gathers (sizes, then the payload padded to the largest contribution).  The callers order what they get by keys
// that are global (hit index, block), so the rank order itself carries no meaning.
int comm_gather_host(fz_ctx *ctx, const void *data, uint64_t bytes, std::vector<uint8_t> &all) {
    const auto t_start = std::chrono::steady_clock::now();
    const uint64_t world = (uint64_t)ctx->comm_world;
    std::vector<uint64_t> sizes(world, 0);
    int rc = comm_allgather_fixed(ctx, &bytes, sizeof bytes, sizes.data());
    if (rc) return rc;
    uint64_t top = 0, total = 0;
    for (uint64_t v : sizes) { top = std::max(top, v); total += v; }
    all.clear();
    if (top == 0) return FZ_OK;
    const uint64_t cap = (top + 255) / 256 * 256;
    std::vector<uint8_t> mine(cap, 0), got(world * cap);
    if (bytes) memcpy(mine.data(), data, bytes);
    rc = comm_allgather_fixed(ctx, mine.data(), cap, got.data());
    if (rc) return rc;
    all.resize(total);
    uint64_t o = 0;
    for (uint64_t r = 0; r < world; ++r) {
        if (sizes[r]) memcpy(all.data() + o, got.data() + r * cap, sizes[r]);
        o += sizes[r];
    }
    ctx->last_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    return FZ_OK;
}

// has_near_match_* of a one-process-per-GPU job: true on every rank if any rank found something
int comm_or(fz_ctx *ctx, bool &flag) {
    uint64_t mine = flag ? 1 : 0;
    std::vector<uint64_t> all((size_t)ctx->comm_world, 0);
    int rc = comm_allgather_fixed(ctx, &mine, sizeof mine, all.data());
    if (rc) return rc;
    for (uint64_t v : all) flag = flag || v != 0;
    return FZ_OK;
}

// seq->rank_lo: the first index every rank owns of this sequence.  A multi-device context knows them; the ranks of a
// one-process-per-GPU job exchange them on the sequence's first collective search (every rank is in that search).
int comm_rank_lows(fz_ctx *ctx, fz_seq *seq) {
    const size_t world = (size_t)ctx->comm_world;
    if (seq->rank_lo.size() == world) return FZ_OK;
    seq->rank_lo.assign(world, ~0ull);
    if (ctx->devs.size() == world && world > 1) {
        for (const Shard &sh : seq->shards)
            if (sh.geom.own_hi > sh.geom.own_lo) seq->rank_lo[(size_t)ctx->devs[sh.dev].comm_rank] = sh.geom.own_lo;
        return FZ_OK;
    }
    uint64_t mine = ~0ull;
    for (const Shard &sh : seq->shards) if (sh.geom.own_hi > sh.geom.own_lo) mine = std::min(mine, sh.geom.own_lo);
    int rc = comm_allgather_fixed(ctx, &mine, sizeof mine, seq->rank_lo.data());
    if (rc) seq->rank_lo.clear();
    return rc;
}

}  // namespace

extern "C" {

int fz_comm_unique_id(void *id, uint64_t id_bytes) {
    static_assert(FZ_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id || id_bytes < FZ_COMM_ID_BYTES) return fail(FZ_EINVAL, "the id buffer must hold %d bytes", FZ_COMM_ID_BYTES);
    RCCL_NEED();
    ncclUniqueId u;
    NCCL_TRY(rccl_api()->GetUniqueId(&u));
    memcpy(id, u.internal, FZ_COMM_ID_BYTES);
    return FZ_OK;
}

int fz_comm_init_rank(fz_ctx *ctx, const void *id, int world, int rank) {
    int rc = comm_busy(ctx);
    if (rc) return rc;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(FZ_EINVAL, "bad communicator arguments");
    if (ctx->devs.size() != 1) return fail(FZ_EINVAL, "fz_comm_init_rank needs a single-device context (one process per GPU); use fz_comm_init_all");
    if (ctx->comm_world) return fail(FZ_EINVAL, "the context already joined a communicator");
    RCCL_NEED();
    DevState &d = ctx->devs[0];
    rc = comm_setup_dev(d);
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, FZ_COMM_ID_BYTES);
    NCCL_TRY(rccl_api()->CommInitRank(&d.comm, world, u, rank));
    d.comm_rank = rank;
    ctx->comm_world = world;
    ctx->snapshot = true;
    return FZ_OK;
}

int fz_comm_init_all(fz_ctx *ctx) {
    int rc = comm_busy(ctx);
    if (rc) return rc;
    if (ctx->comm_world) return fail(FZ_EINVAL, "the context already joined a communicator");
    RCCL_NEED();
    const int nd = (int)ctx->devs.size();
    std::vector<int> ids(nd);
    for (int i = 0; i < nd; ++i) {
        ids[i] = ctx->devs[i].device;
        // RCCL refuses two ranks on one GPU; the test suite's stand-in library (tests/mock_rccl.cpp, loaded through
        // FZ_RCCL_LIB) does not, which is how the N-rank code runs where there is only one device
        for (int j = 0; j < i && !rccl_api()->stand_in; ++j)
            if (ids[j] == ids[i]) return fail(FZ_EUNSUPPORTED, "device %d appears twice in the context: RCCL needs one rank per GPU", ids[i]);
    }
    for (DevState &d : ctx->devs) { rc = comm_setup_dev(d); if (rc) return rc; }
    std::vector<ncclComm_t> comms(nd, nullptr);
    NCCL_TRY(rccl_api()->CommInitAll(comms.data(), nd, ids.data()));
    for (int i = 0; i < nd; ++i) { ctx->devs[i].comm = comms[i]; ctx->devs[i].comm_rank = i; }
    ctx->comm_world = nd;
    ctx->snapshot = true;
    return FZ_OK;
}

int fz_comm_info(fz_ctx *ctx, int *world, int *rank, int *collective) {
    if (!ctx) return fail(FZ_EINVAL, "null argument");
    if (world) *world = ctx->comm_world;
    if (rank) *rank = ctx->comm_world ? ctx->devs[0].comm_rank : -1;
    if (collective) *collective = ctx->snapshot ? 1 : 0;
    return FZ_OK;
}

int fz_comm_set_collective(fz_ctx *ctx, int on) {
    int rc = comm_busy(ctx);
    if (rc) return rc;
    if (!ctx->comm_world) return fail(FZ_EINVAL, "the context has not joined a communicator");
    ctx->snapshot = on != 0;
    return FZ_OK;
}

void fz_comm_destroy(fz_ctx *ctx) {
    if (ctx) comm_teardown(ctx);
}

// host buffers: nbytes from every rank, rank order (load-time exchanges: halos, the job's clock)
int fz_comm_allgather(fz_ctx *ctx, const void *send, uint64_t nbytes, void *recv) {
    int rc = comm_busy(ctx);
    if (rc) return rc;
    if (!ctx->comm_world) return fail(FZ_EINVAL, "the context has not joined a communicator");
    if (ctx->devs.size() != 1) return fail(FZ_EINVAL, "fz_comm_allgather is for one-process-per-GPU jobs (a multi-device context holds every rank's data itself)");
    if (!nbytes) return FZ_OK;
    if (!send || !recv) return fail(FZ_EINVAL, "null argument");
    return comm_allgather_fixed(ctx, send, nbytes, recv);
}

// max over the ranks of one double per rank (the job's step time); doubles as a barrier
int fz_comm_max_f64(fz_ctx *ctx, double *value) {
    int rc = comm_busy(ctx);
    if (rc) return rc;
    if (!ctx->comm_world || !value) return fail(FZ_EINVAL, "no communicator / null argument");
    if (ctx->devs.size() != 1) return FZ_OK;               // one process holds every rank: nothing to reduce
    if (ctx->comm_broken) return fail(FZ_ETIMEOUT, "the communicator was abandoned after a collective ran into its deadline");
    DevState &d = ctx->devs[0];
    HIP_TRY(hipSetDevice(d.device));
    double *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), sizeof(double)));
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(tmp, value, sizeof(double), hipMemcpyHostToDevice, d.comm_stream));
        NCCL_TRY(rccl_api()->AllReduce(tmp, tmp, 1, ncclDouble, ncclMax, d.comm, d.comm_stream));
        HIP_TRY(hipMemcpyAsync(value, tmp, sizeof(double), hipMemcpyDeviceToHost, d.comm_stream));
        return comm_wait(d.comm_stream, "an all-reduce over the job's ranks");
    };
    rc = body();
    if (rc == FZ_ETIMEOUT) ctx->comm_broken = true;
    else (void)hipFree(tmp);
    return rc;
}

int fz_comm_barrier(fz_ctx *ctx) {
    if (ctx) for (DevState &d : ctx->devs) { (void)hipSetDevice(d.device); (void)hipStreamSynchronize(d.stream); (void)hipStreamSynchronize(d.stream_alt); }
    double one = 1.0;
    return fz_comm_max_f64(ctx, &one);
}

// Test hook: read the FZ_* switches from the environment again (they are read once, on first use; a test that changes one
// inside a running process calls this — no search may be in flight anywhere in the process).
void fz_debug_reload_switches(void) { switches_storage() = read_switches(); }

int fz_comm_backend(void) { return !rccl_api()->ok ? 0 : rccl_api()->stand_in ? 2 : 1; }

int fz_comm_gather_ms(fz_ctx *ctx, double *ms) {
    if (!ctx || !ms) return fail(FZ_EINVAL, "null argument");
    *ms = ctx->last_gather_ms;
    return FZ_OK;
}

int fz_debug_scan_regions(uint64_t ntiles, uint64_t grid, uint32_t n_cus, int steps, double fmin, int wg_per_cu, uint32_t *n_regions,
                          uint64_t *table) {
    if (!n_regions || !table) return fail(FZ_EINVAL, "null argument");
    static FzScanArgs fa;                                   // (1.7 KB: not on the stack of a ctypes call for nothing)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    plan_scan_regions(fa, ntiles, grid, n_cus, steps, fmin, wg_per_cu);
    *n_regions = fa.nreg;
    for (uint32_t r = 0; r < fa.nreg; ++r) {
        table[4 * r] = fa.reg_wg0[r]; table[4 * r + 1] = fa.reg_nwg[r]; table[4 * r + 2] = fa.reg_tile0[r]; table[4 * r + 3] = fa.reg_end[r];
    }
    return FZ_OK;
}

int fz_debug_gather_merge(const void *blocks, uint32_t world, uint64_t cap, const uint64_t *own_lo, uint32_t L, fz_match **out,
                          uint64_t *n_out, uint64_t *need_cap) {
    if (!blocks || !out || !n_out || !need_cap || world == 0) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n_out = 0; *need_cap = 0;
    std::vector<FzRec> recs;
    std::vector<size_t> seg_ends;
    std::vector<uint32_t> seg_order;
    uint64_t top = 0, total = 0;
    parse_gathered(static_cast<const uint8_t *>(blocks), (int)world, kHeaderBytes + cap * sizeof(FzRec), cap, own_lo, recs, seg_ends,
                   seg_order, &top, &total);
    if (top > cap) { *need_cap = (top + top / 4 + 1023) / 1024 * 1024; return FZ_OK; }
    return emit_matches_segments(nullptr, recs.data(), seg_ends, seg_order, L, out, n_out, 0, 0, true);
}

}  // extern "C"

// ---- (f)3: the reference's linear-programming fallbacks for short patterns, on the GPU ---------
namespace {

struct LpRec { int64_t start, end; int32_t dist; uint64_t step; uint64_t w0; uint32_t seq; };

constexpr uint32_t kLpStarts = 256;      // start positions owned by one window (one wave)

// Whole-sequence automaton (generic or Levenshtein), tiled by start position: tile w spawns the
// candidates of starts [w0, w0 + 256) and runs them over [w0, w0 + 256 + m + k) — a candidate lives
// at most m - 1 + k characters, so tiles are independent; only the tile reaching the sequence end
// performs the reference's end-of-sequence flush.
int run_lp(fz_ctx *ctx, fz_seq *seq, const Search &q, uint32_t lp_kind, std::vector<LpRec> &out) {
    // tile-relative start / end travel in 16 bits each: a tile's window is m + 2k + kLpStarts bytes
    if ((uint64_t)q.m + 2ull * q.k + kLpStarts > 65535ull)
        return fail(FZ_EUNSUPPORTED, "linear-programming route: len(subsequence) + 2 * max_l_dist + %u = %llu exceeds 65535 (16-bit window coordinates)",
                    kLpStarts, (unsigned long long)q.m + 2ull * q.k + kLpStarts);
    memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    uint32_t cand_cap = 1024;
    for (int attempt = 0; attempt < 8; ++attempt) {
        out.clear();
        bool rerun = false;
        const uint32_t mpad = (q.m + 15u) & ~15u, wpad = (q.m + 2 * q.k + kLpStarts + 15u) & ~15u;
        for (const Shard &sh : seq->shards) {
            DevState &d = ctx->devs[sh.dev];
            HIP_TRY(hipSetDevice(d.device));
            size_t lds = 0;
            uint64_t scratch = 0;
            int rc_lists = cand_lists(d, cand_cap, mpad + wpad + FZ_GEN_MCAP * 8, lds, scratch);
            if (rc_lists) return rc_lists;
            unsigned long long *counters = reinterpret_cast<unsigned long long *>(d.d_out);
            FzGenRec *recs = reinterpret_cast<FzGenRec *>(d.d_out + kHeaderBytes);
            HIP_TRY(hipMemsetAsync(d.d_out, 0, kHeaderBytes, d.stream));
            d.header_zeroed = false;
            HIP_TRY(hipEventRecord(d.ev[0], d.stream));
            FzScanArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.geom = sh.geom;
            fa.mode = q.mode; fa.m = q.m; fa.k = q.k; fa.L = 0;
            fa.max_subs = q.max_subs; fa.max_ins = q.max_ins; fa.max_dels = q.max_dels;
            fa.cand_cap = cand_cap;
            fa.cand_scratch = scratch;
            fa.lp_kind = lp_kind;
            fa.lp_starts = kLpStarts;
            fa.rec_cap = d.rec_cap;
            if (q.m <= FZ_MAX_M) memcpy(fa.pat, q.p, q.m);
            { int rcp = stage_pattern(d, fa, q.p, q.m); if (rcp) return rcp; }
            const uint64_t own = sh.geom.own_hi - sh.geom.own_lo;
            const uint64_t nwin = (own + kLpStarts - 1) / kLpStarts;
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(lp_kernel(lp_kind, scratch != 0)),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (nwin) {
                const unsigned grid = (unsigned)std::min<uint64_t>(nwin, scratch ? kCandScratchGrid : (uint64_t)d.n_cus * 64);
                hipLaunchKernelGGL(lp_kernel(lp_kind, scratch != 0), dim3(grid), dim3(64), lds, d.stream, sh.d_buf, fa, d.d_hits,
                                   nwin, recs, counters);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipEventRecord(d.ev[1], d.stream));
            HIP_TRY(hipMemcpyAsync(d.h_stage, d.d_out, kHeaderBytes + kFirstCopyRecs * sizeof(FzRec),
                                   hipMemcpyDeviceToHost, d.stream));
            HIP_TRY(hipEventRecord(d.ev[3], d.stream));
        }
        for (const Shard &sh : seq->shards) {
            DevState &d = ctx->devs[sh.dev];
            HIP_TRY(hipSetDevice(d.device));
            HIP_TRY(hipStreamSynchronize(d.stream));
            const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(d.h_stage);
            const uint64_t nr = cnt[1], novf = cnt[2];
            if (nr > d.rec_cap) { int rc = ensure_recs(d, nr + nr / 8 + 1024); if (rc) return rc; rerun = true; }
            if (novf) { cand_cap *= 4; rerun = true; }
            if (rerun) continue;
            float f = 0, t = 0;
            HIP_TRY(hipEventElapsedTime(&f, d.ev[0], d.ev[1]));
            HIP_TRY(hipEventElapsedTime(&t, d.ev[0], d.ev[3]));
            ctx->stats.verify_ms = std::max<double>(ctx->stats.verify_ms, f);
            ctx->stats.device_ms = std::max<double>(ctx->stats.device_ms, t);
            ctx->stats.bytes_scanned += sh.geom.buf_len;
            std::vector<FzGenRec> tmp(nr);
            const uint64_t first = std::min<uint64_t>(nr, kFirstCopyRecs);
            if (first) memcpy(tmp.data(), d.h_stage + kHeaderBytes, first * sizeof(FzGenRec));
            if (nr > first)
                HIP_TRY(hipMemcpy(tmp.data() + first, d.d_out + kHeaderBytes + first * sizeof(FzGenRec),
                                  (nr - first) * sizeof(FzGenRec), hipMemcpyDeviceToHost));
            for (const FzGenRec &r : tmp) {
                LpRec o;
                o.w0 = sh.geom.own_lo + (uint64_t)r.win * kLpStarts;
                o.start = (int64_t)(o.w0 + (r.se & 0xffffu));
                o.end = (int64_t)(o.w0 + (r.se >> 16));
                o.dist = (int32_t)r.dist;
                o.step = r.key;
                o.seq = r.seq;
                out.push_back(o);
            }
        }
        if (!rerun) { ctx->stats.raw_matches = out.size(); return FZ_OK; }
    }
    return fail(FZ_EUNSUPPORTED, "automaton candidate lists / result buffers kept overflowing");
}

// one-process-per-GPU jobs: the ranks' records back to back (emit_lp orders them by step / tile / list position,
// all of which are global)
int lp_exchange(fz_ctx *ctx, std::vector<LpRec> &recs) {
    if (!comm_multi_process(ctx)) return FZ_OK;
    std::vector<uint8_t> all;
    int rc = comm_gather_host(ctx, recs.data(), recs.size() * sizeof(LpRec), all);
    if (rc) return rc;
    recs.resize(all.size() / sizeof(LpRec));
    if (!all.empty()) memcpy(static_cast<void *>(recs.data()), all.data(), all.size());
    ctx->stats.raw_matches = recs.size();
    return FZ_OK;
}

int emit_lp(std::vector<LpRec> &recs, bool newest_first, fz_match **out, uint64_t *n) {
    // reference emission order: by sequence step; within a step by candidate-list order, i.e. by
    // start ascending (generic: fresh candidates are appended) or descending (Levenshtein: the fresh
    // candidate goes first); within one tile the kernel already kept list order (seq).
    std::sort(recs.begin(), recs.end(), [newest_first](const LpRec &a, const LpRec &b) {
        if (a.step != b.step) return a.step < b.step;
        if (a.w0 != b.w0) return newest_first ? a.w0 > b.w0 : a.w0 < b.w0;
        return a.seq < b.seq;
    });
    void *mem = nullptr;
    int rc = alloc_out(recs.size(), sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *mo = static_cast<fz_match *>(mem);
    for (size_t i = 0; i < recs.size(); ++i) {
        mo[i].start = recs[i].start; mo[i].end = recs[i].end; mo[i].dist = recs[i].dist; mo[i].block = -1;
    }
    *out = mo;
    *n = recs.size();
    return FZ_OK;
}

}  // namespace

extern "C" int fz_lev_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    int rc = validate(ctx, seq, p, m);
    if (rc) return rc;
    if (k >= m) {                                              // levenshtein.py:62-65: data independent
        const uint64_t cnt = seq->n + 1;
        void *mem = nullptr;
        rc = alloc_out(cnt, sizeof(fz_match), &mem);
        if (rc) return rc;
        fz_match *mo = static_cast<fz_match *>(mem);
        for (uint64_t i = 0; i < cnt; ++i) { mo[i].start = mo[i].end = (int64_t)i; mo[i].dist = (int32_t)m; mo[i].block = -1; }
        *out = mo; *n = cnt;
        return FZ_OK;
    }
    if (k > FZ_MAX_K) return fail(FZ_EUNSUPPORTED, "max_l_dist above %d is not supported", FZ_MAX_K);
    rc = check_halo(seq, (uint64_t)m + k);
    if (rc) return rc;
    Search q;
    q.mode = FZ_MODE_LEV; q.m = m; q.k = k; q.p = p;
    std::vector<LpRec> recs;
    rc = run_lp(ctx, seq, q, FZ_LP_LEV_SEQ, recs);
    if (rc == FZ_OK) rc = lp_exchange(ctx, recs);
    if (rc) return rc;
    return emit_lp(recs, /*newest_first=*/true, out, n);
}

extern "C" int fz_generic_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                             uint32_t max_dels, uint32_t max_l, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    int rc = validate(ctx, seq, p, m);
    if (rc) return rc;
    if (max_l > FZ_MAX_K) return fail(FZ_EUNSUPPORTED, "max_l_dist above %d is not supported", FZ_MAX_K);
    rc = check_halo(seq, (uint64_t)m + max_l);
    if (rc) return rc;
    Search q;
    q.mode = FZ_MODE_GENERIC; q.m = m; q.k = max_l; q.p = p;
    q.max_subs = std::min(max_subs, 255u); q.max_ins = std::min(max_ins, 255u); q.max_dels = std::min(max_dels, 255u);
    std::vector<LpRec> recs;
    rc = run_lp(ctx, seq, q, FZ_LP_GENERIC_SEQ, recs);
    if (rc == FZ_OK) rc = lp_exchange(ctx, recs);
    if (rc) return rc;
    return emit_lp(recs, /*newest_first=*/false, out, n);
}

static int subs_lp_impl(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n, int *found);

extern "C" int fz_subs_lp(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n) {
    if (!out || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n = 0;
    return subs_lp_impl(ctx, seq, p, m, k, out, n, nullptr);
}

extern "C" int fz_subs_lp_any(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, int *found) {
    if (!found) return fail(FZ_EINVAL, "null argument");
    *found = 0;
    return subs_lp_impl(ctx, seq, p, m, k, nullptr, nullptr, found);
}

static int subs_lp_impl(fz_ctx *ctx, fz_seq *seq, const uint8_t *p, uint32_t m, uint32_t k, fz_match **out, uint64_t *n, int *found) {
    int rc = validate(ctx, seq, p, m);
    if (rc) return rc;
    rc = check_halo(seq, m);
    if (rc) return rc;
    memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    std::vector<FzRec> recs;
    for (int attempt = 0; attempt < 4; ++attempt) {
        recs.clear();
        bool rerun = false;
        for (const Shard &sh : seq->shards) {
            DevState &d = ctx->devs[sh.dev];
            HIP_TRY(hipSetDevice(d.device));
            unsigned long long *counters = reinterpret_cast<unsigned long long *>(d.d_out);
            FzRec *drecs = reinterpret_cast<FzRec *>(d.d_out + kHeaderBytes);
            HIP_TRY(hipMemsetAsync(d.d_out, 0, kHeaderBytes, d.stream));
            d.header_zeroed = false;
            FzScanArgs fa;
            memset(&fa, 0, sizeof fa);
            fa.geom = sh.geom; fa.mode = FZ_MODE_SUBS; fa.m = m; fa.k = k; fa.rec_cap = d.rec_cap;
            fa.flags = found ? FZ_FLAG_ANY : 0u;
            if (m <= FZ_MAX_M) memcpy(fa.pat, p, m);
            rc = stage_pattern(d, fa, p, m);
            if (rc) return rc;
            const uint64_t own = sh.geom.own_hi - sh.geom.own_lo;
            const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((own + 255) / 256, (uint64_t)d.n_cus * 16));
            hipLaunchKernelGGL(fz_hamming_kernel, dim3(grid), dim3(256), 0, d.stream, sh.d_buf, fa, drecs, counters);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(d.h_stage, d.d_out, kHeaderBytes + kFirstCopyRecs * sizeof(FzRec),
                                   hipMemcpyDeviceToHost, d.stream));
        }
        for (const Shard &sh : seq->shards) {
            DevState &d = ctx->devs[sh.dev];
            HIP_TRY(hipSetDevice(d.device));
            HIP_TRY(hipStreamSynchronize(d.stream));
            const unsigned long long *cnt = reinterpret_cast<const unsigned long long *>(d.h_stage);
            const uint64_t nr = cnt[1];
            if (found) { if (nr) *found = 1; continue; }
            if (nr > d.rec_cap) { rc = ensure_recs(d, nr + nr / 8 + 1024); if (rc) return rc; rerun = true; continue; }
            const size_t base = recs.size();
            recs.resize(base + nr);
            const uint64_t first = std::min<uint64_t>(nr, kFirstCopyRecs);
            if (first) memcpy(recs.data() + base, d.h_stage + kHeaderBytes, first * sizeof(FzRec));
            if (nr > first)
                HIP_TRY(hipMemcpy(recs.data() + base + first, d.d_out + kHeaderBytes + first * sizeof(FzRec),
                                  (nr - first) * sizeof(FzRec), hipMemcpyDeviceToHost));
            ctx->stats.bytes_scanned += sh.geom.buf_len;
        }
        if (!rerun) break;
        if (attempt == 3) return fail(FZ_EDEVICE, "result buffers kept overflowing");
    }
    if (found) {
        bool any = *found != 0;
        if (comm_multi_process(ctx)) { rc = comm_or(ctx, any); if (rc) return rc; }
        *found = any ? 1 : 0;
        return FZ_OK;
    }
    if (comm_multi_process(ctx)) {
        std::vector<uint8_t> all;
        rc = comm_gather_host(ctx, recs.data(), recs.size() * sizeof(FzRec), all);
        if (rc) return rc;
        recs.resize(all.size() / sizeof(FzRec));
        if (!all.empty()) memcpy(static_cast<void *>(recs.data()), all.data(), all.size());
    }
    sort_recs(recs);
    void *mem = nullptr;
    rc = alloc_out(recs.size(), sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *mo = static_cast<fz_match *>(mem);
    for (size_t i = 0; i < recs.size(); ++i) {
        mo[i].start = (int64_t)recs[i].key; mo[i].end = (int64_t)recs[i].key + m; mo[i].dist = (int32_t)recs[i].dist; mo[i].block = -1;
    }
    *out = mo; *n = recs.size();
    ctx->stats.raw_matches = recs.size();
    return FZ_OK;
}

// ---- (f)2 / a12: find_near_matches_in_file as a pipeline -------------------------------------------
// The reference reads a file in chunks and searches every chunk as an independent sequence
// (__init__.py:129-200).  Here the file crosses PCIe in large batches (many chunks each) from pinned,
// double-buffered staging memory; the chunks of a batch are SEGMENTS of one resident buffer with their
// own clamps (FzGeom), searched by one launch, while the host already fills the next staging buffer.
struct fz_stream {
    fz_ctx *ctx = nullptr;
    uint32_t mode = 0, m = 0, k = 0, max_subs = 0, max_ins = 0, max_dels = 0;
    std::vector<uint8_t> pat;
    uint64_t S = 0;                      // segment stride
    uint32_t pre = 0, post = 0;          // segment j = [j*S - pre, (j+1)*S + post) clipped to the file
    uint64_t cap = 0;                    // bytes a staging buffer holds
    uint64_t cap_alloc = 0;              // bytes the staging allocations really have (reused buffers may be larger)
    uint8_t *h_buf[2] = {nullptr, nullptr};   // pinned staging
    int cur = 0;                         // staging buffer being filled
    uint64_t stage_off = 0;              // file offset of h_buf[cur][0]
    uint64_t stage_len = 0;              // valid bytes in h_buf[cur]
    uint64_t next_seg = 0;               // first segment not launched yet
    bool eof = false;
    fz_seq *seq = nullptr;               // one transient shard over the device batch buffer
    bool inflight = false;
    uint64_t fl_j0 = 0, fl_j1 = 0;       // segments of the batch in flight
    Search q;
    std::vector<fz_match> out;
    std::vector<uint32_t> out_seg;
    uint64_t bytes_total = 0;
    double t_fill = 0, t_collect = 0, t_launch = 0, t_carry = 0;   // FZ_STREAM_TRACE=1: where the host's time went (ms)
};

namespace {

uint64_t stream_segments_at_eof(const fz_stream *st, uint64_t F) {
    if (F == 0) return 0;
    if (F <= st->post) return 1;
    return (F - st->post - 1) / st->S + 1;                 // segment j >= 1 exists iff j*S + post < F
}

// segment that owns the window [i0, i0 + m) of an exact / substitutions-only match: such windows fit
// exactly one chunk (the overlap is m - 1 items)
uint64_t stream_window_segment(const fz_stream *st, uint64_t i0) {
    return st->post ? i0 / st->S : (i0 + st->m - 1) / st->S;
}

int stream_build_search(fz_stream *st) {
    Search &q = st->q;
    q = Search();
    q.mode = st->mode; q.m = st->m; q.k = st->k; q.p = st->pat.data();
    q.max_subs = st->max_subs; q.max_ins = st->max_ins; q.max_dels = st->max_dels;
    if (st->mode == FZ_MODE_EXACT) {
        q.plan.L = st->m;
        q.plan.s = {0};
        return FZ_OK;
    }
    const uint32_t L = st->m / (st->k + 1);
    if (L == 0) return fail(FZ_EINVAL, "the subsequence length must be greater than the distance limit");
    q.plan.L = L;
    for (uint32_t s = 0; s + L <= st->m; s += L) q.plan.s.push_back(s);
    if (q.plan.s.size() > FZ_MAX_BLOCKS) return fail(FZ_EUNSUPPORTED, "more than %u n-gram blocks", FZ_MAX_BLOCKS);
    return FZ_OK;
}

int stream_collect(fz_stream *st) {
    if (!st->inflight) return FZ_OK;
    fz_ctx *ctx = st->ctx;
    st->inflight = false;
    ctx->stream_inflight = nullptr;
    const uint32_t L = st->q.plan.L;
    fz_match *mo = nullptr;
    uint32_t *so = nullptr;
    uint64_t cnt = 0;
    int rc = FZ_OK;
    if (st->mode == FZ_MODE_GENERIC) {
        std::vector<FzGenRec> recs;
        rc = run_generic(ctx, st->seq, st->q, recs);
        if (rc) return rc;
        rc = emit_generic(ctx, st->seq, recs, L, st->k, &mo, &cnt, &so);
        if (rc) return rc;
    } else {
        std::vector<FzRec> recs;
        std::vector<uint64_t> hits;
        const bool with_verify = st->mode != FZ_MODE_EXACT;
        rc = search_collect(ctx, st->seq, st->q, with_verify, recs, hits);
        if (rc) return rc;
        if (st->mode == FZ_MODE_LEV) {
            const FzRec *r = ctx->view ? ctx->view : recs.data();
            const size_t nr = ctx->view ? (size_t)ctx->view_n : recs.size();
            rc = emit_matches_seg(r, nr, L, &mo, &cnt, &so);
            if (rc) return rc;
        } else {
            // exact / substitutions-only: no clamp depends on the chunk; the chunk of a window follows from
            // its position, and this batch keeps the windows of its own chunks
            std::vector<fz_match> tmp;
            if (st->mode == FZ_MODE_EXACT) {
                std::sort(hits.begin(), hits.end());
                tmp.resize(hits.size());
                for (size_t i = 0; i < hits.size(); ++i) {
                    const int64_t idx = (int64_t)fz_hit_index(hits[i]);
                    tmp[i] = fz_match{idx, idx + (int64_t)st->m, 0, 0};
                }
            } else {
                fz_match *em = nullptr;
                uint64_t en = 0;
                rc = emit_matches(ctx, recs, L, &em, &en);
                if (rc) return rc;
                tmp.assign(em, em + en);
                release_out(em);
            }
            std::vector<std::pair<uint64_t, uint32_t>> order;      // (segment, position in tmp)
            for (size_t i = 0; i < tmp.size(); ++i) {
                const uint64_t j = stream_window_segment(st, (uint64_t)tmp[i].start);
                if (j >= st->fl_j0 && j < st->fl_j1) order.emplace_back(j, (uint32_t)i);
            }
            std::stable_sort(order.begin(), order.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) { return x.first < y.first; });
            for (const auto &o : order) { st->out.push_back(tmp[o.second]); st->out_seg.push_back((uint32_t)o.first); }
            return FZ_OK;
        }
    }
    st->out.insert(st->out.end(), mo, mo + cnt);
    st->out_seg.insert(st->out_seg.end(), so, so + cnt);
    release_out(mo);
    release_out(so);
    return FZ_OK;
}

// Upload the staged bytes the segments [j0, j1) need and launch their search (asynchronous).
int stream_launch(fz_stream *st, uint64_t j0, uint64_t j1, uint64_t data_hi) {
    fz_ctx *ctx = st->ctx;
    // the context's result slots, counters and hit list serve ONE search at a time
    if (ctx->npend || (ctx->stream_inflight && ctx->stream_inflight != st))
        return fail(FZ_EINVAL, "another search or file stream of this context is in flight");
    DevState &d = ctx->devs[0];
    Shard &sh = st->seq->shards[0];
    HIP_TRY(hipSetDevice(d.device));
    const uint64_t len = data_hi - st->stage_off;
    FzGeom g{};
    g.n = data_hi;
    g.buf_off = st->stage_off;
    g.buf_len = len;
    g.own_lo = 0;
    g.own_hi = data_hi;
    const bool segmented = st->mode == FZ_MODE_LEV || st->mode == FZ_MODE_GENERIC;
    if (segmented) {
        g.seg_stride = st->S; g.seg_org = 0; g.seg_pre = st->pre; g.seg_post = st->post;
        g.seg_j0 = j0; g.seg_j1 = j1;
    }
    sh.geom = g;
    st->seq->n = data_hi;
    if (len) HIP_TRY(hipMemcpyAsync(sh.d_buf, st->h_buf[st->cur], len, hipMemcpyHostToDevice, d.stream));
    const uint64_t tiles = (len + FZ_TILE_BYTES - 1) / FZ_TILE_BYTES;
    const uint64_t body = std::max<uint64_t>(1, tiles) * FZ_TILE_BYTES;
    HIP_TRY(hipMemsetAsync(sh.d_buf + len, 0, body + FZ_PAD_BACK - len, d.stream));
    st->fl_j0 = j0;
    st->fl_j1 = j1;
    int rc = ensure_hits(d, std::max<uint64_t>(1u << 20, len / 64));
    if (rc) return rc;
    if (st->mode != FZ_MODE_GENERIC) {
        memset(&ctx->stats, 0, sizeof ctx->stats); ctx->tref.clear();
        ctx->stats.n_devices = 1;
        rc = search_enqueue(ctx, st->seq, st->q, st->mode != FZ_MODE_EXACT);
        if (rc) return rc;
    }
    st->inflight = true;
    ctx->stream_inflight = st;
    return FZ_OK;
}

}  // namespace

extern "C" {

int fz_stream_open(fz_ctx *ctx, uint32_t mode, const uint8_t *p, uint32_t m, uint32_t max_subs, uint32_t max_ins,
                   uint32_t max_dels, uint32_t k, uint64_t seg_stride, uint32_t seg_pre, uint32_t seg_post,
                   uint64_t batch_bytes, fz_stream **out) {
    if (!out) return fail(FZ_EINVAL, "null argument");
    *out = nullptr;
    if (!ctx || ctx->devs.empty()) return fail(FZ_EINVAL, "bad ctx handle");
    if (ctx->devs.size() != 1) return fail(FZ_EUNSUPPORTED, "file streams run on a single-device context");
    if (ctx->npend || ctx->stream_inflight) return fail(FZ_EINVAL, "another search of this context is in flight");
    if (!p || m == 0) return fail(FZ_EINVAL, "subsequence must not be empty");
    if (m > FZ_MAX_M_ANY) return fail(FZ_EUNSUPPORTED, "subsequence longer than %u bytes", FZ_MAX_M_ANY);
    if (mode > FZ_MODE_GENERIC) return fail(FZ_EINVAL, "bad mode");
    if (mode == FZ_MODE_GENERIC && (uint64_t)m + 2ull * k > 65535ull)
        return fail(FZ_EUNSUPPORTED, "generic search: len(subsequence) + 2 * max_l_dist exceeds 65535 (16-bit window coordinates)");
    if (k > (mode == FZ_MODE_GENERIC ? (uint32_t)FZ_MAX_K : FZ_MAX_K_ANY)) return fail(FZ_EUNSUPPORTED, "distance limit %u is not supported", k);
    if (seg_stride == 0 || (seg_pre && seg_post)) return fail(FZ_EINVAL, "bad segment geometry");
    // a position may lie in at most two segments, and a chunk must hold a whole pattern window with its reach
    if ((uint64_t)seg_pre + seg_post > seg_stride / 2 || (uint64_t)m + 2ull * k + 2 > seg_stride)
        return fail(FZ_EUNSUPPORTED, "chunk too small for this pattern (use the per-chunk path)");
    fz_stream *st = new (std::nothrow) fz_stream();
    if (!st) return fail(FZ_ENOMEM, "out of memory");
    st->ctx = ctx;
    st->mode = mode; st->m = m; st->k = k;
    st->max_subs = std::min(max_subs, 255u); st->max_ins = std::min(max_ins, 255u); st->max_dels = std::min(max_dels, 255u);
    st->pat.assign(p, p + m);
    st->S = seg_stride; st->pre = seg_pre; st->post = seg_post;
    int rc = stream_build_search(st);
    if (rc) { delete st; return rc; }
    const uint64_t ext = (uint64_t)seg_pre + seg_post;
    st->cap = std::max<uint64_t>(batch_bytes, seg_stride + ext) + seg_stride + ext;   // >= one whole chunk + the carried overlap
    DevState &d = ctx->devs[0];
    auto init = [&]() -> int {
        HIP_TRY(hipSetDevice(d.device));
        const uint64_t need_d = FZ_PAD_FRONT + ((st->cap + FZ_TILE_BYTES - 1) / FZ_TILE_BYTES + 1) * FZ_TILE_BYTES + FZ_PAD_BACK;
        uint8_t *cached_d = nullptr;
        if (d.stream_h[0] && d.stream_cap >= st->cap && d.stream_d_bytes >= need_d) {       // reuse the previous stream's buffers
            st->h_buf[0] = d.stream_h[0]; st->h_buf[1] = d.stream_h[1];
            cached_d = d.stream_d;
            st->cap_alloc = d.stream_cap;
            d.stream_h[0] = d.stream_h[1] = nullptr; d.stream_d = nullptr;
        } else {
            for (int i = 0; i < 2; ++i) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&st->h_buf[i]), st->cap, hipHostMallocDefault));
            st->cap_alloc = st->cap;
        }
        st->seq = new (std::nothrow) fz_seq();
        if (!st->seq) return fail(FZ_ENOMEM, "out of memory");
        st->seq->ctx = ctx;
        st->seq->shards.emplace_back();
        Shard &sh = st->seq->shards[0];
        sh.dev = 0;
        if (cached_d) {
            sh.d_alloc = cached_d;
            sh.alloc_bytes = d.stream_d_bytes;
        } else {
            sh.alloc_bytes = need_d;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sh.d_alloc), sh.alloc_bytes));
        }
        sh.d_buf = sh.d_alloc + FZ_PAD_FRONT;
        HIP_TRY(hipMemsetAsync(sh.d_alloc, 0, FZ_PAD_FRONT, d.stream));
        return FZ_OK;
    };
    rc = init();
    if (rc) { fz_stream_close(st); return rc; }
    *out = st;
    return FZ_OK;
}

int fz_stream_buffer(fz_stream *st, uint8_t **host, uint64_t *capacity) {
    if (!st || !host || !capacity) return fail(FZ_EINVAL, "null argument");
    *host = st->h_buf[st->cur] + st->stage_len;
    *capacity = st->eof ? 0 : st->cap - st->stage_len;
    return FZ_OK;
}

int fz_stream_submit(fz_stream *st, uint64_t nbytes, int last) {
    if (!st) return fail(FZ_EINVAL, "null argument");
    if (st->eof) return fail(FZ_EINVAL, "the stream has already seen its last bytes");
    if (nbytes > st->cap - st->stage_len) return fail(FZ_EINVAL, "more bytes than the staging buffer holds");
    st->stage_len += nbytes;
    st->bytes_total += nbytes;
    if (last) st->eof = true;
    const uint64_t data_end = st->stage_off + st->stage_len;
    const uint64_t ext_lo = st->pre, ext_hi = st->post;
    for (;;) {
        uint64_t j1;
        if (st->eof) j1 = stream_segments_at_eof(st, data_end);
        else j1 = data_end >= ext_hi + st->S ? (data_end - ext_hi) / st->S : 0;      // (j + 1) * S + post <= data_end
        if (j1 <= st->next_seg) {
            if (st->eof || st->stage_len < st->cap) break;
            return fail(FZ_EUNSUPPORTED, "staging buffer smaller than one chunk");
        }
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        const auto t0 = now();
        int rc = stream_collect(st);                        // the previous batch (its staging buffer becomes free)
        if (rc) return rc;
        const auto t1 = now();
        const uint64_t data_hi = st->eof ? data_end : j1 * st->S + ext_hi;
        rc = stream_launch(st, st->next_seg, j1, data_hi);
        if (rc) return rc;
        const auto t2 = now();
        st->t_collect += ms(t0, t1);
        st->t_launch += ms(t1, t2);
        st->next_seg = j1;
        if (st->eof) break;
        // carry the bytes the next segments need into the other staging buffer
        const uint64_t keep_from = j1 * st->S > ext_lo ? j1 * st->S - ext_lo : 0;
        const uint64_t from = std::max(keep_from, st->stage_off);
        const uint64_t carry = data_end - from;
        const int other = st->cur ^ 1;
        memcpy(st->h_buf[other], st->h_buf[st->cur] + (from - st->stage_off), carry);
        st->t_carry += ms(t2, now());
        st->cur = other;
        st->stage_off = from;
        st->stage_len = carry;
        break;
    }
    return FZ_OK;
}

namespace {

// Readers of fz_stream_read_fd: page cache -> pinned staging is a memcpy (one core moves ~5-10 GB/s, the PCIe link
// ~55).  The threads live for the whole file and take 1 MiB pieces of the current batch from a shared counter;
// starting and joining 16-48 threads for every 64 MiB batch cost ~0.5 ms of the batch's ~1.7 ms.
struct ReadPool {
    static constexpr uint64_t kPiece = 1u << 20;
    std::mutex mu;
    std::condition_variable go, done;
    uint64_t generation = 0;
    int busy = 0;
    bool stop = false;
    int fd = -1;
    uint8_t *dst = nullptr;
    uint64_t room = 0, pos = 0, npieces = 0;
    std::atomic<uint64_t> next{0};
    std::vector<uint64_t> got;           // bytes read per piece
    std::atomic<int> err{0};
    std::vector<std::thread> workers;

    explicit ReadPool(int n) {
        for (int t = 0; t < n; ++t) workers.emplace_back([this]() { run(); });
    }
    ~ReadPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        go.notify_all();
        for (auto &w : workers) w.join();
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                go.wait(lk, [&]() { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            for (;;) {
                const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= npieces) break;
                const uint64_t lo = i * kPiece, hi = std::min(room, lo + kPiece);
                uint64_t n = 0;
                while (lo + n < hi) {
                    const ssize_t r = pread(fd, dst + lo + n, hi - lo - n, (off_t)(pos + lo + n));
                    if (r < 0) { err.store(1); break; }
                    if (r == 0) break;
                    n += (uint64_t)r;
                }
                got[i] = n;
            }
            std::lock_guard<std::mutex> g(mu);
            if (--busy == 0) done.notify_one();
        }
    }
    // fill dst[0, room) from the file at `pos`; -> bytes valid from the front (short at the end of the file)
    int fill(int fd_, uint8_t *dst_, uint64_t room_, uint64_t pos_, uint64_t &n, bool &short_read) {
        fd = fd_; dst = dst_; room = room_; pos = pos_;
        npieces = (room + kPiece - 1) / kPiece;
        got.assign(npieces, 0);
        next.store(0);
        {
            std::unique_lock<std::mutex> lk(mu);
            busy = (int)workers.size();
            ++generation;
            go.notify_all();
            done.wait(lk, [&]() { return busy == 0; });
        }
        if (err.load()) return fail(FZ_EDEVICE, "pread failed");
        n = 0;
        short_read = false;
        for (uint64_t i = 0; i < npieces && !short_read; ++i) {
            n += got[i];
            if (got[i] < std::min(room, (i + 1) * kPiece) - i * kPiece) short_read = true;   // end of file inside this piece
        }
        return FZ_OK;
    }
};

}  // namespace

int fz_stream_read_fd(fz_stream *st, int fd, int64_t offset, int threads, uint64_t *total) {
    if (!st || fd < 0 || offset < 0) return fail(FZ_EINVAL, "bad argument");
    if (threads <= 0) threads = (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency() / 2));
    uint64_t pos = (uint64_t)offset, sum = 0;
    ReadPool pool(threads);
    while (!st->eof) {
        uint8_t *dst = nullptr;
        uint64_t room = 0;
        int rc = fz_stream_buffer(st, &dst, &room);
        if (rc) return rc;
        if (room == 0) return fail(FZ_EDEVICE, "internal: no staging room");
        uint64_t n = 0;
        bool short_read = false;
        const auto tf = std::chrono::steady_clock::now();
        const bool nofill = sw().stream_nofill;     // lab: the H2D + scan pipeline alone
        if (nofill) {
            struct stat sb;
            if (fstat(fd, &sb) != 0) return fail(FZ_EDEVICE, "fstat failed");
            const uint64_t left = (uint64_t)sb.st_size > pos ? (uint64_t)sb.st_size - pos : 0;
            n = std::min(room, left);
            short_read = n < room;
        } else {
            rc = pool.fill(fd, dst, room, pos, n, short_read);
        }
        if (rc) return rc;
        st->t_fill += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count();
        rc = fz_stream_submit(st, n, short_read ? 1 : 0);
        if (rc) return rc;
        pos += n;
        sum += n;
    }
    if (total) *total = sum;
    return FZ_OK;
}

int fz_stream_finish(fz_stream *st, fz_match **out, uint32_t **seg, uint64_t *n) {
    if (!st || !out || !seg || !n) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *seg = nullptr; *n = 0;
    if (!st->eof) {
        int rc = fz_stream_submit(st, 0, 1);
        if (rc) return rc;
    }
    int rc = stream_collect(st);
    if (rc) return rc;
    if (sw().stream_trace)
        fprintf(stderr, "[fz_stream] %.1f MiB: fill %.2f ms, collect %.2f ms, launch %.2f ms, carry %.2f ms\n",
                st->bytes_total / 1048576.0, st->t_fill, st->t_collect, st->t_launch, st->t_carry);
    void *mem = nullptr, *smem_ = nullptr;
    rc = alloc_out(st->out.size(), sizeof(fz_match), &mem);
    if (rc) return rc;
    rc = alloc_out(st->out_seg.size(), sizeof(uint32_t), &smem_);
    if (rc) { release_out(mem); return rc; }
    if (!st->out.empty()) {
        memcpy(mem, st->out.data(), st->out.size() * sizeof(fz_match));
        memcpy(smem_, st->out_seg.data(), st->out_seg.size() * sizeof(uint32_t));
    }
    *out = static_cast<fz_match *>(mem);
    *seg = static_cast<uint32_t *>(smem_);
    *n = st->out.size();
    st->out.clear();
    st->out_seg.clear();
    return FZ_OK;
}

void fz_stream_close(fz_stream *st) {
    if (!st) return;
    fz_ctx *ctx = st->ctx;
    if (ctx && !ctx->devs.empty()) {
        DevState &d = ctx->devs[0];
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        if (ctx->stream_inflight == st) ctx->stream_inflight = nullptr;
    }
    // hand the staging to the context for the next stream (freed with the context)
    bool kept = false;
    if (ctx && !ctx->devs.empty() && st->h_buf[0] && st->h_buf[1] && st->seq && !st->seq->shards.empty() && st->seq->shards[0].d_alloc) {
        DevState &d = ctx->devs[0];
        if (!d.stream_h[0] || d.stream_cap < st->cap_alloc) {
            for (int i = 0; i < 2; ++i) if (d.stream_h[i]) (void)hipHostFree(d.stream_h[i]);
            if (d.stream_d) (void)hipFree(d.stream_d);
            d.stream_h[0] = st->h_buf[0]; d.stream_h[1] = st->h_buf[1];
            d.stream_d = st->seq->shards[0].d_alloc;
            d.stream_cap = st->cap_alloc;
            d.stream_d_bytes = st->seq->shards[0].alloc_bytes;
            kept = true;
        }
    }
    if (!kept) {
        for (int i = 0; i < 2; ++i) if (st->h_buf[i]) (void)hipHostFree(st->h_buf[i]);
        if (st->seq) for (Shard &sh : st->seq->shards) if (sh.d_alloc) (void)hipFree(sh.d_alloc);
    }
    delete st->seq;
    delete st;
}

}  // extern "C"

namespace {

// Second stage of the consolidation: (hull, best match) pairs of connected sets of matches — from the host's run
// folding (fz_consolidate) or from the device's per-hit folding (fz_generic_ngrams_consolidated) — are ordered by
// (hull start, zero-length first) and a sweep merges overlapping hulls; the survivors are sorted by (start, end, dist).
int consolidate_hulls(std::vector<Hull> &hulls, fz_match **out, uint64_t *n_out) {
    auto by_start_end_dist = [](const fz_match &a, const fz_match &b) {
        if (a.start != b.start) return a.start < b.start;
        if (a.end != b.end) return a.end < b.end;
        if (a.dist != b.dist) return a.dist < b.dist;
        return a.block < b.block;                              // (equal matches of different n-gram hits: a fixed order)
    };
    auto better = [](const fz_match &x, const fz_match &y) {
        const int64_t lx = x.end - x.start, ly = y.end - y.start;
        return x.dist < y.dist || (x.dist == y.dist && (lx > ly || (lx == ly && (x.start < y.start || (x.start == y.start && x.block < y.block)))));
    };
    const uint64_t nh = hulls.size();
    if (nh > 0xffffffffull) return fail(FZ_EUNSUPPORTED, "more than 2^32 disjoint runs of matches");
    // Sweep order = (hull start, zero-length first, input order).  The hulls of a search are spread over the sequence
    // (one or a few per n-gram hit), so ONE counting pass over ~nh equal slices of [smin, smax] leaves almost every hull
    // alone in its slice: count, prefix, scatter the hull numbers, then one pass over the slices that fixes the order
    // inside the few holding several (stable insertion; a stable sort for the crowded slices of clustered input) and
    // sweeps.  (This replaces an 11-bit LSD radix over 64-bit words — a histogram pass and three scatter passes; moving
    // the 40-byte hulls themselves into the slices instead of their numbers measured slower.)
    static thread_local std::vector<uint32_t> order, cnt;       // hull numbers in sweep order; slice boundaries
    Trace trc;
    // The stream of an n-gram search arrives block-major with every block's rows in index order: the hulls are a handful
    // of ASCENDING RUNS (one per n-gram block).  Those are merged and swept in one sequential pass — no counting pass, no
    // scatter, no gathers in random order (configs[1], 2 409 rows: 40 -> ~15 us on the host).  Ties go to the earlier
    // run, which is the input order the sweep order asks for.
    {
        constexpr uint32_t kMaxRuns = 8;
        uint64_t head[kMaxRuns], end[kMaxRuns];
        uint32_t nruns = nh ? 1u : 0u;
        head[0] = 0;
        auto hull_before = [](const Hull &a, const Hull &b) {
            if (a.h0 != b.h0) return a.h0 < b.h0;
            return (a.h1 == a.h0) && (b.h1 != b.h0);
        };
        for (uint64_t i = 1; i < nh && nruns <= kMaxRuns; ++i)
            if (hull_before(hulls[i], hulls[i - 1])) {
                if (nruns < kMaxRuns) { end[nruns - 1] = i; head[nruns] = i; }
                ++nruns;
            }
        if (nruns && nruns <= kMaxRuns) {
            end[nruns - 1] = nh;
            void *mem = nullptr;
            int rc = alloc_out(nh, sizeof(fz_match), &mem);
            if (rc) return rc;
            fz_match *best = static_cast<fz_match *>(mem);
            uint64_t nbest = 0;
            int64_t h0 = 0, h1 = 0;
            for (;;) {
                uint32_t r = nruns;
                for (uint32_t c = 0; c < nruns; ++c)
                    if (head[c] < end[c] && (r == nruns || hull_before(hulls[head[c]], hulls[head[r]]))) r = c;
                if (r == nruns) break;
                const Hull &h = hulls[head[r]++];
                if (nbest && !(h.h1 <= h0 || h.h0 >= h1)) {
                    h0 = std::min(h0, h.h0);
                    h1 = std::max(h1, h.h1);
                    if (better(h.best, best[nbest - 1])) best[nbest - 1] = h.best;
                } else {
                    h0 = h.h0; h1 = h.h1;
                    best[nbest++] = h.best;
                }
            }
            trc.mark("  runs merged + sweep");
            if (!std::is_sorted(best, best + nbest, by_start_end_dist)) std::sort(best, best + nbest, by_start_end_dist);
            trc.mark("  final order");
            *out = best;
            *n_out = nbest;
            return FZ_OK;
        }
    }
    order.resize(nh);
    auto before = [&](uint32_t x, uint32_t y) {
        const Hull &a = hulls[x], &b = hulls[y];
        if (a.h0 != b.h0) return a.h0 < b.h0;
        return (a.h1 == a.h0) && (b.h1 != b.h0);
    };
    uint64_t nb = 1;
    if (nh >= 32) {
        int64_t smin = hulls[0].h0, smax = smin;
        for (const Hull &h : hulls) { smin = std::min(smin, h.h0); smax = std::max(smax, h.h0); }
        while (nb < nh && nb < (1u << 24)) nb <<= 1;
        const uint64_t range = (uint64_t)(smax - smin);
        int shift = 0;
        while ((range >> shift) >= nb) ++shift;
        cnt.assign(nb + 1, 0u);
        for (uint64_t i = 0; i < nh; ++i) ++cnt[(((uint64_t)(hulls[i].h0 - smin)) >> shift) + 1];
        for (uint64_t b = 1; b <= nb; ++b) cnt[b] += cnt[b - 1];                      // cnt[b] = where slice b starts
        for (uint64_t i = 0; i < nh; ++i) order[cnt[((uint64_t)(hulls[i].h0 - smin)) >> shift]++] = (uint32_t)i;
        // now cnt[b] = where slice b ends
    } else {
        for (uint64_t i = 0; i < nh; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), before);
        cnt.assign(1, (uint32_t)nh);
    }
    trc.mark("  hulls in slices");
    // the survivors go straight into the caller's buffer (at most one per hull)
    void *mem = nullptr;
    int rc = alloc_out(nh, sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *best = static_cast<fz_match *>(mem);
    uint64_t nbest = 0;
    int64_t h0 = 0, h1 = 0;
    uint32_t lo = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint32_t hi = cnt[b];
        if (hi - lo > 1 && nh >= 32) {
            if (hi - lo > 24) std::stable_sort(order.begin() + lo, order.begin() + hi, before);
            else
                for (uint32_t i = lo + 1; i < hi; ++i) {
                    const uint32_t v = order[i];
                    uint32_t j = i;
                    while (j > lo && before(v, order[j - 1])) { order[j] = order[j - 1]; --j; }
                    order[j] = v;
                }
        }
        for (uint32_t vi = lo; vi < hi; ++vi) {
            if (vi + 12 < nh) __builtin_prefetch(&hulls[order[vi + 12]]);    // (the hulls are gathered in random order)
            const Hull &h = hulls[order[vi]];
            if (nbest && !(h.h1 <= h0 || h.h0 >= h1)) {
                h0 = std::min(h0, h.h0);
                h1 = std::max(h1, h.h1);
                if (better(h.best, best[nbest - 1])) best[nbest - 1] = h.best;
            } else {
                h0 = h.h0; h1 = h.h1;
                best[nbest++] = h.best;
            }
        }
        lo = hi;
    }
    trc.mark("  sweep");
    if (!std::is_sorted(best, best + nbest, by_start_end_dist)) std::sort(best, best + nbest, by_start_end_dist);
    trc.mark("  final order");
    *out = best;
    *n_out = nbest;
    return FZ_OK;
}

}  // namespace

extern "C" {

// ---- host-side consolidation (common.py:145-189) -------------------------------------------
// The partition into overlap groups is input-order independent, so for
// fz_consolidate a sort + sweep replaces the reference's O(M * groups) loop.
int fz_consolidate(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out) {
    if (!out || !n_out || (!in && n)) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    // Two stages.  (1) One pass in input order folds every row that overlaps the running hull of the rows right
    // before it into that hull (the reference's own group test, common.py:150-159, applied to consecutive rows;
    // the partition is order independent).  The records of one n-gram hit are emitted back to back and nearly all
    // overlap, so the 2.1e5 rows of configs[3b] leave ~6e3 (hull, best row) pairs; streams without such runs pass
    // through unchanged.  (2) The pairs are ordered by (hull start, zero-length first) — 11-bit LSD radix passes on
    // one 64-bit word per pair when there are many, a comparison sort otherwise (2e5 24-byte rows: ~6 ms) — and a
    // sweep merges overlapping hulls.  Zero-length matches (start == end) never overlap anything under
    // `not (end <= g.start or start >= g.end)` unless strictly inside a group, so both stages use that predicate
    // against the running hull instead of assuming sorted-interval merging.
    auto better = [](const fz_match &x, const fz_match &y) {
        const int64_t lx = x.end - x.start, ly = y.end - y.start;
        return x.dist < y.dist || (x.dist == y.dist && (lx > ly || (lx == ly && (x.start < y.start || (x.start == y.start && x.block < y.block)))));
    };
    static thread_local std::vector<Hull> hulls;                // scratch kept per thread between calls
    hulls.clear();
    Trace trc;
    for (uint64_t i = 0; i < n; ++i) {
        const fz_match &mt = in[i];
        if (!hulls.empty()) {
            Hull &h = hulls.back();
            if (!(mt.end <= h.h0 || mt.start >= h.h1)) {
                h.h0 = std::min(h.h0, mt.start);
                h.h1 = std::max(h.h1, mt.end);
                if (better(mt, h.best)) h.best = mt;
                continue;
            }
        }
        hulls.push_back(Hull{mt.start, mt.end, mt});
    }
    trc.mark("  runs folded");
    return consolidate_hulls(hulls, out, n_out);
}

int fz_merge_ranks(const fz_match *const *parts, const uint64_t *counts, const uint64_t *block_counts,
                   uint32_t world, uint32_t nb, fz_match *out) {
    if (!parts || !counts || !block_counts || !out) return fail(FZ_EINVAL, "null argument");
    std::vector<uint64_t> pos(world, 0);                       // read position inside every rank's stream
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t sum = 0;
        for (uint32_t g = 0; g < nb; ++g) sum += block_counts[(size_t)r * nb + g];
        if (sum != counts[r]) return fail(FZ_EINVAL, "rank %u: block counts do not add up to its record count", r);
    }
    uint64_t o = 0;
    for (uint32_t g = 0; g < nb; ++g)
        for (uint32_t r = 0; r < world; ++r) {
            const uint64_t c = block_counts[(size_t)r * nb + g];
            if (c) memcpy(out + o, parts[r] + pos[r], c * sizeof(fz_match));
            o += c;
            pos[r] += c;
        }
    return FZ_OK;
}

namespace {
struct WireRow { int64_t start; uint32_t len; uint16_t dist; uint16_t block; };
struct WireHeader { uint64_t count, nblocks; uint32_t per_block[256]; };
static_assert(sizeof(WireRow) == 16 && sizeof(WireHeader) == FZ_WIRE_HEADER_ROWS * 16, "wire layout");
}  // namespace

int fz_wire_pack(const fz_match *in, uint64_t n, uint64_t cap_rows, void *dst) {
    if ((!in && n) || !dst) return fail(FZ_EINVAL, "null argument");
    WireHeader *h = static_cast<WireHeader *>(dst);
    WireRow *rows = reinterpret_cast<WireRow *>(h + 1);
    memset(h, 0, sizeof *h);
    h->count = n;
    const uint64_t fit = std::min(n, cap_rows);
    uint32_t nb = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const int32_t b = in[i].block;
        if (b < 0 || b > 255 || in[i].end < in[i].start || in[i].end - in[i].start > 0xffffffffll || in[i].dist < 0 ||
            in[i].dist > 0xffff)
            return fail(FZ_EUNSUPPORTED, "record %llu does not fit the torch glue's wire format (at most 256 n-gram blocks, i.e. max_l_dist "
                        "< 256, 32-bit lengths, 16-bit distances): use the native collective (fz_comm_*), which has no such limit",
                        (unsigned long long)i);
        ++h->per_block[b];
        nb = std::max(nb, (uint32_t)b + 1);
        if (i < fit) rows[i] = WireRow{in[i].start, (uint32_t)(in[i].end - in[i].start), (uint16_t)in[i].dist, (uint16_t)b};
    }
    h->nblocks = nb;
    return FZ_OK;
}

int fz_wire_merge(const void *recv, uint32_t world, uint64_t rows_per_rank, uint64_t cap_rows,
                  fz_match *out, uint64_t out_cap, uint64_t *n_out, uint64_t *max_count) {
    if (!recv || !n_out || !max_count || rows_per_rank < FZ_WIRE_HEADER_ROWS + cap_rows) return fail(FZ_EINVAL, "bad argument");
    const uint8_t *base = static_cast<const uint8_t *>(recv);
    uint64_t total = 0, top = 0;
    uint32_t nb = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const WireHeader *h = reinterpret_cast<const WireHeader *>(base + (size_t)r * rows_per_rank * 16);
        if (h->nblocks > 256) return fail(FZ_EINVAL, "rank %u: corrupt header", r);
        total += h->count;
        top = std::max(top, h->count);
        nb = std::max(nb, (uint32_t)h->nblocks);
    }
    *n_out = total;
    *max_count = top;
    if (top > cap_rows) return FZ_OK;                          // caller re-gathers with a larger capacity
    if (total > out_cap || (total && !out)) return fail(FZ_EINVAL, "output too small for %llu records", (unsigned long long)total);
    std::vector<uint64_t> pos(world, 0);
    uint64_t o = 0;
    for (uint32_t g = 0; g < nb; ++g)
        for (uint32_t r = 0; r < world; ++r) {
            const WireHeader *h = reinterpret_cast<const WireHeader *>(base + (size_t)r * rows_per_rank * 16);
            const WireRow *rows = reinterpret_cast<const WireRow *>(h + 1) + pos[r];
            const uint32_t c = h->per_block[g];
            if (pos[r] + c > h->count) return fail(FZ_EINVAL, "rank %u: block counts exceed its record count", r);
            for (uint32_t i = 0; i < c; ++i, ++o) {
                out[o].start = rows[i].start;
                out[o].end = rows[i].start + (int64_t)rows[i].len;
                out[o].dist = rows[i].dist;
                out[o].block = rows[i].block;
            }
            pos[r] += c;
        }
    return FZ_OK;
}

int fz_debug_order_records(const void *recs, uint64_t n, uint32_t L, fz_match **out, uint64_t *n_out) {
    if ((!recs && n) || !out || !n_out) return fail(FZ_EINVAL, "null argument");
    static_assert(sizeof(FzRec) == 24, "record layout");
    return emit_matches(static_cast<const FzRec *>(recs), (size_t)n, L, out, n_out);
}

int fz_debug_order_records_bounded(const void *recs, uint64_t n, uint32_t L, uint64_t idx_bound, uint32_t blk_bound,
                                   fz_match **out, uint64_t *n_out) {
    if ((!recs && n) || !out || !n_out || !idx_bound || !blk_bound) return fail(FZ_EINVAL, "null argument");
    return emit_matches(static_cast<const FzRec *>(recs), (size_t)n, L, out, n_out, idx_bound, blk_bound, /*may_have_empty=*/true);
}

int fz_debug_order_segments(const void *recs, const uint64_t *seg_ends, uint32_t n_segments, uint32_t L, fz_match **out, uint64_t *n_out) {
    if ((!recs && n_segments) || !seg_ends || !out || !n_out || n_segments == 0) return fail(FZ_EINVAL, "null argument");
    std::vector<size_t> ends(seg_ends, seg_ends + n_segments);
    std::vector<uint32_t> order(n_segments);
    for (uint32_t i = 0; i < n_segments; ++i) order[i] = i;
    return emit_matches_segments(nullptr, static_cast<const FzRec *>(recs), ends, order, L, out, n_out, 0, 0, true);
}

int fz_debug_launch_plan(const uint8_t *p, uint32_t m, uint32_t L, uint32_t *out, uint32_t cap, uint32_t *n_launches) {
    if (!p || !out || !n_launches || L == 0 || L > m) return fail(FZ_EINVAL, "bad argument");
    std::vector<uint32_t> starts;
    for (uint32_t s = 0; s + L <= m; s += L) starts.push_back(s);
    const uint32_t G = (uint32_t)starts.size();
    uint32_t nl = 0;
    for (uint32_t g0 = 0; g0 < G;) {
        uint32_t hk = 0, sh = 0;
        const uint32_t nb = choose_launch_blocks(p, starts.data(), g0, G, L, hk, sh);
        if (nb == 0) return fail(FZ_EDEVICE, "internal: a launch without blocks");
        if (nl < cap) { out[4 * nl] = g0; out[4 * nl + 1] = nb; out[4 * nl + 2] = hk; out[4 * nl + 3] = sh; }
        ++nl;
        g0 += nb;
    }
    *n_launches = nl;
    return FZ_OK;
}

// Faithful group-list-order version (common.py:161-177), needed by the substitutions-only path: a match joins
// the group(s) whose hull it overlaps; one group is extended in place, several are removed and their union is
// appended at the END of the list; the result is in final list order.  The reference tests every group for
// every match (O(M * groups): 2 ms for the 2e3 raw matches of configs[2], 50 ms for 2e4).  Group hulls never
// overlap each other (a match that would make two hulls overlap overlaps both groups and merges them; a
// zero-length group is a point that is never strictly inside another hull), so the groups are kept ordered by
// (hull start, hull end) and the groups a match overlaps are a contiguous run found by one search; the list
// position of a group is a sequence number handed out on creation and on merges.
static int group_best_exact(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out);

// Fast form for what the searches produce (round 4: the std::map walk below cost 0.39 ms on the 2.8e3 rows of configs[2],
// more than the search it follows): thousands of matches in a few hundred to a few thousand SMALL groups.  The partition
// into groups is order independent (connected components of the overlap relation: sort + sweep, as fz_consolidate);
// what depends on the input order is only WHERE in the list a group ends up, and that is decided inside the group: a
// group's position is the moment (input index) of the last event that gave it a new list entry — its creation, or the
// last merge of two or more groups into one (the union is appended at the end, common.py:169-175); extensions in place
// keep the entry.  So every component replays its own members in input order on a handful of sub-groups, and the
// components are emitted in the order of those moments.  Components with many members (dense repeats) or anything the
// sweep and the replay disagree on take the exact walk.
int fz_group_best(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out) {
    if (!out || !n_out || (!in && n)) return fail(FZ_EINVAL, "null argument");
    *out = nullptr; *n_out = 0;
    const bool no_fast = sw().group_best_exact;                                // test knob
    if (n < 32 || n > 0x7fffffffull || no_fast) return group_best_exact(in, n, out, n_out);
    auto better = [](const fz_match &a, const fz_match &b) {
        const int64_t la = a.end - a.start, lb = b.end - b.start;
        return a.dist < b.dist || (a.dist == b.dist && (la > lb || (la == lb && a.start < b.start)));
    };
    static thread_local std::vector<uint32_t> order, cnt, comp, first, members, slot;
    // (1) sweep order = (start, zero-length first, input order): one counting pass over ~n slices of the start range,
    //     then the few crowded slices are fixed (as consolidate_hulls does)
    order.resize(n);
    int64_t smin = in[0].start, smax = smin;
    for (uint64_t i = 0; i < n; ++i) { smin = std::min(smin, in[i].start); smax = std::max(smax, in[i].start); }
    uint64_t nb = 1;
    while (nb < n && nb < (1u << 24)) nb <<= 1;
    const uint64_t range = (uint64_t)(smax - smin);
    int shift = 0;
    while ((range >> shift) >= nb) ++shift;
    cnt.assign(nb + 1, 0u);
    for (uint64_t i = 0; i < n; ++i) ++cnt[(((uint64_t)(in[i].start - smin)) >> shift) + 1];
    for (uint64_t b = 1; b <= nb; ++b) cnt[b] += cnt[b - 1];
    for (uint64_t i = 0; i < n; ++i) order[cnt[((uint64_t)(in[i].start - smin)) >> shift]++] = (uint32_t)i;
    auto before = [&](uint32_t x, uint32_t y) {
        const fz_match &a = in[x], &b = in[y];
        if (a.start != b.start) return a.start < b.start;
        const bool pa = a.end == a.start, pb = b.end == b.start;
        if (pa != pb) return pa;
        return x < y;
    };
    // (2) components
    comp.resize(n);
    uint32_t ncomp = 0, lo = 0, biggest = 0, cur = 0;
    int64_t h0 = 0, h1 = 0;
    for (uint64_t b = 0; b < nb; ++b) {
        const uint32_t hi = cnt[b];
        if (hi - lo > 1) {
            if (hi - lo > 24) std::sort(order.begin() + lo, order.begin() + hi, before);
            else
                for (uint32_t i = lo + 1; i < hi; ++i) {
                    const uint32_t v = order[i];
                    uint32_t j = i;
                    while (j > lo && before(v, order[j - 1])) { order[j] = order[j - 1]; --j; }
                    order[j] = v;
                }
        }
        for (uint32_t vi = lo; vi < hi; ++vi) {
            const fz_match &mt = in[order[vi]];
            if (ncomp && !(mt.end <= h0 || mt.start >= h1)) {
                h0 = std::min(h0, mt.start);
                h1 = std::max(h1, mt.end);
                biggest = std::max(biggest, ++cur);
            } else {
                h0 = mt.start; h1 = mt.end;
                ++ncomp;
                cur = 1;
            }
            comp[order[vi]] = ncomp - 1;
        }
        lo = hi;
    }
    if (biggest > 256) return group_best_exact(in, n, out, n_out);
    // (3) members of every component in input order
    first.assign((size_t)ncomp + 1, 0u);
    for (uint64_t i = 0; i < n; ++i) ++first[comp[i] + 1];
    for (uint32_t c = 0; c < ncomp; ++c) first[c + 1] += first[c];
    members.resize(n);
    {
        static thread_local std::vector<uint32_t> fill;
        fill.assign(first.begin(), first.end() - 1);
        for (uint64_t i = 0; i < n; ++i) members[fill[comp[i]]++] = (uint32_t)i;
    }
    // (4) replay every component; slot[moment] = component + 1
    slot.assign(n, 0u);
    struct Sub { int64_t s, e; uint32_t when; };
    Sub subs[260];
    for (uint32_t c = 0; c < ncomp; ++c) {
        const uint32_t b0 = first[c], b1 = first[c + 1];
        if (b1 - b0 == 1) { slot[members[b0]] = c + 1; continue; }
        uint32_t ns = 0;
        for (uint32_t q = b0; q < b1; ++q) {
            const uint32_t i = members[q];
            const fz_match &mt = in[i];
            uint32_t nov = 0, firstov = 0;
            for (uint32_t g = 0; g < ns; ++g)
                if (!(mt.end <= subs[g].s || mt.start >= subs[g].e)) { if (!nov) firstov = g; ++nov; }
            if (nov == 0) {
                subs[ns++] = Sub{mt.start, mt.end, i};
            } else if (nov == 1) {
                subs[firstov].s = std::min(subs[firstov].s, mt.start);
                subs[firstov].e = std::max(subs[firstov].e, mt.end);
            } else {
                Sub u{mt.start, mt.end, i};
                uint32_t w = 0;
                for (uint32_t g = 0; g < ns; ++g) {
                    if (!(mt.end <= subs[g].s || mt.start >= subs[g].e)) {
                        u.s = std::min(u.s, subs[g].s);
                        u.e = std::max(u.e, subs[g].e);
                    } else {
                        subs[w++] = subs[g];
                    }
                }
                subs[w++] = u;
                ns = w;
            }
        }
        if (ns != 1) return group_best_exact(in, n, out, n_out);     // the sweep saw one component, the replay did not
        slot[subs[0].when] = c + 1;
    }
    void *mem = nullptr;
    int rc = alloc_out(ncomp, sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *o = static_cast<fz_match *>(mem);
    uint64_t w = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (!slot[i]) continue;
        const uint32_t c = slot[i] - 1;
        fz_match best = in[members[first[c]]];
        for (uint32_t q = first[c] + 1; q < first[c + 1]; ++q)
            if (better(in[members[q]], best)) best = in[members[q]];
        o[w++] = best;
    }
    *out = o;
    *n_out = w;
    return FZ_OK;
}

static int group_best_exact(const fz_match *in, uint64_t n, fz_match **out, uint64_t *n_out) {
    struct Key {
        int64_t s, e; uint64_t seq;
        bool operator<(const Key &o) const { return s != o.s ? s < o.s : (e != o.e ? e < o.e : seq < o.seq); }
    };
    auto better = [](const fz_match &a, const fz_match &b) {
        const int64_t la = a.end - a.start, lb = b.end - b.start;
        return a.dist < b.dist || (a.dist == b.dist && (la > lb || (la == lb && a.start < b.start)));
    };
    std::map<Key, fz_match> groups;                            // key: hull + list position, value: best match
    uint64_t next_seq = 0;
    std::vector<std::map<Key, fz_match>::iterator> ov;
    for (uint64_t i = 0; i < n; ++i) {
        const fz_match &mt = in[i];
        auto overlaps = [&](const Key &g) { return !(mt.end <= g.s || mt.start >= g.e); };
        ov.clear();
        auto it = groups.lower_bound(Key{mt.start, INT64_MIN, 0});     // first group starting at or after the match
        // before it: points cannot overlap (they lie left of the match's start), and of the groups with a length
        // only the nearest one can reach across the match's start
        for (auto pb = it; pb != groups.begin();) {
            --pb;
            if (pb->first.e == pb->first.s) continue;
            if (overlaps(pb->first)) ov.push_back(pb);
            break;
        }
        for (; it != groups.end() && it->first.s < mt.end; ++it)
            if (overlaps(it->first)) ov.push_back(it);
        if (ov.empty()) {
            groups.emplace(Key{mt.start, mt.end, next_seq++}, mt);
        } else if (ov.size() == 1 && mt.start >= ov[0]->first.s && mt.end <= ov[0]->first.e) {
            if (better(mt, ov[0]->second)) ov[0]->second = mt;   // inside the hull (cross-block duplicates): nothing moves
        } else if (ov.size() == 1) {                           // extended in place: same list position
            auto node = groups.extract(ov[0]);
            node.key().s = std::min(node.key().s, mt.start);
            node.key().e = std::max(node.key().e, mt.end);
            if (better(mt, node.mapped())) node.mapped() = mt;
            groups.insert(std::move(node));
        } else {                                               // union appended at the end of the list
            std::sort(ov.begin(), ov.end(), [](const auto &x, const auto &y) { return x->first.seq < y->first.seq; });
            Key u{mt.start, mt.end, 0};
            fz_match best = mt;
            for (auto g : ov) {
                u.s = std::min(u.s, g->first.s);
                u.e = std::max(u.e, g->first.e);
                if (better(g->second, best)) best = g->second;
            }
            for (auto g : ov) groups.erase(g);
            u.seq = next_seq++;
            groups.emplace(u, best);
        }
    }
    std::vector<std::pair<uint64_t, fz_match>> ordered;
    ordered.reserve(groups.size());
    for (const auto &g : groups) ordered.emplace_back(g.first.seq, g.second);
    std::sort(ordered.begin(), ordered.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    void *mem = nullptr;
    int rc = alloc_out(ordered.size(), sizeof(fz_match), &mem);
    if (rc) return rc;
    fz_match *o = static_cast<fz_match *>(mem);
    for (size_t g = 0; g < ordered.size(); ++g) o[g] = ordered[g].second;
    *out = o;
    *n_out = ordered.size();
    return FZ_OK;
}

int fz_set_timing(fz_ctx *ctx, int on) {
    if (!ctx) return fail(FZ_EINVAL, "null argument");
    if (ctx->npend || ctx->stream_inflight) return fail(FZ_EINVAL, "a search of this context is in flight");
    ctx->timing = on != 0;
    return FZ_OK;
}

int fz_set_streams(fz_ctx *ctx, int n) {
    if (!ctx || (n != 1 && n != 2)) return fail(FZ_EINVAL, "streams must be 1 or 2");
    if (ctx->npend || ctx->stream_inflight) return fail(FZ_EINVAL, "a search of this context is in flight");
    ctx->streams = n;
    return FZ_OK;
}

int fz_stats(fz_ctx *ctx, fz_stats_t *out) {
    if (!ctx || !out) return fail(FZ_EINVAL, "null argument");
    resolve_timing(ctx);
    *out = ctx->stats;
    return FZ_OK;
}

int fz_mem_info(fz_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return fail(FZ_EINVAL, "null argument");
    uint64_t fmin = ~0ull, tmin = ~0ull;
    for (const DevState &d : ctx->devs) {
        HIP_TRY(hipSetDevice(d.device));
        size_t f = 0, t = 0;
        HIP_TRY(hipMemGetInfo(&f, &t));
        fmin = std::min<uint64_t>(fmin, f);
        tmin = std::min<uint64_t>(tmin, t);
    }
    *free_bytes = fmin;
    *total_bytes = tmin;
    return FZ_OK;
}

void fz_free(void *p) { release_out(p); }


}  // extern "C"
