// TEST INFRASTRUCTURE — a stand-in for librccl that lets N ranks live on ONE GPU.
//
// libfzhip.so dlopen-s RCCL through a nine-entry table (fuzzysearch_amd/csrc/fzhip.hip: rccl_api) and honours
// FZ_RCCL_LIB; RCCL itself refuses two ranks on one device, and no box of this pool has more than one GPU.  This
// library implements exactly those nine entry points with the SAME stream semantics RCCL gives them, so that the
// product's N-rank code (gather_records, comm_gather_host, comm_rank_lows, the snapshot / hipStreamWaitEvent
// ordering, the grouped all-gather, capacity regrow, the two-deep pipeline) runs with world 2, 3, 8 on one device:
//
//   * ncclCommInitAll(comms, n, devlist): n ranks in ONE process, duplicate devices allowed.  A collective is a
//     ncclGroupStart / n calls / ncclGroupEnd (as RCCL requires of one thread driving several communicators); it is
//     executed as n*n stream-ordered device-to-device copies: rank r's stream waits for an event on every rank j's
//     stream (j's send buffer is ready), copies send_j into its receive block, and every rank's stream then waits for
//     all readers of its send buffer — i.e. the work is asynchronous, ordered only by the streams, exactly as with RCCL.
//     The receive buffer is poisoned (0xEE) on the stream first, so stale bytes of an earlier gather cannot pass.
//   * ncclCommInitRank(comm, n, id, rank): ranks in DIFFERENT processes (one process per "GPU", the launcher form).
//     The 128-byte id names a POSIX shared-memory control block; every rank owns a shared data file that all ranks map
//     and page-lock (hipHostRegister) at init.  An ALL-GATHER is asynchronous, ordered only by the caller's stream,
//     as RCCL's is (round 6; rounds 4-5 synchronised the stream first, which would have hidden a missing dependency
//     between the search stream and the communicator's stream in exactly this form): on the caller's stream a host
//     function waits until every rank has read this rank's previous contribution, an asynchronous D2H copy moves the
//     send buffer into the rank's data file, a host function publishes it and waits for the other ranks' publications,
//     asynchronous H2D copies fill the (poisoned) receive buffer, a host function reports "read".  The caller's thread
//     never waits.  The (8-byte) all-reduce stays blocking: stream synchronise, copy, barrier, reduce.  Every wait
//     has a deadline (FZ_MOCK_RCCL_TIMEOUT_S, default 120 s): a rank that never arrives marks the communicator failed
//     (the next call returns ncclRemoteError), not a hung GPU box.  FZMOCK_SHM_BLOCKING=1: rounds 4-5's blocking form.
//   * misuse that would hang or corrupt with the real library is an ERROR here: a collective with fewer calls than
//     ranks, differing byte counts between ranks, a rank used twice in one group.
//
//   * fault injection (tests/test_gpu_mock_rccl.py: what bench.py --gpus N must survive on its first real node):
//       FZMOCK_FAIL_INIT=1            ncclCommInitAll / ncclCommInitRank return ncclSystemError
//       FZMOCK_HANG_INIT_S=s          ncclCommInitRank blocks for s seconds first (a peer that never joins: the caller's watchdog)
//       FZMOCK_FAIL_ALLGATHER=n       the n-th all-gather of the process (1-based) returns ncclSystemError (one process per rank: on
//                                     every rank, or with FZMOCK_FAIL_RANK=r on rank r alone — its peers then wait for it in vain)
//       FZMOCK_LATE_RANK=r:ms         rank r's send buffer becomes "ready" ms milliseconds late in every in-process all-gather
//                                     (a host function sleeping on its stream): slower, never wrong
//       FZMOCK_STALL_ALLGATHER=n:ms   the n-th all-gather (either form) stalls rank 0's stream for ms milliseconds — a rank that
//                                     (for the caller's deadline) never arrives
// Not emulated: RCCL's transports (xGMI rings, IPC handles, channels) — those stay unverified until an 8-GPU node
// runs bench.py.  The product never loads this file; tests/ and nothing else names it.
//
//   hipcc -O2 -fPIC -shared tests/mock_rccl.cpp -o tests/libmock_rccl.so        (tests/mock_rccl.py: build())
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 64;

thread_local std::string g_err;

ncclResult_t bad(ncclResult_t code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
ncclResult_t bad(ncclResult_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    fprintf(stderr, "[mock_rccl] %s\n", buf);
    return code;
}

#define HIPQ(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return bad(ncclUnhandledCudaError, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

long env_long(const char *name, long dflt) { const char *e = getenv(name); return e ? atol(e) : dflt; }
// "a:b" -> (a, b); (-1, 0) when unset
void env_pair(const char *name, long &a, long &b) {
    a = -1; b = 0;
    const char *e = getenv(name);
    if (!e) return;
    a = atol(e);
    const char *c = strchr(e, ':');
    b = c ? atol(c + 1) : 0;
}
void sleep_on_stream(void *ms) { std::this_thread::sleep_for(std::chrono::milliseconds((long)(intptr_t)ms)); }

double timeout_s() {
    static const double t = []() { const char *e = getenv("FZ_MOCK_RCCL_TIMEOUT_S"); return e ? atof(e) : 120.0; }();
    return t;
}

// ---- shared control block of a multi-process communicator -----------------------------------------------------
struct Ctl {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> left;
    std::atomic<uint32_t> failed;
    std::atomic<uint64_t> size[kMaxRanks];       // bytes rank r published for the collective in progress
    std::atomic<uint64_t> cap[kMaxRanks];        // size of rank r's data file
    // asynchronous all-gathers: rank r has published / has finished reading its pub[r]-th / rd[r]-th one
    std::atomic<uint64_t> pub[kMaxRanks], pub_bytes[kMaxRanks], rd[kMaxRanks];
};

struct Mapping {
    void *p = nullptr;
    uint64_t bytes = 0;
};

struct Clique;                                   // the ranks of one communicator that live in this process

}  // namespace

struct ncclComm {
    Clique *clique = nullptr;
    int world = 0, rank = 0, device = 0;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    // multi-process form
    std::string name;
    Ctl *ctl = nullptr;
    int data_fd = -1;
    Mapping mine;
    std::vector<Mapping> theirs;
    uint32_t barrier_gen = 0;
    bool async_ok = false;                       // every rank's data file is mapped and page-locked: asynchronous all-gathers
    uint64_t async_seq = 0;                      // asynchronous all-gathers this rank has issued
    uint64_t async_cap = 0;
};

namespace {

struct Clique {
    std::vector<ncclComm *> members;             // by rank
    int alive = 0;
};

struct Op {
    int kind;                                    // 0 all-gather, 1 all-reduce
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t dt;
    ncclRedOp_t red;
    ncclComm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::atomic<uint64_t> g_stats[4];                // all-gathers, all-gather bytes (one rank's contribution), all-reduces, largest world
std::atomic<uint64_t> g_async_allgathers{0};     // cross-process all-gathers that took the asynchronous (stream-ordered) form

size_t dt_bytes(ncclDataType_t dt) {
    switch (dt) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

template <typename T>
void reduce_typed(T *acc, const T *v, size_t n, ncclRedOp_t red) {
    for (size_t i = 0; i < n; ++i) {
        if (red == ncclMax) acc[i] = v[i] > acc[i] ? v[i] : acc[i];
        else if (red == ncclMin) acc[i] = v[i] < acc[i] ? v[i] : acc[i];
        else acc[i] = acc[i] + v[i];
    }
}

ncclResult_t reduce_into(void *acc, const void *v, size_t n, ncclDataType_t dt, ncclRedOp_t red) {
    if (red != ncclMax && red != ncclMin && red != ncclSum) return bad(ncclInvalidArgument, "mock: reduction %d is not implemented", (int)red);
    switch (dt) {
        case ncclFloat64: reduce_typed(static_cast<double *>(acc), static_cast<const double *>(v), n, red); break;
        case ncclFloat32: reduce_typed(static_cast<float *>(acc), static_cast<const float *>(v), n, red); break;
        case ncclInt64: reduce_typed(static_cast<int64_t *>(acc), static_cast<const int64_t *>(v), n, red); break;
        case ncclUint64: reduce_typed(static_cast<uint64_t *>(acc), static_cast<const uint64_t *>(v), n, red); break;
        case ncclInt32: reduce_typed(static_cast<int32_t *>(acc), static_cast<const int32_t *>(v), n, red); break;
        case ncclUint32: reduce_typed(static_cast<uint32_t *>(acc), static_cast<const uint32_t *>(v), n, red); break;
        default: return bad(ncclInvalidArgument, "mock: all-reduce of data type %d is not implemented", (int)dt);
    }
    return ncclSuccess;
}

// ---- one process, several ranks: stream-ordered device-to-device copies ---------------------------------------
ncclResult_t ensure_events(ncclComm *c) {
    if (c->ev_ready) return ncclSuccess;
    HIPQ(hipSetDevice(c->device));
    HIPQ(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    HIPQ(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    return ncclSuccess;
}

ncclResult_t local_allgather(Clique *q, const std::vector<const Op *> &ops) {      // ops[rank]
    const int n = (int)q->members.size();
    const size_t bytes = ops[0]->count * dt_bytes(ops[0]->dt);
    for (int r = 0; r < n; ++r) {
        if (ops[r]->count * dt_bytes(ops[r]->dt) != bytes)
            return bad(ncclInvalidArgument, "mock: all-gather with %zu bytes on rank 0 and %zu on rank %d (RCCL would corrupt or hang)",
                       bytes, ops[r]->count * dt_bytes(ops[r]->dt), r);
        ncclResult_t rc = ensure_events(q->members[r]);
        if (rc != ncclSuccess) return rc;
    }
    const uint64_t call_no = ++g_stats[0];
    g_stats[1] += bytes;
    if ((long)call_no == env_long("FZMOCK_FAIL_ALLGATHER", -1)) return bad(ncclSystemError, "mock: injected failure of all-gather %llu", (unsigned long long)call_no);
    if (!bytes) return ncclSuccess;
    long late_rank, late_ms, stall_no, stall_ms;
    env_pair("FZMOCK_LATE_RANK", late_rank, late_ms);
    env_pair("FZMOCK_STALL_ALLGATHER", stall_no, stall_ms);
    for (int j = 0; j < n; ++j) {                              // "send buffer of rank j is ready" — on j's stream
        ncclComm *c = q->members[j];
        HIPQ(hipSetDevice(c->device));
        if (j == late_rank && late_ms > 0) HIPQ(hipLaunchHostFunc(ops[j]->stream, sleep_on_stream, (void *)(intptr_t)late_ms));
        if (j == 0 && (long)call_no == stall_no && stall_ms > 0) HIPQ(hipLaunchHostFunc(ops[j]->stream, sleep_on_stream, (void *)(intptr_t)stall_ms));
        HIPQ(hipEventRecord(c->ev_ready, ops[j]->stream));
    }
    for (int r = 0; r < n; ++r) {
        ncclComm *c = q->members[r];
        HIPQ(hipSetDevice(c->device));
        uint8_t *recv = static_cast<uint8_t *>(ops[r]->recv);
        const uint8_t *own = static_cast<const uint8_t *>(ops[r]->send);
        const bool in_place = own == recv + (size_t)r * bytes;
        for (int j = 0; j < n; ++j) {                          // poison what is about to be received (not an in-place send block)
            if (in_place && j == r) continue;
            HIPQ(hipMemsetAsync(recv + (size_t)j * bytes, 0xEE, bytes, ops[r]->stream));
        }
        for (int j = 0; j < n; ++j) {
            if (j != r) HIPQ(hipStreamWaitEvent(ops[r]->stream, q->members[j]->ev_ready, 0));
            if (in_place && j == r) continue;
            HIPQ(hipMemcpyAsync(recv + (size_t)j * bytes, ops[j]->send, bytes, hipMemcpyDeviceToDevice, ops[r]->stream));
        }
        HIPQ(hipEventRecord(c->ev_done, ops[r]->stream));
    }
    for (int j = 0; j < n; ++j) {                              // rank j's stream goes on once every reader of its send buffer is done
        HIPQ(hipSetDevice(q->members[j]->device));
        for (int r = 0; r < n; ++r)
            if (r != j) HIPQ(hipStreamWaitEvent(ops[j]->stream, q->members[r]->ev_done, 0));
    }
    return ncclSuccess;
}

ncclResult_t local_allreduce(Clique *q, const std::vector<const Op *> &ops) {      // blocking through the host: tiny payloads only
    const int n = (int)q->members.size();
    const size_t esz = dt_bytes(ops[0]->dt), bytes = ops[0]->count * esz;
    if (!esz) return bad(ncclInvalidArgument, "mock: data type %d", (int)ops[0]->dt);
    g_stats[2]++;
    std::vector<uint8_t> acc(bytes), tmp(bytes);
    for (int r = 0; r < n; ++r) {
        if (ops[r]->count != ops[0]->count || ops[r]->dt != ops[0]->dt || ops[r]->red != ops[0]->red)
            return bad(ncclInvalidArgument, "mock: all-reduce arguments differ between ranks");
        HIPQ(hipSetDevice(q->members[r]->device));
        HIPQ(hipStreamSynchronize(ops[r]->stream));
        HIPQ(hipMemcpy(r ? tmp.data() : acc.data(), ops[r]->send, bytes, hipMemcpyDeviceToHost));
        if (r) { ncclResult_t rc = reduce_into(acc.data(), tmp.data(), ops[0]->count, ops[0]->dt, ops[0]->red); if (rc != ncclSuccess) return rc; }
    }
    for (int r = 0; r < n; ++r) {
        HIPQ(hipSetDevice(q->members[r]->device));
        HIPQ(hipMemcpy(ops[r]->recv, acc.data(), bytes, hipMemcpyHostToDevice));
    }
    return ncclSuccess;
}

bool spin_until(ncclComm *c, const std::atomic<uint64_t> *arr, uint64_t want, const char *what);

// ---- one process per rank: shared memory, blocking ------------------------------------------------------------
ncclResult_t shm_barrier(ncclComm *c) {
    Ctl *ctl = c->ctl;
    const uint32_t gen = ctl->generation.load(std::memory_order_acquire);
    if (ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        ctl->arrived.store(0, std::memory_order_relaxed);
        ctl->generation.fetch_add(1, std::memory_order_acq_rel);
        return ncclSuccess;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    unsigned spins = 0;
    while (ctl->generation.load(std::memory_order_acquire) == gen) {
        if (ctl->failed.load(std::memory_order_acquire)) return bad(ncclRemoteError, "mock: another rank reported a failure");
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 1023) == 0 && std::chrono::steady_clock::now() > deadline) {
            ctl->failed.store(1, std::memory_order_release);
            return bad(ncclSystemError, "mock: rank %d waited %.0f s at a barrier of %d ranks (a rank never made the matching call)",
                       c->rank, timeout_s(), c->world);
        }
    }
    return ncclSuccess;
}

std::string data_name(const std::string &base, int rank) { return base + "_d" + std::to_string(rank); }

ncclResult_t map_file(const std::string &name, uint64_t need, Mapping &m, bool writable) {
    if (m.p && m.bytes >= need) return ncclSuccess;
    if (m.p) { munmap(m.p, m.bytes); m.p = nullptr; m.bytes = 0; }
    int fd = shm_open(name.c_str(), writable ? O_RDWR : O_RDONLY, 0600);
    if (fd < 0) return bad(ncclSystemError, "mock: shm_open(%s): %s", name.c_str(), strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < need) { close(fd); return bad(ncclSystemError, "mock: %s is smaller than %llu bytes", name.c_str(), (unsigned long long)need); }
    void *p = mmap(nullptr, (size_t)st.st_size, writable ? PROT_READ | PROT_WRITE : PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return bad(ncclSystemError, "mock: mmap(%s): %s", name.c_str(), strerror(errno));
    m.p = p; m.bytes = (uint64_t)st.st_size;
    return ncclSuccess;
}

// every rank's `bytes` land in c->theirs[j].p after the first barrier; the caller reads them and calls shm_barrier again
ncclResult_t shm_publish(ncclComm *c, const void *send_dev, size_t bytes, hipStream_t stream) {
    HIPQ(hipSetDevice(c->device));
    HIPQ(hipStreamSynchronize(stream));
    // (the data file is shared with the asynchronous all-gathers: every rank must have read this rank's last one)
    if (c->async_ok && !spin_until(c, c->ctl->rd, c->async_seq, "finish reading"))
        return bad(ncclRemoteError, "mock: the communicator failed (a rank never finished an asynchronous all-gather)");
    if (c->async_ok && bytes > c->async_cap)
        return bad(ncclInvalidArgument, "mock: %zu bytes per rank exceed the shared data files (FZ_MOCK_SHM_CAP_MIB)", bytes);
    if (c->mine.bytes < bytes) {
        const uint64_t want = (bytes + (bytes >> 2) + 65535) / 65536 * 65536;
        if (c->mine.p) { munmap(c->mine.p, c->mine.bytes); c->mine.p = nullptr; c->mine.bytes = 0; }
        if (ftruncate(c->data_fd, (off_t)want) != 0) return bad(ncclSystemError, "mock: ftruncate: %s", strerror(errno));
        void *p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, c->data_fd, 0);
        if (p == MAP_FAILED) return bad(ncclSystemError, "mock: mmap: %s", strerror(errno));
        c->mine.p = p; c->mine.bytes = want;
        c->ctl->cap[c->rank].store(want, std::memory_order_release);
    }
    if (bytes) HIPQ(hipMemcpy(c->mine.p, send_dev, bytes, hipMemcpyDeviceToHost));
    c->ctl->size[c->rank].store(bytes, std::memory_order_release);
    ncclResult_t rc = shm_barrier(c);
    if (rc != ncclSuccess) return rc;
    for (int j = 0; j < c->world; ++j) {
        const uint64_t theirs = c->ctl->size[j].load(std::memory_order_acquire);
        if (theirs != bytes) {
            c->ctl->failed.store(1, std::memory_order_release);
            return bad(ncclInvalidArgument, "mock: collective with %zu bytes on rank %d and %llu on rank %d (RCCL would corrupt or hang)",
                       bytes, c->rank, (unsigned long long)theirs, j);
        }
        if (j == c->rank || !bytes) continue;
        rc = map_file(data_name(c->name, j), bytes, c->theirs[j], false);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

const void *shm_block(ncclComm *c, int j) { return j == c->rank ? c->mine.p : c->theirs[j].p; }

ncclResult_t shm_allgather(const Op &op) {
    ncclComm *c = op.comm;
    const size_t bytes = op.count * dt_bytes(op.dt);
    g_stats[0]++; g_stats[1] += bytes;
    ncclResult_t rc = shm_publish(c, op.send, bytes, op.stream);
    if (rc != ncclSuccess) return rc;
    if (bytes) {
        HIPQ(hipMemset(op.recv, 0xEE, bytes * (size_t)c->world));
        for (int j = 0; j < c->world; ++j)
            HIPQ(hipMemcpy(static_cast<uint8_t *>(op.recv) + (size_t)j * bytes, shm_block(c, j), bytes, hipMemcpyHostToDevice));
    }
    return shm_barrier(c);
}

ncclResult_t shm_allreduce(const Op &op) {
    ncclComm *c = op.comm;
    const size_t bytes = op.count * dt_bytes(op.dt);
    if (!dt_bytes(op.dt)) return bad(ncclInvalidArgument, "mock: data type %d", (int)op.dt);
    g_stats[2]++;
    ncclResult_t rc = shm_publish(c, op.send, bytes, op.stream);
    if (rc != ncclSuccess) return rc;
    std::vector<uint8_t> acc(bytes);
    if (bytes) memcpy(acc.data(), shm_block(c, 0), bytes);
    for (int j = 1; j < c->world && bytes; ++j) {
        rc = reduce_into(acc.data(), shm_block(c, j), op.count, op.dt, op.red);
        if (rc != ncclSuccess) return rc;
    }
    if (bytes) HIPQ(hipMemcpy(op.recv, acc.data(), bytes, hipMemcpyHostToDevice));
    return shm_barrier(c);
}

// ---- one process per rank: the asynchronous all-gather -----------------------------------------------------------
struct AsyncStep { ncclComm *c; uint64_t seq; uint64_t bytes; int what; };   // what: 0 wait-for-readers, 1 publish-and-wait, 2 read-done

bool spin_until(ncclComm *c, const std::atomic<uint64_t> *arr, uint64_t want, const char *what) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s());
    for (int j = 0; j < c->world; ++j) {
        unsigned spins = 0;
        while (arr[j].load(std::memory_order_acquire) < want) {
            if (c->ctl->failed.load(std::memory_order_acquire)) return false;
            if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 1023) == 0 && std::chrono::steady_clock::now() > deadline) {
                c->ctl->failed.store(1, std::memory_order_release);
                fprintf(stderr, "[mock_rccl] rank %d waited %.0f s for rank %d to %s all-gather %llu\n", c->rank, timeout_s(), j, what,
                        (unsigned long long)want);
                return false;
            }
        }
    }
    return true;
}

void async_step(void *arg) {                     // runs on the HIP runtime's callback thread, in stream order; no HIP calls here
    AsyncStep *st = static_cast<AsyncStep *>(arg);
    ncclComm *c = st->c;
    Ctl *ctl = c->ctl;
    if (st->what == 0) {
        (void)spin_until(c, ctl->rd, st->seq - 1, "finish reading");
    } else if (st->what == 1) {
        ctl->pub_bytes[c->rank].store(st->bytes, std::memory_order_release);
        ctl->pub[c->rank].store(st->seq, std::memory_order_release);
        if (spin_until(c, ctl->pub, st->seq, "publish"))
            for (int j = 0; j < c->world; ++j)
                if (ctl->pub_bytes[j].load(std::memory_order_acquire) != st->bytes) {
                    ctl->failed.store(1, std::memory_order_release);
                    fprintf(stderr, "[mock_rccl] all-gather %llu with %llu bytes on rank %d and %llu on rank %d (RCCL would corrupt or hang)\n",
                            (unsigned long long)st->seq, (unsigned long long)st->bytes, c->rank,
                            (unsigned long long)ctl->pub_bytes[j].load(), j);
                }
    } else {
        ctl->rd[c->rank].store(st->seq, std::memory_order_release);
    }
    delete st;
}

ncclResult_t shm_allgather_async(const Op &op) {
    ncclComm *c = op.comm;
    const size_t bytes = op.count * dt_bytes(op.dt);
    if (c->ctl->failed.load(std::memory_order_acquire)) return bad(ncclRemoteError, "mock: the communicator failed earlier (a rank never arrived, or ranks disagreed on a size)");
    const uint64_t call_no = ++g_stats[0];
    g_stats[1] += bytes;
    // fault injection, launcher form: the n-th all-gather of every process — or of rank FZMOCK_FAIL_RANK alone, whose peers then
    // wait for a rank that never comes — returns an error; rank 0's stream stalls in front of its n-th all-gather
    const long fail_rank = env_long("FZMOCK_FAIL_RANK", -1);
    if ((long)call_no == env_long("FZMOCK_FAIL_ALLGATHER", -1) && (fail_rank < 0 || fail_rank == c->rank))
        return bad(ncclSystemError, "mock: injected failure of all-gather %llu on rank %d", (unsigned long long)call_no, c->rank);
    long stall_no, stall_ms;
    env_pair("FZMOCK_STALL_ALLGATHER", stall_no, stall_ms);
    g_async_allgathers++;
    const uint64_t seq = ++c->async_seq;
    HIPQ(hipSetDevice(c->device));
    if (c->rank == 0 && (long)call_no == stall_no && stall_ms > 0) HIPQ(hipLaunchHostFunc(op.stream, sleep_on_stream, (void *)(intptr_t)stall_ms));
    uint8_t *recv = static_cast<uint8_t *>(op.recv);
    HIPQ(hipLaunchHostFunc(op.stream, async_step, new AsyncStep{c, seq, bytes, 0}));
    if (bytes) HIPQ(hipMemcpyAsync(c->mine.p, op.send, bytes, hipMemcpyDeviceToHost, op.stream));
    HIPQ(hipLaunchHostFunc(op.stream, async_step, new AsyncStep{c, seq, bytes, 1}));
    if (bytes) {
        HIPQ(hipMemsetAsync(recv, 0xEE, bytes * (size_t)c->world, op.stream));
        for (int j = 0; j < c->world; ++j)
            HIPQ(hipMemcpyAsync(recv + (size_t)j * bytes, shm_block(c, j), bytes, hipMemcpyHostToDevice, op.stream));
    }
    HIPQ(hipLaunchHostFunc(op.stream, async_step, new AsyncStep{c, seq, bytes, 2}));
    return ncclSuccess;
}

// ---- the group machinery ---------------------------------------------------------------------------------------
ncclResult_t run_ops(std::vector<Op> &ops) {
    // multi-process communicators: every op blocks on its own, in call order (all ranks make the same calls in the same order)
    // one-process cliques: the i-th op of every member forms the i-th collective of the group
    std::vector<Clique *> cliques;
    for (const Op &op : ops) {
        if (op.comm->ctl) continue;
        bool seen = false;
        for (Clique *q : cliques) seen |= q == op.comm->clique;
        if (!seen) cliques.push_back(op.comm->clique);
    }
    for (Clique *q : cliques) {
        const int n = (int)q->members.size();
        std::vector<std::vector<const Op *>> per_rank(n);
        for (const Op &op : ops)
            if (!op.comm->ctl && op.comm->clique == q) per_rank[op.comm->rank].push_back(&op);
        const size_t rounds = per_rank[0].size();
        for (int r = 0; r < n; ++r)
            if (per_rank[r].size() != rounds)
                return bad(ncclInvalidUsage, "mock: rank 0 made %zu collective calls in this group and rank %d made %zu: with RCCL the "
                           "communicator of %d ranks would hang (every rank of a one-process communicator must call inside ONE ncclGroupStart/End)",
                           rounds, r, per_rank[r].size(), n);
        for (size_t i = 0; i < rounds; ++i) {
            std::vector<const Op *> round(n);
            for (int r = 0; r < n; ++r) {
                round[r] = per_rank[r][i];
                if (round[r]->kind != per_rank[0][i]->kind) return bad(ncclInvalidUsage, "mock: ranks disagree on the kind of collective %zu", i);
            }
            ncclResult_t rc = round[0]->kind == 0 ? local_allgather(q, round) : local_allreduce(q, round);
            if (rc != ncclSuccess) return rc;
        }
    }
    for (const Op &op : ops) {
        if (!op.comm->ctl) continue;
        if (op.kind == 0 && op.comm->async_ok && op.count * dt_bytes(op.dt) > op.comm->async_cap)
            return bad(ncclInvalidArgument, "mock: %zu bytes per rank exceed the shared data files (FZ_MOCK_SHM_CAP_MIB)", op.count * dt_bytes(op.dt));
        const bool async = op.kind == 0 && op.comm->async_ok;
        ncclResult_t rc = async ? shm_allgather_async(op) : op.kind == 0 ? shm_allgather(op) : shm_allreduce(op);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

ncclResult_t submit(const Op &op) {
    if (!op.comm) return bad(ncclInvalidArgument, "mock: null communicator");
    uint64_t w = g_stats[3].load();
    while ((uint64_t)op.comm->world > w && !g_stats[3].compare_exchange_weak(w, (uint64_t)op.comm->world)) {}
    g_ops.push_back(op);
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

std::atomic<uint32_t> g_id_counter{0};

}  // namespace

extern "C" {

// fzhip.hip looks this symbol up: with the stand-in loaded, fz_comm_init_all accepts a device listed more than once
int fzmock_rccl = 1;

void fzmock_rccl_stats(uint64_t out[4]) { for (int i = 0; i < 4; ++i) out[i] = g_stats[i].load(); }
uint64_t fzmock_rccl_async_allgathers(void) { return g_async_allgathers.load(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return bad(ncclInvalidArgument, "mock: null id");
    memset(id->internal, 0, sizeof id->internal);
    const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id->internal, sizeof id->internal, "/fzmock_%d_%u_%llx", (int)getpid(), g_id_counter.fetch_add(1), (unsigned long long)t);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev < 1 || ndev > kMaxRanks) return bad(ncclInvalidArgument, "mock: ncclCommInitAll(%d)", ndev);
    if (env_long("FZMOCK_FAIL_INIT", 0)) return bad(ncclSystemError, "mock: injected failure of ncclCommInitAll");
    Clique *q = new Clique;
    q->members.resize(ndev);
    q->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        ncclComm *c = new ncclComm;
        c->clique = q; c->world = ndev; c->rank = r; c->device = devlist ? devlist[r] : r;
        q->members[r] = c;
        comms[r] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return bad(ncclInvalidArgument, "mock: ncclCommInitRank(%d, %d)", nranks, rank);
    if (env_long("FZMOCK_FAIL_INIT", 0)) return bad(ncclSystemError, "mock: injected failure of ncclCommInitRank");
    if (const long hang_s = env_long("FZMOCK_HANG_INIT_S", 0)) std::this_thread::sleep_for(std::chrono::seconds(hang_s));   // a peer that never joins
    int dev = 0;
    HIPQ(hipGetDevice(&dev));
    ncclComm *c = new ncclComm;
    c->world = nranks; c->rank = rank; c->device = dev;
    if (nranks == 1) {                                          // a communicator of one: nothing to share
        Clique *q = new Clique;
        q->members.push_back(c); q->alive = 1;
        c->clique = q;
        *comm = c;
        return ncclSuccess;
    }
    id.internal[sizeof id.internal - 1] = 0;
    c->name = id.internal;
    if (c->name.size() < 8 || c->name[0] != '/') { delete c; return bad(ncclInvalidArgument, "mock: the unique id was not made by this library"); }
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return bad(ncclSystemError, "mock: shm_open(%s): %s", c->name.c_str(), strerror(errno)); }
    if (ftruncate(fd, sizeof(Ctl)) != 0) { close(fd); delete c; return bad(ncclSystemError, "mock: ftruncate: %s", strerror(errno)); }
    void *p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return bad(ncclSystemError, "mock: mmap: %s", strerror(errno)); }
    c->ctl = static_cast<Ctl *>(p);                             // a fresh shm object is zero-filled: every counter starts at 0
    c->theirs.resize(nranks);
    c->data_fd = shm_open(data_name(c->name, rank).c_str(), O_CREAT | O_RDWR, 0600);
    if (c->data_fd < 0) return bad(ncclSystemError, "mock: shm_open(data): %s", strerror(errno));
    c->ctl->joined.fetch_add(1);
    *comm = c;
    // the data files of the asynchronous all-gather: a fixed capacity (FZ_MOCK_SHM_CAP_MIB, 32), created before the init
    // barrier, mapped and page-locked by every rank behind it (a failure anywhere leaves the blocking form)
    const bool want_async = !getenv("FZMOCK_SHM_BLOCKING");
    const uint64_t cap = (uint64_t)std::max(1L, env_long("FZ_MOCK_SHM_CAP_MIB", 32)) << 20;
    bool mine_ok = false;
    if (want_async && ftruncate(c->data_fd, (off_t)cap) == 0) {
        void *mp = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, c->data_fd, 0);
        if (mp != MAP_FAILED) { c->mine.p = mp; c->mine.bytes = cap; c->ctl->cap[rank].store(cap, std::memory_order_release); mine_ok = true; }
    }
    ncclResult_t rc = shm_barrier(c);                           // RCCL's init is collective as well
    if (rc != ncclSuccess) return rc;
    bool ok = want_async && mine_ok;
    for (int j = 0; j < nranks && ok; ++j) {
        if (j == rank) continue;
        ok = c->ctl->cap[j].load(std::memory_order_acquire) == cap && map_file(data_name(c->name, j), cap, c->theirs[j], true) == ncclSuccess;
    }
    for (int j = 0; j < nranks && ok; ++j) {
        void *q = j == rank ? c->mine.p : c->theirs[j].p;
        if (hipHostRegister(q, cap, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); ok = false; }
    }
    if (!ok) c->ctl->failed.load();                             // (nothing to do: the blocking form serves)
    // every rank must take the same form: the slowest common denominator, agreed through one more barrier
    if (!ok) c->ctl->size[rank].store(~0ull, std::memory_order_release); else c->ctl->size[rank].store(1, std::memory_order_release);
    rc = shm_barrier(c);
    if (rc != ncclSuccess) return rc;
    for (int j = 0; j < nranks; ++j) ok = ok && c->ctl->size[j].load(std::memory_order_acquire) == 1;
    rc = shm_barrier(c);                                        // (sizes are reused by the blocking collectives: read them first)
    if (rc != ncclSuccess) return rc;
    c->async_ok = ok;
    c->async_cap = cap;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->ev_ready) { (void)hipEventDestroy(c->ev_ready); (void)hipEventDestroy(c->ev_done); }
    if (c->ctl) {
        if (c->async_ok) {
            if (c->mine.p) (void)hipHostUnregister(c->mine.p);
            for (Mapping &m : c->theirs) if (m.p) (void)hipHostUnregister(m.p);
        }
        if (c->mine.p) munmap(c->mine.p, c->mine.bytes);
        for (Mapping &m : c->theirs) if (m.p) munmap(m.p, m.bytes);
        if (c->data_fd >= 0) close(c->data_fd);
        shm_unlink(data_name(c->name, c->rank).c_str());
        if (c->ctl->left.fetch_add(1) + 1 == (uint32_t)c->world) shm_unlink(c->name.c_str());
        munmap(c->ctl, sizeof(Ctl));
    } else if (c->clique && --c->clique->alive == 0) {
        delete c->clique;
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    if (!dt_bytes(dt)) return bad(ncclInvalidArgument, "mock: data type %d", (int)dt);
    return submit(Op{0, send, recv, count, dt, ncclSum, comm, stream});
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t dt, ncclRedOp_t red, ncclComm_t comm, hipStream_t stream) {
    return submit(Op{1, send, recv, count, dt, red, comm, stream});
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return bad(ncclInvalidUsage, "mock: ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

const char *ncclGetErrorString(ncclResult_t r) {
    static thread_local std::string s;
    const char *base = r == ncclSuccess ? "no error" : r == ncclUnhandledCudaError ? "unhandled HIP error" : r == ncclSystemError ? "system error"
                     : r == ncclInvalidArgument ? "invalid argument" : r == ncclInvalidUsage ? "invalid usage" : r == ncclRemoteError ? "remote error" : "error";
    s = std::string(base) + (g_err.empty() ? "" : " (" + g_err + ")");
    return s.c_str();
}

}  // extern "C"
