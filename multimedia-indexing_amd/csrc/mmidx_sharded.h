// mmidx_sharded.h -- the multi-GPU index behind the C ABI: mmidx_create_sharded and everything a sharded handle does.
// Included by mmidx_api.hip (one translation unit: it uses the handle's internals).
//
// The reference's caller is ONE JVM holding the whole index (YFCC100MExample.java:93-99, query at :155), so the multi-GPU
// engine lives in ONE process: a sharded handle owns one sub-index per shard (whole inverted lists, list c on shard
// c mod n_shards; codebooks replicated), one host worker thread per shard (each drives its own device and stream), and an
// RCCL communicator per device (ncclCommInitAll).  SURVEY section 8e / DESIGN.md section 6.
//
// One search round of B = n_shards x per queries (shard r OWNS queries [r per, (r+1) per): it receives their vectors and
// gets their answers), per worker r:
//   0. own query slice -> Qall[r per ...]; ALL-GATHER (in place) of the query vectors       RCCL   B D 8 bytes
//   1. coarse top-w of the own slice -> cells / cdist[r per ...]; ALL-GATHER of both        RCCL   B w 12 bytes
//   2. pass A over the local lists (probe rank 0) -> thresholds T[B]; ALL-REDUCE MIN         RCCL   B 8 bytes
//   3. pass B under the global thresholds; K4 stores every query's sorted partial list straight into its OWNER's
//      receive buffers [n_shards][per][k+1] -- stores over xGMI (peer access), only the valid entries travel, no sizes
//      to agree on and no host synchronisation ("shard_exchange" = 1: dense local lists + ncclSend / ncclRecv instead)
//   4. events: the owner's stream waits for every shard's pass B; K5 merges the n_shards lists of the owned queries and
//      counts the queries whose k-th and (k+1)-th distances tie
//   5. one 4-byte read-back per shard; if any shard flagged a query: the cross-shard replay of the bounded queue
//      (k_shard_tie: three passes over the local lists, ALL-REDUCE SUM / SUM / MAX in between), in rounds of
//      `tie_slots` flagged queries per owner until every one is handled
// Devices that are not pairwise distinct (virtual shards on one GPU: the test configuration of a one-GPU box -- RCCL
// refuses duplicate devices) run the same steps with in-process collectives: peer copies and a small reduction kernel
// between host barriers.  That path is functional, not a measurement.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <functional>
#include <thread>

namespace {

// ---- RCCL, loaded when the first sharded handle is created (librccl is 0.5 GB: plain handles never map it) -------------
struct RcclApi {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
};

RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    static bool ok = false;
    static std::string why;
    std::call_once(once, [] {
        // (a process that already loaded a librccl.so.1 -- PyTorch bundles one -- gets that copy: same SONAME)
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.so) break;
        }
        if (!api.so) {
            const char *e = dlerror();
            why = e ? e : "dlopen failed";
            return;
        }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(api.so, n);
            if (!p && why.empty()) why = std::string("missing symbol ") + n;
            return p;
        };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        ok = why.empty();
    });
    if (!ok) {
        fail(MMIDX_ERR_UNSUPPORTED, "RCCL is not available (%s): a sharded handle over distinct devices needs librccl.so.1", why.c_str());
        return nullptr;
    }
    return &api;
}

#define NCCLCK(expr)                                                                                          \
    do {                                                                                                      \
        ncclResult_t r__ = (expr);                                                                            \
        if (r__ != ncclSuccess)                                                                               \
            return fail(MMIDX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

// ---- small kernels of the sharded step -----------------------------------------------------------------------------
#define MMIDX_MAX_SHARDS 64
struct PeerPtrs {
    const void *p[MMIDX_MAX_SHARDS];
};

// in-process all-reduce (virtual shards): out[i] = op over the shards' buffers.  OP 0 = min (f64), 1 = sum (i32), 2 = max (i32)
template <typename T, int OP>
__global__ void k_reduce_peers(const PeerPtrs src, int n_src, T *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = ((const T *)src.p[0])[i];
    for (int s = 1; s < n_src; s++) {
        const T v = ((const T *)src.p[s])[i];
        if (OP == 0) acc = v < acc ? v : acc;
        if (OP == 1) acc = acc + v;
        if (OP == 2) acc = v > acc ? v : acc;
    }
    out[i] = acc;
}

// rows [from, to) of X := row 0 (the padding of a round's last slices)
__global__ void k_repeat_row(double *__restrict__ X, int D, long long from, long long to) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (to - from) * D) return;
    X[from * D + i] = X[i % D];
}

__global__ void k_fill_i32(int32_t *__restrict__ p, int32_t v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// The round's one word for the host: the number of flagged queries, then the round's sequence number, stored straight into pinned
// host memory behind everything the stream has done (the host spins on the sequence number: a few microseconds after the last
// kernel instead of a copy + hipStreamSynchronize wake-up)
__global__ void k_publish_nflag(const int32_t *__restrict__ nflag, volatile int32_t *pin, int32_t seq) {
    pin[0] = *nflag;
    __threadfence_system();
    pin[1] = seq;
}

// One wave: the flagged queries of this owner with ordinal in [round Fo, (round + 1) Fo) -> slots.  rows_own[f] = local row
// (-1: unused slot), fq_own[f] = global query index (row + q0; -1), tau_own[f] = the query's k-th distance.
__global__ __launch_bounds__(64) void k_tie_slots(const int32_t *__restrict__ flag, const double *__restrict__ dist, int n_own, int k, int q0,
                                                  int round, int Fo, int32_t *__restrict__ rows_own, int32_t *__restrict__ fq_own,
                                                  double *__restrict__ tau_own) {
    const int lane = threadIdx.x;
    for (int f = lane; f < Fo; f += 64) {
        rows_own[f] = -1;
        fq_own[f] = -1;
        tau_own[f] = 0.0;
    }
    __syncthreads();
    const int lo = round * Fo, hi = lo + Fo;
    int base = 0;
    for (int r0 = 0; r0 < n_own && base < hi; r0 += 64) {
        const int row = r0 + lane;
        const bool fl = row < n_own && flag[row] != 0;
        const u64 mask = __ballot(fl);
        const int ord = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
        if (fl && ord >= lo && ord < hi) {
            rows_own[ord - lo] = row;
            fq_own[ord - lo] = row + q0;
            tau_own[ord - lo] = dist[(size_t)row * k + (k - 1)];
        }
        base += (int)__popcll(mask);
    }
}

// the kept ties of a flagged query overwrite their answer slots (ties_mine[f][s] >= 0)
__global__ void k_tie_patch(const int32_t *__restrict__ rows_own, const int32_t *__restrict__ ties_mine, int Fo, int k,
                            int32_t *__restrict__ iid_out) {
    const int f = blockIdx.x;
    const int row = rows_own[f];
    if (row < 0) return;
    for (int s = threadIdx.x; s < k; s += blockDim.x) {
        const int32_t v = ties_mine[(size_t)f * k + s];
        if (v >= 0) iid_out[(size_t)row * k + s] = v;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// A barrier over the shard workers that a failing worker can abort (the others then leave their waits with `false`).
struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, count = 0;
    uint64_t gen = 0;
    bool aborted = false;
    std::atomic<uint64_t> gen_pub{0};
    std::atomic<bool> aborted_pub{false};
    void reset(int n_) {
        std::lock_guard<std::mutex> lk(mu);
        n = n_;
        count = 0;
        aborted = false;
        aborted_pub.store(false, std::memory_order_release);
    }
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t g = gen;
        if (++count == n) {
            count = 0;
            gen++;
            gen_pub.store(gen, std::memory_order_release);
            cv.notify_all();
            return true;
        }
        lk.unlock();
        for (int spin = 0; spin < 20000 && gen_pub.load(std::memory_order_acquire) == g && !aborted_pub.load(std::memory_order_relaxed); spin++)
            __builtin_ia32_pause();
        lk.lock();
        cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        aborted_pub.store(true, std::memory_order_release);
        cv.notify_all();
    }
};

struct ShardBufs {  // everything on the shard's device
    DevBuf<double> Q, Q2, cdist, T, pd, rpd, odist, tau_own, tau;  // (Q / Q2: the rounds' query buffers alternate)
    DevBuf<long long> pk, rpk;
    DevBuf<int32_t> cells, pc, rpc, oiid, ocnt, flag, nflag, rows_own, fq_own, fq, counts, pB, ties;
    DevBuf<unsigned char> tmp;  // in-process reductions
    DevBuf<ShardDest> dest;     // [n] where pass B's lists go
    DevBuf<double> X;           // add / encode staging
    DevBuf<int32_t> ecell;
    DevBuf<unsigned char> ecode;
    DevBuf<int32_t> rblk, rcell, riid;  // device-side routing of a round's records (sharded_add_vectors): block counts / offsets, the shard's own records
    DevBuf<unsigned char> rcode;
    int32_t *pin_nflag = nullptr;  // pinned host words: [0] flagged queries of the round, [1] the round's sequence number
    int32_t seq = 0;
    ShardDest *pin_dest = nullptr; // pinned host copy of `dest` (what was last uploaded: re-sent only when a pointer changed)
    int dest_n = 0;
    hipEvent_t ev_b = nullptr;     // pass B of this shard has been enqueued up to here
    // the query exchange runs on the shard's SECOND stream (its own communicator), next to the coarse stage of the own slice and,
    // in a call of several rounds, under the previous round's scans.  Per buffer: own slice landed / everybody's landed / the
    // main stream has finished reading it
    hipEvent_t ev_own[2] = {nullptr, nullptr}, ev_q[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    bool free_valid[2] = {false, false};
    void release() {
        for (int i = 0; i < 2; i++) {
            if (ev_own[i]) (void)hipEventDestroy(ev_own[i]);
            if (ev_q[i]) (void)hipEventDestroy(ev_q[i]);
            if (ev_free[i]) (void)hipEventDestroy(ev_free[i]);
            ev_own[i] = ev_q[i] = ev_free[i] = nullptr;
        }
        Q2.release();
        Q.release(); cdist.release(); T.release(); pd.release(); rpd.release(); odist.release(); tau_own.release(); tau.release();
        pk.release(); rpk.release(); cells.release(); pc.release(); rpc.release(); oiid.release(); ocnt.release(); flag.release();
        nflag.release(); rows_own.release(); fq_own.release(); fq.release(); counts.release(); pB.release(); ties.release();
        tmp.release(); dest.release(); X.release(); ecell.release(); ecode.release();
        rblk.release(); rcell.release(); riid.release(); rcode.release();
        if (pin_nflag) (void)hipHostFree(pin_nflag);
        pin_nflag = nullptr;
        if (pin_dest) (void)hipHostFree(pin_dest);
        pin_dest = nullptr;
        if (ev_b) (void)hipEventDestroy(ev_b);
        ev_b = nullptr;
    }
};

}  // namespace

struct ShardGroup {
    int n = 0;
    std::vector<int> dev;
    std::vector<mmidx_index *> sub;
    std::vector<hipStream_t> st;
    std::vector<hipStream_t> st2;   // the query exchange's streams
    std::vector<ncclComm_t> comm2;  // ... and communicators (null: in-process collectives)
    int pipeline = 1;               // option "shard_pipeline": 1 = query exchange on the second stream (overlapped), 0 = everything on one stream
                                    // (default 1 for in-process shards, 0 for two or more physical devices: mmidx_create_sharded)
    bool rccl = false;       // collectives through RCCL (devices pairwise distinct); else in-process (virtual shards)
    bool peer_ok = true;     // every shard can store into every other shard's memory
    int route_host = 0;      // option "shard_route_host": 1 = mmidx_add_vectors_sliced_device routes its records through the host (A/B switch;
                             // default 1 on two or more physical devices until a multi-GPU run has passed: mmidx_create_sharded)
    int exchange = 0;        // option "shard_exchange": 0 = pass B stores into the owners' buffers, 1 = ncclSend / ncclRecv of dense lists
    int tie_slots = 32;      // option "tie_slots": flagged queries replayed per owner and round
    int64_t max_round = 262144;  // option "shard_max_round": queries per collective round over all shards
    std::vector<ncclComm_t> comm;
    std::vector<ShardBufs> buf;
    // what the shards publish to each other between barriers (in-process collectives, peer tables)
    std::vector<void *> pub;
    std::vector<int32_t> nflag_host;
    std::atomic<int64_t> tie_rounds{0}, tie_queries{0};  // statistics: replay rounds run, flagged queries seen
    bool broken = false;     // a worker failed while RCCL collectives may have been in flight: the communicators were aborted
    // worker threads
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<int(int)> job;
    uint64_t job_gen = 0;
    int pending = 0;
    bool quit = false;
    std::atomic<uint64_t> job_gen_pub{0};  // copies of job_gen / pending / quit that the short spins read without the mutex
    std::atomic<int> pending_pub{0};
    std::atomic<bool> quit_pub{false};
    int32_t round_seq = 0;                 // sequence number of the search rounds (k_publish_nflag)
    std::vector<int> rc;
    std::vector<std::string> err;
    HostBarrier bar;
    std::mutex call_mu;  // one sharded call at a time (searches and adds share the workers)
    Combiner comb;       // concurrent mmidx_search callers are served together, as on a plain handle
};

namespace {

void shard_worker(ShardGroup *g, int r) {
    (void)hipSetDevice(g->dev[(size_t)r]);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(int)> f;
        // (a caller that searches in a loop hands over the next job within microseconds: look for it briefly before sleeping --
        //  a condition-variable wake-up costs 10-20 us on each side of every round)
        for (int spin = 0; spin < 20000 && g->job_gen_pub.load(std::memory_order_acquire) == seen && !g->quit_pub.load(std::memory_order_relaxed); spin++)
            __builtin_ia32_pause();
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_go.wait(lk, [&] { return g->quit || g->job_gen != seen; });
            if (g->quit) return;
            seen = g->job_gen;
            f = g->job;
        }
        g_err.clear();
        int rc = f(r);
        if (rc) g->bar.abort();  // (nobody may wait for a worker that has left)
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rc[(size_t)r] = rc;
            g->err[(size_t)r] = rc ? g_err : std::string();
            if (--g->pending == 0) {
                g->pending_pub.store(0, std::memory_order_release);
                g->cv_done.notify_all();
            }
        }
    }
}

// runs f(r) on every shard's worker thread (device current), returns the first failure
int shard_run(ShardGroup *g, std::function<int(int)> f) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->bar.reset(g->n);
    g->job = std::move(f);
    g->pending = g->n;
    g->pending_pub.store(g->n, std::memory_order_release);
    g->job_gen++;
    g->job_gen_pub.store(g->job_gen, std::memory_order_release);
    g->cv_go.notify_all();
    lk.unlock();
    for (int spin = 0; spin < 200000 && g->pending_pub.load(std::memory_order_acquire) != 0; spin++) __builtin_ia32_pause();
    lk.lock();
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
    g->job = nullptr;
    int first = MMIDX_OK;
    std::string msg;
    for (int r = 0; r < g->n; r++) {
        if (!g->rc[(size_t)r]) continue;
        // prefer a real error over "aborted because another shard failed"
        if (!first || (msg.rfind("shard barrier aborted", 0) == 0 && g->err[(size_t)r].rfind("shard barrier aborted", 0) != 0)) {
            first = g->rc[(size_t)r];
            msg = g->err[(size_t)r];
        }
    }
    if (first) g_err = msg.empty() ? "a shard worker failed" : msg;
    return first;
}

// A worker that fails between two collectives leaves its peers' streams inside an RCCL kernel that waits for it: no host barrier
// can release those.  The failing call aborts the communicators (which ends the kernels) and the handle refuses further
// collective calls; mmidx_destroy still works.
void shard_poison(ShardGroup *g) {
    if (!g->rccl || g->broken) return;
    RcclApi *R = rccl_api();
    g->broken = true;
    if (!R || !R->CommAbort) return;
    for (size_t r = 0; r < g->comm.size(); r++) {
        (void)hipSetDevice(g->dev[r]);
        if (g->comm[r]) (void)R->CommAbort(g->comm[r]);
        g->comm[r] = nullptr;
        if (r < g->comm2.size() && g->comm2[r]) (void)R->CommAbort(g->comm2[r]);
        if (r < g->comm2.size()) g->comm2[r] = nullptr;
    }
}
#define SHARD_ALIVE(g)                                                                                                       \
    do {                                                                                                                     \
        if ((g)->broken)                                                                                                     \
            return fail(MMIDX_ERR_HIP, "this sharded handle's communicators were aborted after a failed collective round: destroy it"); \
    } while (0)

#define BARRIER(g)                                                                          \
    do {                                                                                    \
        if (!(g)->bar.wait()) return fail(MMIDX_ERR_HIP, "shard barrier aborted: another shard failed"); \
    } while (0)

// ---- collectives (in place), worker r, on the shard's stream ------------------------------------------------------------
// all-gather: slice s of `buf` (bytes each) comes from shard s
int coll_allgather(ShardGroup *g, int r, void *buf, size_t bytes, bool second = false) {
    if (bytes == 0) return MMIDX_OK;
    hipStream_t st = second ? g->st2[(size_t)r] : g->st[(size_t)r];
    if (g->rccl) {
        RcclApi *R = rccl_api();
        BARRIER(g);  // (a worker that failed on the way never enqueues: its peers must not either, or their streams hang in the collective)
        NCCLCK(R->AllGather((const char *)buf + (size_t)r * bytes, buf, bytes, ncclInt8, second ? g->comm2[(size_t)r] : g->comm[(size_t)r], st));
        return MMIDX_OK;
    }
    g->pub[(size_t)r] = buf;
    HIPCK(hipStreamSynchronize(st));
    BARRIER(g);
    for (int s = 0; s < g->n; s++) {
        if (s == r) continue;
        HIPCK(hipMemcpyPeerAsync((char *)buf + (size_t)s * bytes, g->dev[(size_t)r], (const char *)g->pub[(size_t)s] + (size_t)s * bytes,
                                 g->dev[(size_t)s], bytes, st));
    }
    HIPCK(hipStreamSynchronize(st));
    BARRIER(g);
    return MMIDX_OK;
}

// all-reduce of n elements: op 0 = min over f64, 1 = sum over i32, 2 = max over i32
int coll_allreduce(ShardGroup *g, int r, void *buf, long long n, int op) {
    if (n == 0) return MMIDX_OK;
    hipStream_t st = g->st[(size_t)r];
    if (g->rccl) {
        RcclApi *R = rccl_api();
        BARRIER(g);
        if (op == 0) NCCLCK(R->AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclMin, g->comm[(size_t)r], st));
        else NCCLCK(R->AllReduce(buf, buf, (size_t)n, ncclInt32, op == 1 ? ncclSum : ncclMax, g->comm[(size_t)r], st));
        return MMIDX_OK;
    }
    ShardBufs &B = g->buf[(size_t)r];
    const size_t esz = op == 0 ? 8 : 4;
    HIPCK(B.tmp.reserve((size_t)n * esz));
    g->pub[(size_t)r] = buf;
    HIPCK(hipStreamSynchronize(st));
    BARRIER(g);
    PeerPtrs pp{};
    for (int s = 0; s < g->n; s++) pp.p[s] = g->pub[(size_t)s];
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (op == 0) hipLaunchKernelGGL((k_reduce_peers<double, 0>), dim3(grid), dim3(256), 0, st, pp, g->n, (double *)B.tmp.p, n);
    else if (op == 1) hipLaunchKernelGGL((k_reduce_peers<int32_t, 1>), dim3(grid), dim3(256), 0, st, pp, g->n, (int32_t *)B.tmp.p, n);
    else hipLaunchKernelGGL((k_reduce_peers<int32_t, 2>), dim3(grid), dim3(256), 0, st, pp, g->n, (int32_t *)B.tmp.p, n);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(st));
    BARRIER(g);  // every shard has read every buffer
    HIPCK(hipMemcpyAsync(buf, B.tmp.p, (size_t)n * esz, hipMemcpyDeviceToDevice, st));
    return MMIDX_OK;
}

// largest number of queries one collective round may carry (the shard phases take one sub-batch per call).  Computed on the
// workers: a shard's plan depends on its CSR (longest list, pool size), which is rebuilt there -- device current -- when records
// were added since the last search; the caller's thread has neither the device nor an up-to-date CSR.
int shard_round_cap(ShardGroup *g, int k, int64_t *cap_out) {
    std::vector<int64_t> qb((size_t)g->n, 0);
    const int64_t want = g->max_round;
    int rc = shard_run(g, [&qb, g, k, want](int r) -> int {
        mmidx_index *s = g->sub[(size_t)r];
        {
            std::lock_guard<std::mutex> lk(s->mu);
            int rc2 = build_csr(s);
            if (rc2) return rc2;
            rc2 = build_grp_tables(s);
            if (rc2) return rc2;
            rc2 = build_mfma_tables(s);
            if (rc2) return rc2;
        }
        SearchPlan pl;
        int rc2 = make_plan(s, k, want, pl, false);
        if (rc2) return rc2;
        qb[(size_t)r] = pl.qb;
        return MMIDX_OK;
    });
    if (rc) return rc;
    int64_t cap = want;
    for (int r = 0; r < g->n; r++) cap = std::min<int64_t>(cap, qb[(size_t)r]);
    cap = std::max<int64_t>(g->n, cap - cap % g->n);
    *cap_out = cap;
    return MMIDX_OK;
}

// One collective round on worker r.  The shard's own queries: host rows Qh (nreal of them, the rest of the slice repeats the
// last row) or a device pointer dQ_own (per rows).  Answers: device pointers (d_iid / d_dist / d_cnt, per rows) or, when those
// are null, the shard's own buffers (the caller copies them out).
// The round's query rows on worker r: the own slice into buffer `slot` (host rows Qh -- nreal of them, the rest of the slice repeats
// the first row -- or the device rows dQ_own), then the in-place all-gather of everybody's.  On the shard's second stream and
// communicator ("shard_pipeline" = 1): the main stream only waits for the events -- own rows before the coarse stage, all rows
// before pass A -- and a call of several rounds enqueues round i + 1's exchange before round i's kernels.
int shard_queries(ShardGroup *g, int r, int64_t per, const double *Qh, int64_t nreal, const double *dQ_own, int slot) {
    const int W = g->n;
    mmidx_index *s = g->sub[(size_t)r];
    ShardBufs &B = g->buf[(size_t)r];
    const bool two = g->pipeline != 0;
    hipStream_t st = two ? g->st2[(size_t)r] : g->st[(size_t)r];
    const int D = s->D;
    DevBuf<double> &QB = slot ? B.Q2 : B.Q;
    if (QB.cap < (size_t)per * W * D) {  // (growing the buffer frees the old one: nothing may still be reading it)
        HIPCK(hipStreamSynchronize(g->st[(size_t)r]));
        HIPCK(hipStreamSynchronize(g->st2[(size_t)r]));
        HIPCK(QB.reserve((size_t)per * W * D));
        B.free_valid[slot] = false;
    }
    if (two && B.free_valid[slot]) HIPCK(hipStreamWaitEvent(st, B.ev_free[slot], 0));  // the round that last used this buffer has finished with it
    double *Qown = QB.p + (size_t)r * per * D;
    if (dQ_own) {
        HIPCK(hipMemcpyAsync(Qown, dQ_own, (size_t)per * D * 8, hipMemcpyDeviceToDevice, st));
    } else {
        if (nreal > 0) HIPCK(hipMemcpyAsync(Qown, Qh, (size_t)nreal * D * 8, hipMemcpyHostToDevice, st));
        if (nreal < per) {  // padding rows of the last slices: any valid query -- the slice's first row, replicated by one kernel
            if (nreal == 0) HIPCK(hipMemcpyAsync(Qown, Qh, (size_t)D * 8, hipMemcpyHostToDevice, st));
            const long long first_pad = std::max<int64_t>(nreal, 1), tot = ((long long)per - first_pad) * D;
            if (tot > 0) {
                hipLaunchKernelGGL(k_repeat_row, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Qown, D, first_pad, (long long)per);
                HIPCK(hipGetLastError());
            }
        }
    }
    if (two) HIPCK(hipEventRecord(B.ev_own[slot], st));
    int rc = coll_allgather(g, r, QB.p, (size_t)per * D * 8, two);
    if (rc) return rc;
    if (two) HIPCK(hipEventRecord(B.ev_q[slot], st));
    return MMIDX_OK;
}

int shard_search_round(ShardGroup *g, int r, int k, int64_t per, const double *Qh, int64_t nreal, const double *dQ_own, int32_t *d_iid,
                       double *d_dist, int32_t *d_cnt, int slot = 0, bool prefetched = false) {
    const int W = g->n;
    mmidx_index *s = g->sub[(size_t)r];
    ShardBufs &B = g->buf[(size_t)r];
    hipStream_t st = g->st[(size_t)r];
    const int D = s->D, w = s->w, K1 = k + 1;
    const int64_t nq = per * W;
    const bool p2p = g->exchange == 0 && g->peer_ok;
    HIPCK(B.cells.reserve((size_t)nq * w));
    HIPCK(B.cdist.reserve((size_t)nq * w));
    HIPCK(B.T.reserve((size_t)nq));
    HIPCK(B.rpd.reserve((size_t)nq * K1));
    HIPCK(B.rpk.reserve((size_t)nq * K1));
    HIPCK(B.rpc.reserve((size_t)nq));
    if (!p2p) {
        HIPCK(B.pd.reserve((size_t)nq * K1));
        HIPCK(B.pk.reserve((size_t)nq * K1));
        HIPCK(B.pc.reserve((size_t)nq));
    }
    if (!d_iid) {
        HIPCK(B.oiid.reserve((size_t)per * k));
        HIPCK(B.odist.reserve((size_t)per * k));
        HIPCK(B.ocnt.reserve((size_t)per));
        d_iid = B.oiid.p;
        d_dist = B.odist.p;
        d_cnt = B.ocnt.p;
    }
    HIPCK(B.flag.reserve((size_t)per));
    HIPCK(B.nflag.reserve(1));
    HIPCK(B.dest.reserve((size_t)W));
    HIPCK(hipMemsetAsync(B.nflag.p, 0, sizeof(int32_t), st));

    // 0. the own slice of the queries, then everybody's (shard_queries: on the second stream unless the caller already enqueued it
    //    for this round, i.e. under the previous round's scans)
    double *Qall = slot ? B.Q2.p : B.Q.p;
    double *Qown = Qall + (size_t)r * per * D;
    int rc = MMIDX_OK;
    if (!prefetched) {
        rc = shard_queries(g, r, per, Qh, nreal, dQ_own, slot);
        if (rc) return rc;
    }
    const bool two = g->pipeline != 0;
    // 1. coarse stage of the own slice -- it needs the own rows only, so it runs NEXT TO the exchange of everybody's rows --;
    //    cells and exact coarse distances of everybody's
    if (two) HIPCK(hipStreamWaitEvent(st, B.ev_own[slot], 0));
    rc = mmidx_coarse_device(s, per, Qown, B.cells.p + (size_t)r * per * w, B.cdist.p + (size_t)r * per * w, st);
    if (rc) return rc;
    rc = coll_allgather(g, r, B.cells.p, (size_t)per * w * 4);
    if (rc) return rc;
    rc = coll_allgather(g, r, B.cdist.p, (size_t)per * w * 8);
    if (rc) return rc;
    if (two) HIPCK(hipStreamWaitEvent(st, B.ev_q[slot], 0));  // pass A reads everybody's rows
    // 2. pass A on the local lists, thresholds shared
    rc = mmidx_shard_pass_a_device(s, k, nq, Qall, B.cells.p, B.T.p, st);
    if (rc) return rc;
    rc = coll_allreduce(g, r, B.T.p, nq, 0);
    if (rc) return rc;
    // 3. pass B; the partial lists go to the queries' owners
    g->pub[(size_t)r] = &B;  // (the receive buffers were reserved above: publish them)
    BARRIER(g);
    if (p2p) {
        // the owners' receive buffers as a device table; it changes only when some shard's buffers grew, and only then is it
        // uploaded again (from pinned memory that nothing rewrites while a copy may be in flight: the stream is drained first)
        bool changed = B.dest_n != W;
        for (int o = 0; o < W && !changed; o++) {
            const ShardBufs *O = (const ShardBufs *)g->pub[(size_t)o];
            changed = B.pin_dest[o].pd != O->rpd.p || B.pin_dest[o].pk != O->rpk.p || B.pin_dest[o].pc != O->rpc.p;
        }
        if (changed) {
            HIPCK(hipStreamSynchronize(st));
            for (int o = 0; o < W; o++) {
                const ShardBufs *O = (const ShardBufs *)g->pub[(size_t)o];
                B.pin_dest[o] = ShardDest{O->rpd.p, O->rpk.p, O->rpc.p};
            }
            B.dest_n = W;
            HIPCK(hipMemcpyAsync(B.dest.p, B.pin_dest, (size_t)W * sizeof(ShardDest), hipMemcpyHostToDevice, st));
        }
        s->shard_dest = B.dest.p;
        s->shard_dest_per = (int)per;
        s->shard_dest_me = r;
        rc = mmidx_shard_pass_b_device(s, k, nq, Qall, B.cells.p, B.cdist.p, B.T.p, B.rpd.p, (int64_t *)B.rpk.p, B.rpc.p, st);
        s->shard_dest = nullptr;
        if (rc) return rc;
        // 4. the owner's merge waits for every shard's pass B
        HIPCK(hipEventRecord(B.ev_b, st));
        BARRIER(g);
        for (int o = 0; o < W; o++)
            if (o != r) HIPCK(hipStreamWaitEvent(st, g->buf[(size_t)o].ev_b, 0));
    } else {
        s->shard_dest = nullptr;
        rc = mmidx_shard_pass_b_device(s, k, nq, Qall, B.cells.p, B.cdist.p, B.T.p, B.pd.p, (int64_t *)B.pk.p, B.pc.p, st);
        if (rc) return rc;
        if (g->rccl) {
            RcclApi *R = rccl_api();
            const size_t ne = (size_t)per * K1;
            BARRIER(g);
            NCCLCK(R->GroupStart());
            for (int o = 0; o < W; o++) {
                NCCLCK(R->Send(B.pd.p + (size_t)o * ne, ne, ncclFloat64, o, g->comm[(size_t)r], st));
                NCCLCK(R->Recv(B.rpd.p + (size_t)o * ne, ne, ncclFloat64, o, g->comm[(size_t)r], st));
                NCCLCK(R->Send(B.pk.p + (size_t)o * ne, ne, ncclInt64, o, g->comm[(size_t)r], st));
                NCCLCK(R->Recv(B.rpk.p + (size_t)o * ne, ne, ncclInt64, o, g->comm[(size_t)r], st));
                NCCLCK(R->Send(B.pc.p + (size_t)o * per, (size_t)per, ncclInt32, o, g->comm[(size_t)r], st));
                NCCLCK(R->Recv(B.rpc.p + (size_t)o * per, (size_t)per, ncclInt32, o, g->comm[(size_t)r], st));
            }
            NCCLCK(R->GroupEnd());
        } else {  // in-process: fetch my slices from every shard's dense lists
            const size_t ne = (size_t)per * K1;
            HIPCK(hipStreamSynchronize(st));
            BARRIER(g);
            for (int o = 0; o < W; o++) {
                ShardBufs *O = (ShardBufs *)g->pub[(size_t)o];
                HIPCK(hipMemcpyPeerAsync(B.rpd.p + (size_t)o * ne, g->dev[(size_t)r], O->pd.p + (size_t)r * ne, g->dev[(size_t)o], ne * 8, st));
                HIPCK(hipMemcpyPeerAsync(B.rpk.p + (size_t)o * ne, g->dev[(size_t)r], O->pk.p + (size_t)r * ne, g->dev[(size_t)o], ne * 8, st));
                HIPCK(hipMemcpyPeerAsync(B.rpc.p + (size_t)o * per, g->dev[(size_t)r], O->pc.p + (size_t)r * per, g->dev[(size_t)o], (size_t)per * 4, st));
            }
            HIPCK(hipStreamSynchronize(st));
            BARRIER(g);
        }
    }
    rc = launch_merge_partials(k, per, W, B.rpd.p, (const int64_t *)B.rpk.p, B.rpc.p, nullptr, d_iid, d_dist, d_cnt, B.flag.p, B.nflag.p, st);
    if (rc) return rc;
    // 5. any straddling tie anywhere?  (the one host read of the round; the call is synchronous anyway)
    {
        const int32_t seq = ++B.seq;
        hipLaunchKernelGGL(k_publish_nflag, dim3(1), dim3(1), 0, st, B.nflag.p, (volatile int32_t *)B.pin_nflag, seq);
        HIPCK(hipGetLastError());
        volatile int32_t *pin = (volatile int32_t *)B.pin_nflag;
        bool seen = false;
        for (long spin = 0; spin < 4000000 && !(seen = pin[1] == seq); spin++) __builtin_ia32_pause();  // (~ tens of ms at most)
        if (!seen) HIPCK(hipStreamSynchronize(st));  // a long round: sleep in the runtime instead (the kernel above then has run)
        g->nflag_host[(size_t)r] = pin[0];
    }
    BARRIER(g);
    int mx = 0;
    for (int o = 0; o < W; o++) mx = std::max(mx, g->nflag_host[(size_t)o]);
    auto done_with_queries = [&]() -> int {  // (the buffer may be refilled by the round after next)
        if (g->pipeline) {
            HIPCK(hipEventRecord(B.ev_free[slot], st));
            B.free_valid[slot] = true;
        }
        return MMIDX_OK;
    };
    if (mx == 0 || g->tie_slots <= 0) return done_with_queries();  // (replay switched off: the merge's (distance, probe rank, iid) order stays)
    const int Fo = g->tie_slots, F = Fo * W;
    HIPCK(B.rows_own.reserve((size_t)Fo));
    HIPCK(B.fq.reserve((size_t)F));
    HIPCK(B.tau.reserve((size_t)F));
    HIPCK(B.counts.reserve((size_t)F * w * 2));
    HIPCK(B.pB.reserve((size_t)F));
    HIPCK(B.ties.reserve((size_t)F * k));
    const int rounds = (mx + Fo - 1) / Fo;
    if (r == 0) {
        g->tie_rounds.fetch_add(rounds, std::memory_order_relaxed);
        for (int o = 0; o < W; o++) g->tie_queries.fetch_add(g->nflag_host[(size_t)o], std::memory_order_relaxed);
    }
    for (int round = 0; round < rounds; round++) {
        hipLaunchKernelGGL(k_tie_slots, dim3(1), dim3(64), 0, st, B.flag.p, d_dist, (int)per, k, (int)(r * per), round, Fo, B.rows_own.p,
                           B.fq.p + (size_t)r * Fo, B.tau.p + (size_t)r * Fo);
        HIPCK(hipGetLastError());
        rc = coll_allgather(g, r, B.fq.p, (size_t)Fo * 4);
        if (rc) return rc;
        rc = coll_allgather(g, r, B.tau.p, (size_t)Fo * 8);
        if (rc) return rc;
        HIPCK(hipMemsetAsync(B.counts.p, 0, (size_t)F * w * 2 * 4, st));
        HIPCK(hipMemsetAsync(B.pB.p, 0, (size_t)F * 4, st));
        hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)(((size_t)F * k + 255) / 256)), dim3(256), 0, st, B.ties.p, -1, (long long)F * k);
        HIPCK(hipGetLastError());
        rc = mmidx_shard_tie_phase_device(s, 0, k, F, Qall, B.cells.p, B.fq.p, B.tau.p, B.counts.p, B.pB.p, B.ties.p, st);
        if (rc) return rc;
        rc = coll_allreduce(g, r, B.counts.p, (long long)F * w * 2, 1);
        if (rc) return rc;
        rc = mmidx_shard_tie_phase_device(s, 1, k, F, Qall, B.cells.p, B.fq.p, B.tau.p, B.counts.p, B.pB.p, B.ties.p, st);
        if (rc) return rc;
        rc = coll_allreduce(g, r, B.pB.p, (long long)F, 1);
        if (rc) return rc;
        rc = mmidx_shard_tie_phase_device(s, 2, k, F, Qall, B.cells.p, B.fq.p, B.tau.p, B.counts.p, B.pB.p, B.ties.p, st);
        if (rc) return rc;
        rc = coll_allreduce(g, r, B.ties.p, (long long)F * k, 2);
        if (rc) return rc;
        hipLaunchKernelGGL(k_tie_patch, dim3((unsigned)Fo), dim3(64), 0, st, B.rows_own.p, B.ties.p + (size_t)r * Fo * k, Fo, k, d_iid);
        HIPCK(hipGetLastError());
    }
    HIPCK(hipStreamSynchronize(st));
    return done_with_queries();
}

int sharded_check_search(mmidx_index *h, int k) {
    ShardGroup *g = h->grp;
    if (k < 1 || k > MMIDX_K_MAX) return fail(MMIDX_ERR_INVALID_ARG, "k must be in 1..%d (got %d)", MMIDX_K_MAX, k);
    int rc = check_ready(g->sub[0]);
    if (rc) return rc;
    if (h->w < 1 || h->w > h->C) return fail(MMIDX_ERR_INVALID_ARG, "w = %d outside 1..%d (setW)", h->w, h->C);
    return MMIDX_OK;
}

// host queries -> host answers (the caller holds call_mu)
int sharded_search_host(mmidx_index *h, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out, int32_t *count_out) {
    ShardGroup *g = h->grp;
    int rc = sharded_check_search(h, k);
    if (rc) return rc;
    SHARD_ALIVE(g);
    int64_t cap = 0;
    rc = shard_round_cap(g, k, &cap);
    if (rc) return rc;
    const int W = g->n, D = h->D;
    // ONE job for the whole call: every worker walks the rounds, and enqueues round i + 1's query exchange (second stream, other
    // buffer) before round i's kernels
    rc = shard_run(g, [=](int r) -> int {
        auto slice = [&](int64_t q0, int64_t &per, int64_t &lo, int64_t &nreal) {
            const int64_t nr = std::min<int64_t>(cap, nq - q0);
            per = (nr + W - 1) / W;
            lo = std::min<int64_t>((int64_t)r * per, nr);
            nreal = std::min<int64_t>(lo + per, nr) - lo;
        };
        int round = 0;
        for (int64_t q0 = 0; q0 < nq; q0 += cap, round++) {
            int64_t per, lo, nreal;
            slice(q0, per, lo, nreal);
            const int slot = round & 1;
            // (a slice with no real query still takes part in every collective: it repeats the round's first query)
            const double *Qh = Q + (size_t)(q0 + (nreal > 0 ? lo : 0)) * D;
            int rc2;
            if (round == 0) {
                rc2 = shard_queries(g, r, per, Qh, nreal, nullptr, slot);
                if (rc2) return rc2;
            }
            if (q0 + cap < nq && g->pipeline) {
                int64_t per1, lo1, nreal1;
                slice(q0 + cap, per1, lo1, nreal1);
                rc2 = shard_queries(g, r, per1, Q + (size_t)(q0 + cap + (nreal1 > 0 ? lo1 : 0)) * D, nreal1, nullptr, slot ^ 1);
                if (rc2) return rc2;
            }
            rc2 = shard_search_round(g, r, k, per, Qh, nreal, nullptr, nullptr, nullptr, nullptr, slot, round == 0 || g->pipeline != 0);
            if (rc2) return rc2;
            if (nreal > 0) {
                ShardBufs &B = g->buf[(size_t)r];
                hipStream_t st = g->st[(size_t)r];
                HIPCK(hipMemcpyAsync(iid_out + (size_t)(q0 + lo) * k, B.oiid.p, (size_t)nreal * k * 4, hipMemcpyDeviceToHost, st));
                HIPCK(hipMemcpyAsync(dist_out + (size_t)(q0 + lo) * k, B.odist.p, (size_t)nreal * k * 8, hipMemcpyDeviceToHost, st));
                HIPCK(hipMemcpyAsync(count_out + (q0 + lo), B.ocnt.p, (size_t)nreal * 4, hipMemcpyDeviceToHost, st));
                HIPCK(hipStreamSynchronize(st));
            }
        }
        return MMIDX_OK;
    });
    if (rc) {
        const std::string keep = g_err;
        shard_poison(g);
        g_err = keep;
        return rc;
    }
    return MMIDX_OK;
}

// serves a batch of combined mmidx_search callers (same k) on a sharded handle
int sharded_search_batch(mmidx_index *h, SearchReq *const *batch, size_t nb) {
    ShardGroup *g = h->grp;
    std::lock_guard<std::mutex> lk(g->call_mu);
    if (nb == 1) return sharded_search_host(h, batch[0]->k, batch[0]->nq, batch[0]->Q, batch[0]->iid, batch[0]->dist, batch[0]->cnt);
    const int k = batch[0]->k, D = h->D;
    int64_t tot = 0;
    for (size_t i = 0; i < nb; i++) tot += batch[i]->nq;
    std::vector<double> Q((size_t)tot * D), dd((size_t)tot * k);
    std::vector<int32_t> ii((size_t)tot * k), cc((size_t)tot);
    {
        size_t off = 0;
        for (size_t i = 0; i < nb; i++) {
            memcpy(Q.data() + off, batch[i]->Q, (size_t)batch[i]->nq * D * 8);
            off += (size_t)batch[i]->nq * D;
        }
    }
    int rc = sharded_search_host(h, k, tot, Q.data(), ii.data(), dd.data(), cc.data());
    if (rc) return rc;
    int64_t q0 = 0;
    for (size_t i = 0; i < nb; i++) {
        const int64_t nq = batch[i]->nq;
        memcpy(batch[i]->dist, dd.data() + (size_t)q0 * k, (size_t)nq * k * 8);
        memcpy(batch[i]->iid, ii.data() + (size_t)q0 * k, (size_t)nq * k * 4);
        memcpy(batch[i]->cnt, cc.data() + q0, (size_t)nq * 4);
        q0 += nq;
    }
    return MMIDX_OK;
}

int sharded_search(mmidx_index *h, int k, int64_t nq, const double *Q, int32_t *iid_out, double *dist_out, int32_t *count_out) {
    int rc = sharded_check_search(h, k);
    if (rc) return rc;
    if (nq == 0) return MMIDX_OK;
    SearchReq me;
    me.k = k;
    me.nq = nq;
    me.Q = Q;
    me.iid = iid_out;
    me.dist = dist_out;
    me.cnt = count_out;
    return combiner_submit(h->grp->comb, me, MMIDX_COMB_MAX_Q, [h](SearchReq *const *batch, size_t nb) { return sharded_search_batch(h, batch, nb); });
}

// ---- indexing ----------------------------------------------------------------------------------------------------------
// Encodes rows [lo, hi) of a round on worker r: X from the host (Xh, the round's first row) or from the shard's device
// (dX_own: the slice itself).  cells / codes (stored form) of the whole round land in the host arrays hc / hk.
int shard_encode_slice(ShardGroup *g, int r, const double *Xh, const double *dX_own, int64_t lo, int64_t hi, int32_t *hc, unsigned char *hk) {  // hc == null: the records stay on the device
    mmidx_index *s = g->sub[(size_t)r];
    ShardBufs &B = g->buf[(size_t)r];
    hipStream_t st = g->st[(size_t)r];
    const int64_t n = hi - lo;
    if (n <= 0) return MMIDX_OK;
    const size_t cb = (size_t)s->m * s->code_bytes;
    const double *dX = dX_own;
    if (!dX) {
        HIPCK(B.X.reserve((size_t)n * s->D));
        HIPCK(hipMemcpyAsync(B.X.p, Xh + (size_t)lo * s->D, (size_t)n * s->D * 8, hipMemcpyHostToDevice, st));
        dX = B.X.p;
    }
    HIPCK(B.ecell.reserve((size_t)n));
    HIPCK(B.ecode.reserve((size_t)n * cb));
    int rc = mmidx_encode_device(s, n, dX, B.ecell.p, B.ecode.p, st);
    if (rc) return rc;
    if (hc) {
        HIPCK(hipMemcpyAsync(hc + lo, B.ecell.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipMemcpyAsync(hk + (size_t)lo * cb, B.ecode.p, (size_t)n * cb, hipMemcpyDeviceToHost, st));
    }
    HIPCK(hipStreamSynchronize(st));
    return MMIDX_OK;
}

// ---- device-side routing of a round's records (round 5) ---------------------------------------------------------------------------
// Every shard has encoded its slice on its own device (ecell / ecode of ShardBufs).  Shard r picks ITS records (cell mod n == r) out
// of all slices, in batch order -- the order in which the reference appends to a list (invertedLists[c].add, IVFPQ.java:337-346) --
// with three small kernels on its own device, reading the peers' slices over xGMI (peer access; the same device for in-process
// shards): per-block counts, one block scan, a stable scatter into (iid, cell, code) arrays that mmidx_add_codes_device takes.
// Until round 5 every slice went to the host, every shard walked all of the round's records there, and its share came back.
#define ROUTE_MAXW 64
#define ROUTE_BLK 4096  // records per block: 256 threads x 16 consecutive records (a thread's records stay in order)
struct RouteSrc {
    const int32_t *cell[ROUTE_MAXW];
    const unsigned char *code[ROUTE_MAXW];
    long long n[ROUTE_MAXW], lo[ROUTE_MAXW];  // rows of slice s; its first row's index in the batch
    int blk0[ROUTE_MAXW + 1];                 // first block of slice s
    int W, r, cb;
};
__device__ __forceinline__ int route_slice(const RouteSrc &S, int b) {
    int s = 0;
    while (s + 1 < S.W && b >= S.blk0[s + 1]) s++;
    return s;
}
__global__ __launch_bounds__(256) void k_route_count(const RouteSrc S, int32_t *__restrict__ blkcnt) {
    __shared__ int s_w[4];
    const int b = blockIdx.x, s = route_slice(S, b), tid = threadIdx.x;
    const long long i0 = (long long)(b - S.blk0[s]) * ROUTE_BLK + (long long)tid * 16;
    int c = 0;
    for (int u = 0; u < 16; u++) {
        const long long i = i0 + u;
        if (i < S.n[s]) {
            const int cell = S.cell[s][i];
            c += (cell >= 0 && cell % S.W == S.r);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((tid & 63) == 0) s_w[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) blkcnt[b] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// exclusive scan of the block counts in place; blk[nblk] = total
__global__ __launch_bounds__(1024) void k_route_scan(int32_t *__restrict__ blk, int nblk) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const u32 c = i < nblk ? (u32)blk[i] : 0u;
        const u32 incl = wave_incl_scan_u32(c);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        u32 before = s_carry;
#pragma unroll
        for (int j = 0; j < 16; j++) before += (j < wv) ? s_wave[j] : 0u;
        if (i < nblk) blk[i] = (int32_t)(before + incl - c);
        __syncthreads();
        if (tid == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) blk[nblk] = (int32_t)s_carry;
}
__global__ __launch_bounds__(256) void k_route_scatter(const RouteSrc S, const int32_t *__restrict__ blkoff, int32_t iid0, int32_t *__restrict__ oiid,
                                                       int32_t *__restrict__ ocell, unsigned char *__restrict__ ocode) {
    __shared__ u32 s_w[4];
    const int b = blockIdx.x, s = route_slice(S, b), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long i0 = (long long)(b - S.blk0[s]) * ROUTE_BLK + (long long)tid * 16;
    int cells[16];
    u32 c = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const long long i = i0 + u;
        cells[u] = i < S.n[s] ? S.cell[s][i] : -1;
        c += (cells[u] >= 0 && cells[u] % S.W == S.r);
    }
    const u32 incl = wave_incl_scan_u32(c);
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    u32 o = (u32)blkoff[b] + incl - c;
    for (int j = 0; j < 4; j++) o += (j < wv) ? s_w[j] : 0u;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        if (cells[u] >= 0 && cells[u] % S.W == S.r) {
            const long long i = i0 + u;
            oiid[o] = (int32_t)(iid0 + S.lo[s] + i);
            ocell[o] = cells[u];
            const unsigned char *src = S.code[s] + (size_t)i * S.cb;
            unsigned char *dst = ocode + (size_t)o * S.cb;
            if ((S.cb & 15) == 0)
                for (int t = 0; t < S.cb; t += 16) *(uint4 *)(dst + t) = *(const uint4 *)(src + t);
            else
                for (int t = 0; t < S.cb; t++) dst[t] = src[t];
            o++;
        }
    }
}

// worker r's part of a device-routed round: count, reserve (all shards, then a barrier), scatter, append
int shard_route_device(ShardGroup *g, int r, const std::vector<int64_t> &lo, int32_t iid0) {
    mmidx_index *s = g->sub[(size_t)r];
    ShardBufs &B = g->buf[(size_t)r];
    hipStream_t st = g->st[(size_t)r];
    const int W = g->n;
    RouteSrc S{};
    S.W = W;
    S.r = r;
    S.cb = s->m * s->code_bytes;
    int nblk = 0;
    for (int q = 0; q < W; q++) {
        S.cell[q] = g->buf[(size_t)q].ecell.p;
        S.code[q] = g->buf[(size_t)q].ecode.p;
        S.n[q] = lo[(size_t)q + 1] - lo[(size_t)q];
        S.lo[q] = lo[(size_t)q];
        S.blk0[q] = nblk;
        nblk += (int)((S.n[q] + ROUTE_BLK - 1) / ROUTE_BLK);
    }
    S.blk0[W] = nblk;
    int32_t mine = 0;
    if (nblk > 0) {
        HIPCK(B.rblk.reserve((size_t)nblk + 2));
        hipLaunchKernelGGL(k_route_count, dim3((unsigned)nblk), dim3(256), 0, st, S, B.rblk.p);
        hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, st, B.rblk.p, nblk);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(&mine, B.rblk.p + nblk, 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
    }
    // all or nothing: every shard makes room for its share of the round first (the one step of an append that can fail for lack of
    // memory); a shard that cannot aborts the barrier and nobody appends
    {
        std::lock_guard<std::recursive_mutex> alk(s->add_mu);
        int rc = ensure_pending(s, mine);
        if (rc) return rc;
        if (mine > 0) {
            HIPCK(B.riid.reserve((size_t)mine));
            HIPCK(B.rcell.reserve((size_t)mine));
            HIPCK(B.rcode.reserve((size_t)mine * (size_t)S.cb + 16));
        }
    }
    BARRIER(g);
    if (mine == 0) return MMIDX_OK;
    hipLaunchKernelGGL(k_route_scatter, dim3((unsigned)nblk), dim3(256), 0, st, S, B.rblk.p, iid0, B.riid.p, B.rcell.p, B.rcode.p);
    HIPCK(hipGetLastError());
    return mmidx_add_codes_device(s, mine, B.riid.p, B.rcell.p, B.rcode.p, st);
}

// Appends the records of a round that belong to worker r's lists (cell mod n == r), in row order = arrival order
// (invertedLists[c].add, IVFPQ.java:337-346): iid = iids[i] or iid0 + i.
int shard_append_owned(ShardGroup *g, int r, int64_t n, const int32_t *iids, int32_t iid0, const int32_t *hc, const unsigned char *hk) {
    mmidx_index *s = g->sub[(size_t)r];
    const size_t cb = (size_t)s->m * s->code_bytes;
    const int W = g->n;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; i++) cnt += (hc[i] >= 0 && hc[i] % W == r);
    if (cnt == 0) return MMIDX_OK;
    std::vector<int32_t> oi((size_t)cnt), oc((size_t)cnt);
    std::vector<unsigned char> ok((size_t)cnt * cb);
    int64_t j = 0;
    for (int64_t i = 0; i < n; i++) {
        if (hc[i] < 0 || hc[i] % W != r) continue;
        oi[(size_t)j] = iids ? iids[i] : (int32_t)(iid0 + i);
        oc[(size_t)j] = hc[i];
        memcpy(ok.data() + (size_t)j * cb, hk + (size_t)i * cb, cb);
        j++;
    }
    return mmidx_add_codes(s, cnt, oi.data(), oc.data(), ok.data());
}

int64_t sharded_total(const mmidx_index *h) {
    int64_t t = 0;
    for (mmidx_index *s : h->grp->sub) t += total_size(s);
    return t;
}

// indexVectorInternal for a batch on a sharded handle: the rows are encoded 1/n per shard, every record goes to the shard
// that owns its list.  X on the host, or slice r of the batch resident on shard r's device (dXs[r], ns[r] rows; the batch is
// the concatenation of the slices).
int sharded_add_vectors(mmidx_index *h, int64_t n, const double *X, const double *const *dXs, const int64_t *ns, const int32_t *iids,
                        int32_t iid0, int32_t *cell_out, void *code_out) {
    ShardGroup *g = h->grp;
    int rc = check_ready(g->sub[0]);
    if (rc) return rc;
    if (n == 0) return MMIDX_OK;
    std::lock_guard<std::mutex> lk(g->call_mu);
    if (sharded_total(h) + n > 2147483647LL) return fail(MMIDX_ERR_CAPACITY, "Maximum index capacity reached, no more vectors can be indexed!");
    // (loadCounter: read under the lock that serialises the adds -- two callers that let the library number their vectors must
    //  not see the same total; the plain handle reads its offset under add_mu likewise)
    if (iid0 == MMIDX_IID_AUTO) iid0 = (int32_t)sharded_total(h);
    const int W = g->n;
    const size_t cb = (size_t)h->m * h->code_bytes;
    const int64_t R = X ? (int64_t)W * (1 << 18) : n;  // host batches in rounds; device slices in one
    // device slices, nothing asked back, ids numbered by the library: the records never leave the devices (peer access between distinct
    // devices; option "shard_route_host" = 1: the host path)
    if (dXs && !X && !iids && !cell_out && !code_out && g->peer_ok && !g->route_host && W <= ROUTE_MAXW && h->kind == MMIDX_KIND_IVFPQ) {
        std::vector<int64_t> lo((size_t)W + 1, 0);
        for (int r = 0; r < W; r++) lo[(size_t)r + 1] = lo[(size_t)r] + ns[r];
        return shard_run(g, [=, &lo](int r) -> int {
            int rc2 = shard_encode_slice(g, r, nullptr, dXs[r], lo[(size_t)r], lo[(size_t)r + 1], nullptr, nullptr);
            if (rc2) return rc2;
            BARRIER(g);  // every slice's records are in place
            return shard_route_device(g, r, lo, iid0);
        });
    }
    std::vector<int32_t> hc_own;
    std::vector<unsigned char> hk_own;
    for (int64_t i0 = 0; i0 < n; i0 += R) {
        const int64_t nr = std::min<int64_t>(R, n - i0);
        int32_t *hc = cell_out ? cell_out + i0 : nullptr;
        unsigned char *hk = code_out ? (unsigned char *)code_out + (size_t)i0 * cb : nullptr;
        if (!hc) {
            hc_own.resize((size_t)nr);
            hc = hc_own.data();
        }
        if (!hk) {
            hk_own.resize((size_t)nr * cb);
            hk = hk_own.data();
        }
        std::vector<int64_t> lo((size_t)W + 1, 0);
        if (X) {
            const int64_t per = (nr + W - 1) / W;
            for (int r = 0; r <= W; r++) lo[(size_t)r] = std::min<int64_t>((int64_t)r * per, nr);
        } else {
            for (int r = 0; r < W; r++) lo[(size_t)r + 1] = lo[(size_t)r] + ns[r];
        }
        const double *Xr = X ? X + (size_t)i0 * h->D : nullptr;
        const int32_t *ir = iids ? iids + i0 : nullptr;
        const int32_t base = (int32_t)(iid0 + i0);
        rc = shard_run(g, [=, &lo](int r) -> int {
            int rc2 = shard_encode_slice(g, r, Xr, dXs ? dXs[r] : nullptr, lo[(size_t)r], lo[(size_t)r + 1], hc, hk);
            if (rc2) return rc2;
            BARRIER(g);
            // all or nothing: every shard makes room for its share of the round first (the one step of an append that can fail
            // for lack of memory); a shard that cannot aborts the barrier and nobody appends
            {
                mmidx_index *s = g->sub[(size_t)r];
                int64_t mine = 0;
                for (int64_t i = 0; i < nr; i++) mine += (hc[i] >= 0 && hc[i] % g->n == r);
                std::lock_guard<std::recursive_mutex> alk(s->add_mu);
                rc2 = ensure_pending(s, mine);
                if (rc2) return rc2;
            }
            BARRIER(g);
            return shard_append_owned(g, r, nr, ir, base, hc, hk);
        });
        if (rc) return rc;
    }
    if (cell_out && h->kind != MMIDX_KIND_IVFPQ) std::fill(cell_out, cell_out + n, -1);
    return MMIDX_OK;
}

int sharded_add_codes(mmidx_index *h, int64_t n, const int32_t *iids, const int32_t *cells, const void *codes) {
    ShardGroup *g = h->grp;
    if (n == 0) return MMIDX_OK;
    for (int64_t i = 0; i < n; i++)
        if (cells[i] < 0 || cells[i] >= h->C) return fail(MMIDX_ERR_INVALID_ARG, "list id %d outside 0..%d", cells[i], h->C - 1);
    std::lock_guard<std::mutex> lk(g->call_mu);
    if (sharded_total(h) + n > 2147483647LL) return fail(MMIDX_ERR_CAPACITY, "Maximum index capacity reached, no more vectors can be indexed!");
    // (a record that fails validation fails its shard's batch only: validate the code values here so that the call is all-or-nothing)
    const size_t tot = (size_t)n * h->m;
    if (h->code_bytes == 1) {
        if (h->ks < 256) {
            const signed char *c = (const signed char *)codes;
            for (size_t t = 0; t < tot; t++)
                if ((int)c[t] + 128 >= h->ks) return fail(MMIDX_ERR_INVALID_ARG, "code value outside 0..%d (numProductCentroids): the batch was not added", h->ks - 1);
        }
    } else {
        const int16_t *c = (const int16_t *)codes;
        for (size_t t = 0; t < tot; t++)
            if ((int)(uint16_t)c[t] >= h->ks) return fail(MMIDX_ERR_INVALID_ARG, "code value outside 0..%d (numProductCentroids): the batch was not added", h->ks - 1);
    }
    return shard_run(g, [=](int r) -> int { return shard_append_owned(g, r, n, iids, 0, cells, (const unsigned char *)codes); });
}

int sharded_encode(mmidx_index *h, int64_t n, const double *X, int32_t *cell_out, void *code_out) {
    ShardGroup *g = h->grp;
    int rc = check_ready(g->sub[0]);
    if (rc) return rc;
    if (n == 0) return MMIDX_OK;
    std::lock_guard<std::mutex> lk(g->call_mu);
    const int W = g->n;
    const size_t cb = (size_t)h->m * h->code_bytes;
    const int64_t R = (int64_t)W * (1 << 18);
    std::vector<int32_t> hc_own;
    for (int64_t i0 = 0; i0 < n; i0 += R) {
        const int64_t nr = std::min<int64_t>(R, n - i0);
        int32_t *hc = cell_out ? cell_out + i0 : nullptr;
        if (!hc) {
            hc_own.resize((size_t)nr);
            hc = hc_own.data();
        }
        unsigned char *hk = (unsigned char *)code_out + (size_t)i0 * cb;
        const int64_t per = (nr + W - 1) / W;
        const double *Xr = X + (size_t)i0 * h->D;
        rc = shard_run(g, [=](int r) -> int {
            const int64_t lo = std::min<int64_t>((int64_t)r * per, nr), hi = std::min<int64_t>(lo + per, nr);
            return shard_encode_slice(g, r, Xr, nullptr, lo, hi, hc, hk);
        });
        if (rc) return rc;
    }
    return MMIDX_OK;
}

int sharded_sync_locked(ShardGroup *g) { return shard_run(g, [=](int r) -> int { return mmidx_sync_index(g->sub[(size_t)r]); }); }
int sharded_sync(mmidx_index *h) {
    ShardGroup *g = h->grp;
    std::lock_guard<std::mutex> lk(g->call_mu);
    return sharded_sync_locked(g);
}

int sharded_list_sizes(mmidx_index *h, int32_t *sizes_out) {
    ShardGroup *g = h->grp;
    std::lock_guard<std::mutex> lk(g->call_mu);  // (held from the sync to the last offset read: no add in between)
    int rc = sharded_sync_locked(g);
    if (rc) return rc;
    for (int c = 0; c < h->nlists; c++) {
        const mmidx_index *s = g->sub[(size_t)(c % g->n)];
        sizes_out[c] = (int32_t)(s->h_off[(size_t)c + 1] - s->h_off[(size_t)c]);
    }
    return MMIDX_OK;
}

// list-major snapshot over all shards: list c comes from shard c mod n
int sharded_export(mmidx_index *h, int64_t *list_off_out, int32_t *iids_out, void *codes_out) {
    ShardGroup *g = h->grp;
    std::lock_guard<std::mutex> lk(g->call_mu);  // (held across the sync, the offsets and the copies: a concurrent add cannot grow the lists in between)
    int rc = sharded_sync_locked(g);
    if (rc) return rc;
    const size_t cb = (size_t)h->m * h->code_bytes;
    list_off_out[0] = 0;
    for (int c = 0; c < h->nlists; c++) {
        const mmidx_index *s = g->sub[(size_t)(c % g->n)];
        list_off_out[c + 1] = list_off_out[c] + (s->h_off[(size_t)c + 1] - s->h_off[(size_t)c]);
    }
    if (!iids_out && !codes_out) return MMIDX_OK;
    return shard_run(g, [=](int r) -> int {
        mmidx_index *s = g->sub[(size_t)r];
        const int64_t n = s->n_csr;
        if (n == 0) return MMIDX_OK;
        std::vector<int32_t> ids;
        std::vector<unsigned char> codes;
        if (iids_out) {
            ids.resize((size_t)n);
            HIPCK(hipMemcpy(ids.data(), s->d_ids, (size_t)n * 4, hipMemcpyDeviceToHost));
        }
        if (codes_out) {
            codes.resize((size_t)n * cb);
            HIPCK(hipMemcpy(codes.data(), s->d_codes, (size_t)n * cb, hipMemcpyDeviceToHost));
            if (s->code_bytes == 1)
                for (size_t t = 0; t < codes.size(); t++) codes[t] ^= 0x80;  // stored form idx - 128 (PQ.java:555)
        }
        for (int c = r; c < s->nlists; c += g->n) {
            const int64_t a = s->h_off[(size_t)c], len = s->h_off[(size_t)c + 1] - a, o = list_off_out[c];
            if (len == 0) continue;
            if (iids_out) memcpy(iids_out + o, ids.data() + a, (size_t)len * 4);
            if (codes_out) memcpy((unsigned char *)codes_out + (size_t)o * cb, codes.data() + (size_t)a * cb, (size_t)len * cb);
        }
        return MMIDX_OK;
    });
}

// the shard that holds internal id `iid` (each id lives in exactly one shard); the caller holds call_mu
int sharded_locate(mmidx_index *h, int64_t n, const int32_t *iids, std::vector<int> &where) {
    ShardGroup *g = h->grp;
    where.assign((size_t)n, -1);
    for (int r = 0; r < g->n; r++) {
        mmidx_index *s = g->sub[(size_t)r];
        int rc = set_device(s);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(s->mu);
        int32_t *d_pos = nullptr;
        void *d_code = nullptr;
        std::vector<int32_t> pos, cells;
        rc = lookup_records(s, n, iids, &d_pos, &d_code, pos, cells, true);
        if (rc) return rc;
        (void)hipFree(d_pos);
        (void)hipFree(d_code);
        for (int64_t i = 0; i < n; i++)
            if (pos[(size_t)i] >= 0) where[(size_t)i] = r;
    }
    for (int64_t i = 0; i < n; i++)
        if (where[(size_t)i] < 0) return fail(MMIDX_ERR_INVALID_ARG, "Id does not exist!");  // IVFPQ.java:803-805, :868-870
    return MMIDX_OK;
}

int sharded_get_codes(mmidx_index *h, int64_t n, const int32_t *iids, int32_t *cell_out, void *code_out) {
    ShardGroup *g = h->grp;
    std::lock_guard<std::mutex> lk(g->call_mu);
    std::vector<int> where;
    int rc = sharded_locate(h, n, iids, where);
    if (rc) return rc;
    const size_t cb = (size_t)h->m * h->code_bytes;
    for (int r = 0; r < g->n; r++) {
        std::vector<int32_t> sel, ids;
        for (int64_t i = 0; i < n; i++)
            if (where[(size_t)i] == r) {
                sel.push_back((int32_t)i);
                ids.push_back(iids[i]);
            }
        if (sel.empty()) continue;
        std::vector<int32_t> cc(sel.size());
        std::vector<unsigned char> kk(sel.size() * cb);
        rc = mmidx_get_codes(g->sub[(size_t)r], (int64_t)sel.size(), ids.data(), cc.data(), kk.data());
        if (rc) return rc;
        for (size_t j = 0; j < sel.size(); j++) {
            if (cell_out) cell_out[sel[j]] = cc[j];
            if (code_out) memcpy((unsigned char *)code_out + (size_t)sel[j] * cb, kk.data() + j * cb, cb);
        }
    }
    return MMIDX_OK;
}

int sharded_distance(mmidx_index *h, int64_t n, const double *Q, const int32_t *iids, double *dist_out) {
    ShardGroup *g = h->grp;
    int rc = check_ready(g->sub[0]);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g->call_mu);
    std::vector<int> where;
    rc = sharded_locate(h, n, iids, where);
    if (rc) return rc;
    for (int r = 0; r < g->n; r++) {
        std::vector<int32_t> sel, ids;
        for (int64_t i = 0; i < n; i++)
            if (where[(size_t)i] == r) {
                sel.push_back((int32_t)i);
                ids.push_back(iids[i]);
            }
        if (sel.empty()) continue;
        std::vector<double> qq(sel.size() * (size_t)h->D), dd(sel.size());
        for (size_t j = 0; j < sel.size(); j++) memcpy(qq.data() + j * h->D, Q + (size_t)sel[j] * h->D, (size_t)h->D * 8);
        rc = mmidx_distance(g->sub[(size_t)r], (int64_t)sel.size(), qq.data(), ids.data(), dd.data());
        if (rc) return rc;
        for (size_t j = 0; j < sel.size(); j++) dist_out[sel[j]] = dd[j];
    }
    return MMIDX_OK;
}

int sharded_get_stats(mmidx_index *h, mmidx_stats *out) {
    ShardGroup *g = h->grp;
    mmidx_stats acc{};
    acc.passb_items_last = -1;
    for (int r = 0; r < g->n; r++) {
        mmidx_stats s{};
        int rc = mmidx_get_stats(g->sub[(size_t)r], &s);
        if (rc) return rc;
        // times: the slowest shard (the step waits for it); counters: the sum over shards
        acc.total_ms = std::max(acc.total_ms, s.total_ms);
        acc.coarse_ms = std::max(acc.coarse_ms, s.coarse_ms);
        acc.scan_ms = std::max(acc.scan_ms, s.scan_ms);
        acc.merge_ms = std::max(acc.merge_ms, s.merge_ms);
        acc.passa_ms = std::max(acc.passa_ms, s.passa_ms);
        acc.scan_codes += s.scan_codes;
        acc.passa_codes += s.passa_codes;
        acc.verified_codes += s.verified_codes;
        acc.mfma_survivors += s.mfma_survivors;
        acc.mfma_redo_queries += s.mfma_redo_queries;
        acc.mfma_scan_ms += s.mfma_scan_ms;
        acc.mfma_verify_ms += s.mfma_verify_ms;
        acc.mfma_launches += s.mfma_launches;
        acc.passa_mfma_launches = std::max(acc.passa_mfma_launches, s.passa_mfma_launches);
        acc.passa_mfma_sweep1_ms = std::max(acc.passa_mfma_sweep1_ms, s.passa_mfma_sweep1_ms);
        acc.passa_mfma_select_ms = std::max(acc.passa_mfma_select_ms, s.passa_mfma_select_ms);
        acc.passa_mfma_sweep2_ms = std::max(acc.passa_mfma_sweep2_ms, s.passa_mfma_sweep2_ms);
        acc.passa_mfma_verify_ms = std::max(acc.passa_mfma_verify_ms, s.passa_mfma_verify_ms);
        acc.scan_launches = std::max(acc.scan_launches, s.scan_launches);
        acc.passa_launches = std::max(acc.passa_launches, s.passa_launches);
        acc.tie_fallbacks += s.tie_fallbacks;
        if (s.passb_items_last >= 0) acc.passb_items_last = std::max(0, acc.passb_items_last) + s.passb_items_last;
    }
    acc.tie_fallbacks += (int32_t)g->tie_queries.exchange(0, std::memory_order_relaxed);
    g->tie_rounds.store(0, std::memory_order_relaxed);
    *out = acc;
    return MMIDX_OK;
}

const mmidx_index *sharded_first(const mmidx_index *h) { return h->grp->sub[0]; }

int sharded_for_each(mmidx_index *h, const std::function<int(mmidx_index *)> &f) {
    for (mmidx_index *s : h->grp->sub) {
        int rc = f(s);
        if (rc) return rc;
    }
    return MMIDX_OK;
}

// group options: "shard_exchange" (0: pass B stores the partial lists into the owners' buffers over xGMI; 1: dense lists through
// ncclSend / ncclRecv), "tie_slots" (flagged queries per owner and replay round; 0 switches the replay off), "shard_max_round"
// (queries per collective round); every other option goes to the shards' own handles
int sharded_set_option(mmidx_index *h, const char *name, int value) {
    ShardGroup *g = h->grp;
    const std::string n(name);
    if (n == "shard_exchange") {
        if (value != 0 && g->n > 1 && !g->rccl) return fail(MMIDX_ERR_UNSUPPORTED, "shard_exchange = 1 needs RCCL (pairwise distinct devices)");
        g->exchange = value != 0;
        return MMIDX_OK;
    }
    if (n == "tie_slots") {
        g->tie_slots = std::max(0, std::min(value, 4096));
        return MMIDX_OK;
    }
    if (n == "shard_max_round") {
        g->max_round = std::max<int64_t>(g->n, value);
        return MMIDX_OK;
    }
    if (n == "combine") {
        g->comb.enabled = value != 0;
        return MMIDX_OK;
    }
    if (n == "shard_route_host") {
        std::lock_guard<std::mutex> lk(g->call_mu);
        g->route_host = value != 0;
        return MMIDX_OK;
    }
    if (n == "shard_pipeline") {  // 1: the query exchange on the shards' second streams (overlapped); 0: one stream per shard (A/B switch)
        std::lock_guard<std::mutex> lk(g->call_mu);
        g->pipeline = value != 0;
        for (auto &b : g->buf) b.free_valid[0] = b.free_valid[1] = false;
        return MMIDX_OK;
    }
    return sharded_for_each(h, [&](mmidx_index *s) { return mmidx_set_option(s, name, value); });
}

void sharded_destroy(mmidx_index *h) {
    ShardGroup *g = h->grp;
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
        g->quit_pub.store(true, std::memory_order_release);
        g->cv_go.notify_all();
    }
    for (auto &t : g->th)
        if (t.joinable()) t.join();
    for (int r = 0; r < (int)g->buf.size(); r++) {
        (void)hipSetDevice(g->dev[(size_t)r]);
        if (r < (int)g->st.size() && g->st[(size_t)r]) (void)hipStreamSynchronize(g->st[(size_t)r]);
        if (r < (int)g->st2.size() && g->st2[(size_t)r]) {
            (void)hipStreamSynchronize(g->st2[(size_t)r]);
            (void)hipStreamDestroy(g->st2[(size_t)r]);
        }
        g->buf[(size_t)r].release();
    }
    {
        RcclApi *R = (g->rccl || !g->comm.empty() || !g->comm2.empty()) ? rccl_api() : nullptr;
        for (ncclComm_t c : g->comm)
            if (R && c) (void)R->CommDestroy(c);
        for (ncclComm_t c : g->comm2)
            if (R && c) (void)R->CommDestroy(c);
    }
    for (mmidx_index *s : g->sub)
        if (s) mmidx_destroy(s);
    delete g;
    h->grp = nullptr;
}

}  // namespace

extern "C" {

int mmidx_create_sharded(int kind, int D, int m, int ks, int C, int transform, const int32_t *perm, const double *rot, int n_dev,
                         const int *devs, mmidx_index **out) {
    if (!out) return fail(MMIDX_ERR_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (kind != MMIDX_KIND_IVFPQ)
        return fail(MMIDX_ERR_UNSUPPORTED, "a sharded handle partitions inverted lists: IVFPQ only (a flat PQ index has one list)");
    if (n_dev < 1 || n_dev > MMIDX_MAX_SHARDS || !devs) return fail(MMIDX_ERR_INVALID_ARG, "n_dev must be in 1..%d with a device list", MMIDX_MAX_SHARDS);
    const int ndev = mmidx_device_count();
    if (ndev < 1) return fail(MMIDX_ERR_NO_DEVICE, "no HIP device: libmmidx_hip has no CPU fallback");
    bool distinct = true;
    for (int i = 0; i < n_dev; i++) {
        if (devs[i] < 0 || devs[i] >= ndev) return fail(MMIDX_ERR_NO_DEVICE, "device %d outside 0..%d", devs[i], ndev - 1);
        for (int j = 0; j < i; j++) distinct = distinct && devs[i] != devs[j];
    }
    mmidx_index *h = new mmidx_index();
    ShardGroup *g = new ShardGroup();
    h->grp = g;
    g->n = n_dev;
    g->dev.assign(devs, devs + n_dev);
    g->sub.assign((size_t)n_dev, nullptr);
    g->buf.resize((size_t)n_dev);
    g->pub.assign((size_t)n_dev, nullptr);
    g->nflag_host.assign((size_t)n_dev, 0);
    g->rc.assign((size_t)n_dev, 0);
    g->err.resize((size_t)n_dev);
    auto bail = [&](int rc) {
        const std::string keep = g_err;
        sharded_destroy(h);
        delete h;
        g_err = keep;
        return rc;
    };
    for (int r = 0; r < n_dev; r++) {
        int rc = mmidx_create(kind, D, m, ks, C, transform, perm, rot, devs[r], &g->sub[(size_t)r]);
        if (rc) return bail(rc);
        g->st.push_back(g->sub[(size_t)r]->stream);
        if (hipSetDevice(devs[r]) != hipSuccess || hipHostMalloc((void **)&g->buf[(size_t)r].pin_nflag, 64) != hipSuccess ||
            hipHostMalloc((void **)&g->buf[(size_t)r].pin_dest, MMIDX_MAX_SHARDS * sizeof(ShardDest)) != hipSuccess ||
            hipEventCreateWithFlags(&g->buf[(size_t)r].ev_b, hipEventDisableTiming) != hipSuccess)
            return bail(fail(MMIDX_ERR_HIP, "shard %d: pinned word / event allocation failed", r));
        hipStream_t s2 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return bail(fail(MMIDX_ERR_HIP, "shard %d: second stream", r));
        g->st2.push_back(s2);
        for (int i = 0; i < 2; i++)
            if (hipEventCreateWithFlags(&g->buf[(size_t)r].ev_own[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->buf[(size_t)r].ev_q[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->buf[(size_t)r].ev_free[i], hipEventDisableTiming) != hipSuccess)
                return bail(fail(MMIDX_ERR_HIP, "shard %d: event allocation failed", r));
        memset(g->buf[(size_t)r].pin_nflag, 0, 64);
    }
    // every shard stores pass B's lists into the owners' buffers and (in-process collectives) reads its peers': peer access
    for (int i = 0; i < n_dev; i++)
        for (int j = 0; j < n_dev; j++) {
            if (devs[i] == devs[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[i], devs[j]) != hipSuccess || !can) {
                (void)hipGetLastError();
                g->peer_ok = false;
                continue;
            }
            (void)hipSetDevice(devs[i]);
            hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) g->peer_ok = false;
            (void)hipGetLastError();
        }
    if (distinct) {
        RcclApi *R = rccl_api();
        if (!R) return bail(MMIDX_ERR_UNSUPPORTED);
        g->comm.assign((size_t)n_dev, nullptr);
        ncclResult_t nr = R->CommInitAll(g->comm.data(), n_dev, devs);
        if (nr != ncclSuccess) return bail(fail(MMIDX_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_dev, R->GetErrorString(nr)));
        g->comm2.assign((size_t)n_dev, nullptr);  // the query exchange's own communicators: it runs next to the main stream's collectives
        nr = R->CommInitAll(g->comm2.data(), n_dev, devs);
        if (nr != ncclSuccess) return bail(fail(MMIDX_ERR_HIP, "ncclCommInitAll (second set) over %d devices failed: %s", n_dev, R->GetErrorString(nr)));
        g->rccl = true;
        // Two communicators used concurrently per device can deadlock when the devices schedule the two collective kernels in different
        // orders, and this form has never run on more than one physical GPU: on a real multi-device handle everything goes through the
        // main stream / communicator until such a run has passed ("shard_pipeline" = 1 turns the overlap back on; in-process shards and
        // the one-rank communicator keep it).
        if (n_dev > 1) {
            g->pipeline = 0;
            // ... and the appended records are routed through the host: the device-side routing (kernels of shard r reading the peers'
            // slices over xGMI) has only run between in-process shards of one device; "shard_route_host" = 0 opts in
            g->route_host = 1;
        }
    } else if (!g->peer_ok) {
        return bail(fail(MMIDX_ERR_UNSUPPORTED, "shards on repeated devices use in-process collectives, which need peer access between all of them"));
    }
    for (int r = 0; r < n_dev; r++) g->th.emplace_back(shard_worker, g, r);
    h->kind = kind;
    h->D = D;
    h->m = m;
    h->ks = ks;
    h->dsub = D / m;
    h->C = C;
    h->nlists = C;
    h->transform = transform;
    h->w = g->sub[0]->w;
    h->device = devs[0];
    h->code_bytes = ks <= 256 ? 1 : 2;
    *out = h;
    return MMIDX_OK;
}

int mmidx_shard_count(const mmidx_index *h, int *n_out) {
    if (!h || !n_out) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    *n_out = h->grp ? h->grp->n : 1;
    return MMIDX_OK;
}

int mmidx_shard_info(const mmidx_index *h, int shard, int *device_out, int64_t *size_out, int *uses_rccl_out) {
    if (!h) return fail(MMIDX_ERR_INVALID_ARG, "null handle");
    if (!h->grp) {
        if (shard != 0) return fail(MMIDX_ERR_INVALID_ARG, "a plain handle has one shard");
        if (device_out) *device_out = h->device;
        if (size_out) *size_out = total_size(h);
        if (uses_rccl_out) *uses_rccl_out = 0;
        return MMIDX_OK;
    }
    const ShardGroup *g = h->grp;
    if (shard < 0 || shard >= g->n) return fail(MMIDX_ERR_INVALID_ARG, "shard %d outside 0..%d", shard, g->n - 1);
    if (device_out) *device_out = g->dev[(size_t)shard];
    if (size_out) *size_out = total_size(g->sub[(size_t)shard]);
    if (uses_rccl_out) *uses_rccl_out = g->rccl ? 1 : 0;
    return MMIDX_OK;
}

int mmidx_search_sliced_device(mmidx_index *h, int k, int64_t nq_per_shard, const double *const *dQ, int32_t *const *d_iid_out,
                               double *const *d_dist_out, int32_t *const *d_count_out) {
    if (!h || !h->grp) return fail(MMIDX_ERR_INVALID_ARG, "mmidx_search_sliced_device needs a handle made by mmidx_create_sharded");
    ShardGroup *g = h->grp;
    if (nq_per_shard < 0) return fail(MMIDX_ERR_INVALID_ARG, "nq < 0");
    if (nq_per_shard > 0 && (!dQ || !d_iid_out || !d_dist_out || !d_count_out)) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int rc = sharded_check_search(h, k);
    if (rc) return rc;
    if (nq_per_shard == 0) return MMIDX_OK;
    for (int r = 0; r < g->n; r++)
        if (!dQ[r] || !d_iid_out[r] || !d_dist_out[r] || !d_count_out[r]) return fail(MMIDX_ERR_INVALID_ARG, "null slice pointer (shard %d)", r);
    std::lock_guard<std::mutex> lk(g->call_mu);
    SHARD_ALIVE(g);
    int64_t cap = 0;
    rc = shard_round_cap(g, k, &cap);
    if (rc) return rc;
    const int64_t per_cap = std::max<int64_t>(1, cap / g->n);
    const int D = h->D;
    rc = shard_run(g, [=](int r) -> int {
        int round = 0;
        for (int64_t p0 = 0; p0 < nq_per_shard; p0 += per_cap, round++) {
            const int64_t per = std::min<int64_t>(per_cap, nq_per_shard - p0);
            const int slot = round & 1;
            int rc2;
            if (round == 0) {
                rc2 = shard_queries(g, r, per, nullptr, 0, dQ[r] + (size_t)p0 * D, slot);
                if (rc2) return rc2;
            }
            if (p0 + per_cap < nq_per_shard && g->pipeline) {  // the next round's query exchange, under this round's kernels
                const int64_t p1 = p0 + per_cap, per1 = std::min<int64_t>(per_cap, nq_per_shard - p1);
                rc2 = shard_queries(g, r, per1, nullptr, 0, dQ[r] + (size_t)p1 * D, slot ^ 1);
                if (rc2) return rc2;
            }
            rc2 = shard_search_round(g, r, k, per, nullptr, 0, dQ[r] + (size_t)p0 * D, d_iid_out[r] + (size_t)p0 * k, d_dist_out[r] + (size_t)p0 * k,
                                     d_count_out[r] + p0, slot, round == 0 || g->pipeline != 0);
            if (rc2) return rc2;
        }
        return MMIDX_OK;
    });
    if (rc) {
        const std::string keep = g_err;
        shard_poison(g);
        g_err = keep;
        return rc;
    }
    return MMIDX_OK;
}

int mmidx_add_vectors_sliced_device(mmidx_index *h, const int64_t *n_per_shard, const double *const *dX, int32_t iid0) {
    if (!h || !h->grp) return fail(MMIDX_ERR_INVALID_ARG, "mmidx_add_vectors_sliced_device needs a handle made by mmidx_create_sharded");
    if (!n_per_shard || !dX) return fail(MMIDX_ERR_INVALID_ARG, "null argument");
    int64_t n = 0;
    for (int r = 0; r < h->grp->n; r++) {
        if (n_per_shard[r] < 0 || (n_per_shard[r] > 0 && !dX[r])) return fail(MMIDX_ERR_INVALID_ARG, "bad slice (shard %d)", r);
        n += n_per_shard[r];
    }
    return sharded_add_vectors(h, n, nullptr, dX, n_per_shard, nullptr, iid0, nullptr, nullptr);
}

}  // extern "C"
