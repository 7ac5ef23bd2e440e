// pool.cpp -- several GPUs in one process: tfhe_amd_pool / tfhe_amd_cb_pool of include/tfhe_amd.h.
//
// Host C++ written entirely on the C ABI of this library (contexts, pinned host buffers, copies, the batch entry
// points): no HIP call of its own.  The mapping is the reference's independent-item loop
// (parallel/src/test_parallel_multiplications.cpp:62, `#pragma omp parallel for` over samples that never interact):
// one member per device, each with its own context (= its own stream), its own HOST THREAD -- HIP's current device is
// per thread, and a thread that only ever serves one device never switches -- and its own pinned staging buffers;
// the caller's keys are uploaded once to every member (key ownership as CB/lwe_functions.cpp:287-316: the caller keeps
// the host key); a call cuts its batch into contiguous slices (experimental-tfhe_amd/shard.py's rule), every member runs
// copy in -> launch -> copy out on its slice, and nothing is exchanged between devices.  Inside a member a long slice is cut
// into chunks whose kernels run back to back on the context's stream while a second stream copies the next chunk in and a
// third copies the previous one out (three sets of pinned staging buffers, events between the streams): the host-array
// rate approaches the device-resident one.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tfhe_amd.h"

namespace {

// rows a member stages per round trip: bounds the pinned buffers whatever the call's size (config 5 hands over 2.6 GB)
constexpr size_t STAGE_BYTES = (size_t)64 << 20;

// one persistent host thread; run() hands it a job and returns at once, wait() collects the status
class Worker {
   public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void run(std::function<int()> job) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = std::move(job);
            busy_ = true;
        }
        cv_.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !busy_; });
        return rc_;
    }

   private:
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [this] { return busy_ || quit_; });
            if (quit_) return;
            std::function<int()> job = std::move(job_);
            lk.unlock();
            const int rc = job();
            lk.lock();
            rc_ = rc;
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, quit_ = false;
    int rc_ = 0;
    std::thread th_;  // last: started when everything above exists
};

// pinned host + device staging of one member, grown on demand through the member's own context
struct Staging {
    void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    void *d_mid = nullptr;  // intermediate of a two-kernel operation (blind rotation -> key switch), owned here so that two sets can be in flight
    size_t in_bytes = 0, out_bytes = 0, mid_bytes = 0;
    // buffers grow geometrically (at least 1 MiB, at least twice the previous size): calls of varying size -- the shims' coalesced
    // one-sample calls -- must not pay a pinned + device reallocation (milliseconds; the frees synchronise) at every new maximum
    static size_t grown(size_t have, size_t need) {
        const size_t want = have * 2 > need ? have * 2 : need;
        return want > ((size_t)1 << 20) ? want : ((size_t)1 << 20);
    }
    int ensure_mid(tfhe_amd_ctx *c, size_t need) {
        if (mid_bytes >= need) return TFHE_AMD_OK;
        need = grown(mid_bytes, need);
        if (d_mid) tfhe_amd_free(c, d_mid);
        d_mid = nullptr;
        mid_bytes = 0;
        if (int rc = tfhe_amd_malloc(c, &d_mid, need)) return rc;
        mid_bytes = need;
        return TFHE_AMD_OK;
    }
    int ensure(tfhe_amd_ctx *c, size_t in_need, size_t out_need) {
        if (in_bytes < in_need) {
            in_need = grown(in_bytes, in_need);
            if (h_in) tfhe_amd_host_free(c, h_in);
            if (d_in) tfhe_amd_free(c, d_in);
            h_in = d_in = nullptr;
            in_bytes = 0;
            if (int rc = tfhe_amd_host_alloc(c, &h_in, in_need)) return rc;
            if (int rc = tfhe_amd_malloc(c, &d_in, in_need)) return rc;
            in_bytes = in_need;
        }
        if (out_bytes < out_need) {
            out_need = grown(out_bytes, out_need);
            if (h_out) tfhe_amd_host_free(c, h_out);
            if (d_out) tfhe_amd_free(c, d_out);
            h_out = d_out = nullptr;
            out_bytes = 0;
            if (int rc = tfhe_amd_host_alloc(c, &h_out, out_need)) return rc;
            if (int rc = tfhe_amd_malloc(c, &d_out, out_need)) return rc;
            out_bytes = out_need;
        }
        return TFHE_AMD_OK;
    }
    void release(tfhe_amd_ctx *c) {
        if (h_in) tfhe_amd_host_free(c, h_in);
        if (h_out) tfhe_amd_host_free(c, h_out);
        if (d_in) tfhe_amd_free(c, d_in);
        if (d_out) tfhe_amd_free(c, d_out);
        if (d_mid) tfhe_amd_free(c, d_mid);
        h_in = h_out = d_in = d_out = d_mid = nullptr;
        in_bytes = out_bytes = mid_bytes = 0;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// rows of one chunk of the pipelined form: below 2 x this a slice goes as one piece (the blind rotation needs ~2048 samples
// per launch for its full rate: 2 waves per SIMD on every CU).  TFHE_AMD_POOL_OPT_CHUNK_ROWS changes it per pool.
constexpr int DEFAULT_CHUNK_ROWS = 2048;

// Pipelined copy in -> launch -> copy out of rows [lo, hi) inside ONE context: the kernels of all chunks back to back on the
// context's own stream (exactly the device-resident loop), copies in on a second stream, copies out on a third, tied by events;
// three staging sets, so that while chunk k computes, chunk k + 1 is already queued behind it (copied in, kernels enqueued)
// and chunk k - 1 is being copied out and scattered by this thread.  (Round 5 measured the simpler form -- whole chunks
// alternating on two streams -- first: there the key switch of chunk k waits for a CU behind the blind rotation of chunk k + 1,
// which has taken every CU's LDS, and the host learns of chunk k a whole chunk late.  Both forms measure 0.92-0.95 of the
// device-resident rate; this one keeps the kernel order of the resident loop and the reason for what is left is plain: the
// head and tail of a call and the key switch's lower efficiency at 2048 samples -- DESIGN.md section 7, docs/experiments.md.)
// launch(set, d_out, d_in, rows) issues the kernels on the context's current stream and may use set.d_mid; it must not touch
// the context's own scratch.
constexpr int SETS = 3;
struct Pipe {
    Staging st[SETS];
    void *ev_in[SETS] = {nullptr, nullptr, nullptr}, *ev_done[SETS] = {nullptr, nullptr, nullptr}, *ev_out[SETS] = {nullptr, nullptr, nullptr};
    void *s_in = nullptr, *s_out = nullptr;
    int prepare(tfhe_amd_ctx *c, size_t in_need, size_t out_need, size_t mid_need) {
        for (int s = 0; s < SETS; s++) {
            if (int rc = st[s].ensure(c, in_need, out_need)) return rc;
            if (mid_need)
                if (int rc = st[s].ensure_mid(c, mid_need)) return rc;
            // each event on its own: a set whose second creation failed once is completed by the next call, not skipped
            if (!ev_in[s])
                if (int rc = tfhe_amd_event_create(c, &ev_in[s])) return rc;
            if (!ev_done[s])
                if (int rc = tfhe_amd_event_create(c, &ev_done[s])) return rc;
            if (!ev_out[s])
                if (int rc = tfhe_amd_event_create(c, &ev_out[s])) return rc;
        }
        if (!s_in)
            if (int rc = tfhe_amd_stream_create(c, &s_in)) return rc;
        if (!s_out)
            if (int rc = tfhe_amd_stream_create(c, &s_out)) return rc;
        return TFHE_AMD_OK;
    }
    void release(tfhe_amd_ctx *c) {
        for (int s = 0; s < SETS; s++) {
            st[s].release(c);
            if (ev_in[s]) tfhe_amd_event_destroy(c, ev_in[s]);
            if (ev_done[s]) tfhe_amd_event_destroy(c, ev_done[s]);
            if (ev_out[s]) tfhe_amd_event_destroy(c, ev_out[s]);
            ev_in[s] = ev_done[s] = ev_out[s] = nullptr;
        }
        if (s_in) tfhe_amd_stream_destroy(c, s_in);
        if (s_out) tfhe_amd_stream_destroy(c, s_out);
        s_in = s_out = nullptr;
    }
};
// rows_in(first, rows, dst) fills dst with rows first .. first + rows - 1; rows_out(first, rows, src) takes them back
template <class RowsIn, class RowsOut, class Launch>
int pipelined_rows(tfhe_amd_ctx *c, Pipe &p, RowsOut rows_out, size_t out_row, RowsIn rows_in, size_t in_row, size_t mid_row, int lo, int hi, int chunk,
                   Launch launch) {
    if (int rc = p.prepare(c, (size_t)chunk * in_row, (size_t)chunk * out_row, (size_t)chunk * mid_row)) return rc;
    int rows_of[SETS] = {0, 0, 0}, at_of[SETS] = {0, 0, 0};
    static const bool trace = getenv("TFHE_AMD_POOL_TRACE") != nullptr;  // host-side phase times of one call on stderr (experiments)
    double t_wait = 0, t_out = 0, t_in = 0, t_enq = 0;
    const double t_call = now_s();
    auto drain = [&](int s) -> int {  // wait for set s's chunk to have been copied out and hand its rows to the caller
        if (!rows_of[s]) return TFHE_AMD_OK;
        double t0 = now_s();
        if (int rc = tfhe_amd_event_sync(c, p.ev_out[s])) return rc;
        double t1 = now_s();
        rows_out(at_of[s], rows_of[s], p.st[s].h_out);
        t_wait += t1 - t0;
        t_out += now_s() - t1;
        rows_of[s] = 0;
        return TFHE_AMD_OK;
    };
    int rc = TFHE_AMD_OK, k = 0;
    for (int at = lo; at < hi && !rc; at += chunk, k++) {
        const int s = k % SETS, rows = hi - at < chunk ? hi - at : chunk;
        Staging &set = p.st[s];
        rc = drain(s);  // the set's previous chunk (k - 3): its buffers are reused now
        if (rc) break;
        double t0 = now_s();
        rows_in(at, rows, set.h_in);
        double t1 = now_s();
        t_in += t1 - t0;
        rc = tfhe_amd_set_stream(c, p.s_in);                                              // copy in
        if (!rc) rc = tfhe_amd_memcpy_h2d_async(c, set.d_in, set.h_in, (size_t)rows * in_row);
        if (!rc) rc = tfhe_amd_event_record(c, p.ev_in[s]);
        if (!rc) rc = tfhe_amd_set_stream(c, nullptr);                                    // kernels: the context's own stream
        if (!rc) rc = tfhe_amd_stream_wait_event(c, p.ev_in[s]);
        if (!rc) rc = launch(set, set.d_out, set.d_in, rows);
        if (!rc) rc = tfhe_amd_event_record(c, p.ev_done[s]);
        if (!rc) rc = tfhe_amd_set_stream(c, p.s_out);                                    // copy out
        if (!rc) rc = tfhe_amd_stream_wait_event(c, p.ev_done[s]);
        if (!rc) rc = tfhe_amd_memcpy_d2h_async(c, set.h_out, set.d_out, (size_t)rows * out_row);
        if (!rc) rc = tfhe_amd_event_record(c, p.ev_out[s]);
        if (!rc) {
            rows_of[s] = rows;
            at_of[s] = at;
        }
        t_enq += now_s() - t1;
    }
    const int r0 = tfhe_amd_set_stream(c, nullptr);
    if (rc) {  // something failed mid-way: let everything queued finish before the buffers can be touched again
        (void)tfhe_amd_sync(c);
        (void)tfhe_amd_stream_sync(c, p.s_in);
        (void)tfhe_amd_stream_sync(c, p.s_out);
        return rc;
    }
    for (int j = 0; j < SETS; j++) {  // oldest first
        const int r = drain((k + j) % SETS);
        if (!rc) rc = r;
    }
    if (trace)
        fprintf(stderr, "pool member: %d rows in %d chunks, %.3f ms: gather %.3f, enqueue %.3f, wait %.3f, scatter %.3f\n", hi - lo, k,
                1e3 * (now_s() - t_call), 1e3 * t_in, 1e3 * t_enq, 1e3 * t_wait, 1e3 * t_out);
    return rc ? rc : r0;
}

// [lo, hi) of `total` rows owned by member r of m: contiguous, sizes differ by at most one
void slice_of(int total, int r, int m, int *lo, int *hi) {
    const int base = total / m, rem = total % m;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

// copy in -> launch -> copy out of rows [lo, hi) as one piece (in rounds of at most STAGE_BYTES per direction)
template <class RowsIn, class RowsOut, class Launch>
int staged_rows(tfhe_amd_ctx *c, Staging &st, RowsOut rows_out, size_t out_row, RowsIn rows_in, size_t in_row, size_t mid_row, int lo, int hi, Launch launch) {
    const size_t big = in_row > out_row ? in_row : out_row;
    int per_round = (int)(STAGE_BYTES / big);
    if (per_round < 1) per_round = 1;
    if (per_round > hi - lo) per_round = hi - lo;
    if (int rc = st.ensure(c, (size_t)per_round * in_row, (size_t)per_round * out_row)) return rc;
    if (mid_row)
        if (int rc = st.ensure_mid(c, (size_t)per_round * mid_row)) return rc;
    for (int at = lo; at < hi; at += per_round) {
        const int rows = hi - at < per_round ? hi - at : per_round;
        rows_in(at, rows, st.h_in);  // into pinned memory: the copy below then runs at the link's rate
        if (int rc = tfhe_amd_memcpy_h2d(c, st.d_in, st.h_in, (size_t)rows * in_row)) return rc;
        if (int rc = launch(st.d_out, st.d_in, rows)) return rc;
        if (int rc = tfhe_amd_memcpy_d2h(c, st.h_out, st.d_out, (size_t)rows * out_row)) return rc;  // waits for the launch
        rows_out(at, rows, st.h_out);
    }
    return TFHE_AMD_OK;
}

}  // namespace

// ------------------------------------------------------------------ gate-bootstrap pool
struct tfhe_amd_pool {
    tfhe_amd_params p;
    struct Member {
        int device = 0;
        tfhe_amd_ctx *ctx = nullptr;
        tfhe_amd_gsw *bk = nullptr;
        Staging st;  // one-piece form
        Pipe pipe;   // pipelined form: three sets, copy-in / compute / copy-out streams
        Worker *worker = nullptr;
        int last_count = 0, last_chunks = 0;
        double last_seconds = 0;
        std::string err;
    };
    std::vector<Member> m;
    std::mutex call_mu;  // one sharded call at a time
    std::string err;
    int chunk_rows = DEFAULT_CHUNK_ROWS;
    // a load of a key that succeeded on SOME members and failed on others leaves the members with different keys: the operations
    // that use THAT key refuse to run (their outputs would depend on the slice a sample fell into) until a load of it has had the
    // same outcome on every member.  (A failure that is the same everywhere -- bad parameters, a context without key-switch
    // parameters -- leaves every member with its previous key: nothing is mixed.)
    bool bk_mixed = false, ks_mixed = false;
};

namespace {
// every member runs job(member index) on its own thread; the first failure is the call's status
template <class Pool, class Job>
int on_every_member(Pool *pool, Job job) {
    for (size_t i = 0; i < pool->m.size(); i++) pool->m[i].worker->run([pool, i, job] { return job((int)i); });
    int rc = TFHE_AMD_OK;
    for (size_t i = 0; i < pool->m.size(); i++) {
        const int r = pool->m[i].worker->wait();
        if (r && !rc) {
            rc = r;
            pool->err = "member " + std::to_string(i) + " (device " + std::to_string(pool->m[i].device) + "): " + pool->m[i].err;
        }
    }
    return rc;
}
int member_status(tfhe_amd_pool::Member &mb, int rc) {
    if (rc) mb.err = mb.ctx ? tfhe_amd_last_error(mb.ctx) : "no context";
    return rc;
}

enum : unsigned { NEEDS_BK = 1, NEEDS_KS = 2 };  // which keys an operation reads
// launch(ctx, d_mid, d_out, d_in, rows): the operation on device buffers; d_mid (rows x mid_ints) is scratch owned by the caller.
// rows_in / rows_out move rows between the caller's representation and the members' pinned staging buffers; they are called
// on the members' threads, concurrently for DISJOINT row ranges.
template <class RowsIn, class RowsOut, class Launch>
int pool_rows_fn(tfhe_amd_pool *pool, RowsOut rows_out, size_t out_ints, RowsIn rows_in, size_t in_ints, size_t mid_ints, unsigned needs, int count,
                 Launch launch) {
    if (!pool || count < 0) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    if (((needs & NEEDS_BK) && pool->bk_mixed) || ((needs & NEEDS_KS) && pool->ks_mixed)) {  // only the keys this operation uses
        pool->err = "the last load of a key this operation uses succeeded on some members only: the members hold different keys, load it again";
        return TFHE_AMD_ERR_STATE;
    }
    const int members = (int)pool->m.size();
    return on_every_member(pool, [=](int i) {
        tfhe_amd_pool::Member &mb = pool->m[i];
        int lo, hi;
        slice_of(count, i, members, &lo, &hi);
        mb.last_count = hi - lo;
        mb.last_seconds = 0;
        mb.last_chunks = 0;
        if (hi == lo) return (int)TFHE_AMD_OK;
        const double t0 = now_s();
        const int chunk = pool->chunk_rows;
        int rc;
        if (chunk > 0 && hi - lo >= 2 * chunk) {
            mb.last_chunks = (hi - lo + chunk - 1) / chunk;
            rc = pipelined_rows(mb.ctx, mb.pipe, rows_out, out_ints * 4, rows_in, in_ints * 4, mid_ints * 4, lo, hi, chunk,
                                [&](Staging &set, void *o, const void *in, int rows) {
                                    return launch(mb.ctx, (int32_t *)set.d_mid, (int32_t *)o, (const int32_t *)in, rows);
                                });
        } else {
            mb.last_chunks = 1;
            rc = staged_rows(mb.ctx, mb.st, rows_out, out_ints * 4, rows_in, in_ints * 4, mid_ints * 4, lo, hi,
                             [&](void *o, const void *in, int rows) {
                                 return launch(mb.ctx, (int32_t *)mb.st.d_mid, (int32_t *)o, (const int32_t *)in, rows);
                             });
        }
        mb.last_seconds = now_s() - t0;
        return member_status(mb, rc);
    });
}
// flat host arrays
template <class Launch>
int pool_rows(tfhe_amd_pool *pool, int32_t *out, size_t out_ints, const int32_t *x, size_t in_ints, size_t mid_ints, unsigned needs, int count, Launch launch) {
    if (!out || !x) return TFHE_AMD_ERR_PARAM;
    return pool_rows_fn(
        pool, [=](int first, int rows, const void *src) { memcpy(out + (size_t)first * out_ints, src, (size_t)rows * out_ints * 4); }, out_ints,
        [=](int first, int rows, void *dst) { memcpy(dst, x + (size_t)first * in_ints, (size_t)rows * in_ints * 4); }, in_ints, mid_ints, needs, count, launch);
}
// the caller's own representation, through its two callbacks
template <class Launch>
int pool_rows_cb(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, size_t out_ints, tfhe_amd_rows_in_fn get, size_t in_ints, size_t mid_ints, void *user,
                 unsigned needs, int count, Launch launch) {
    if (!put || !get) return TFHE_AMD_ERR_PARAM;
    return pool_rows_fn(
        pool, [=](int first, int rows, const void *src) { put(user, first, rows, (const int32_t *)src); }, out_ints,
        [=](int first, int rows, void *dst) { get(user, first, rows, (int32_t *)dst); }, in_ints, mid_ints, needs, count, launch);
}
}  // namespace

extern "C" {

int tfhe_amd_pool_create(const tfhe_amd_params *params, const int *devices, int n_devices, tfhe_amd_pool **out) {
    if (!params || !devices || !out || n_devices < 1 || n_devices > 64) return TFHE_AMD_ERR_PARAM;
    *out = nullptr;
    tfhe_amd_pool *pool = new tfhe_amd_pool();
    pool->p = *params;
    pool->m.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        pool->m[i].device = devices[i];
        pool->m[i].worker = new Worker();
    }
    // every member creates its context on its own thread: that thread's current device is the member's from then on
    const int rc = on_every_member(pool, [pool](int i) {
        tfhe_amd_pool::Member &mb = pool->m[i];
        const int r = tfhe_amd_ctx_create(&pool->p, mb.device, &mb.ctx);
        if (r) mb.err = "tfhe_amd_ctx_create failed";
        return r;
    });
    if (rc) {
        tfhe_amd_pool_destroy(pool);
        return rc;
    }
    *out = pool;
    return TFHE_AMD_OK;
}

void tfhe_amd_pool_destroy(tfhe_amd_pool *pool) {
    if (!pool) return;
    (void)on_every_member(pool, [pool](int i) {  // device objects are released on the thread (= device) that made them
        tfhe_amd_pool::Member &mb = pool->m[i];
        if (mb.ctx) {
            mb.st.release(mb.ctx);
            mb.pipe.release(mb.ctx);
            if (mb.bk) tfhe_amd_gsw_free(mb.bk);
            tfhe_amd_ctx_destroy(mb.ctx);
        }
        mb.ctx = nullptr;
        mb.bk = nullptr;
        return (int)TFHE_AMD_OK;
    });
    for (auto &mb : pool->m) delete mb.worker;
    delete pool;
}

const char *tfhe_amd_pool_last_error(const tfhe_amd_pool *pool) { return pool ? pool->err.c_str() : "null pool"; }
int tfhe_amd_pool_size(const tfhe_amd_pool *pool) { return pool ? (int)pool->m.size() : 0; }
int tfhe_amd_pool_device(const tfhe_amd_pool *pool, int member) {
    return pool && member >= 0 && member < (int)pool->m.size() ? pool->m[member].device : -1;
}
tfhe_amd_ctx *tfhe_amd_pool_ctx(tfhe_amd_pool *pool, int member) {
    return pool && member >= 0 && member < (int)pool->m.size() ? pool->m[member].ctx : nullptr;
}

static int pool_load(tfhe_amd_pool *pool, const void *bk, bool bk_is_fft, const int32_t *ks) {
    if (!pool || (!bk && !ks)) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    // per member and per key: 1 = loaded, 0 = failed (the member keeps its previous one), -1 = not attempted
    std::vector<int> bk_done(pool->m.size(), -1), ks_done(pool->m.size(), -1);
    int *bk_ok = bk_done.data(), *ks_ok = ks_done.data();
    const int status = on_every_member(pool, [=](int i) {
        tfhe_amd_pool::Member &mb = pool->m[i];
        if (bk) {
            tfhe_amd_gsw *g = nullptr;
            int rc = bk_is_fft ? tfhe_amd_gsw_from_fft(mb.ctx, (const double *)bk, pool->p.n, &g)
                               : tfhe_amd_gsw_from_torus(mb.ctx, bk, pool->p.n, &g);
            if (!rc) rc = tfhe_amd_set_bootstrap_key(mb.ctx, g);
            bk_ok[i] = rc == TFHE_AMD_OK;
            if (rc) {
                if (g) tfhe_amd_gsw_free(g);
                return member_status(mb, rc);
            }
            if (mb.bk) tfhe_amd_gsw_free(mb.bk);  // the previous key, after the new one is in place
            mb.bk = g;
        }
        if (ks) {
            const int rc = tfhe_amd_load_keyswitch_key(mb.ctx, ks);
            ks_ok[i] = rc == TFHE_AMD_OK;
            if (rc) return member_status(mb, rc);
        }
        return (int)TFHE_AMD_OK;
    });
    // a key is MIXED when its load did not have the same outcome on every member (a member that failed the bk step never tried
    // its ks step: counted as a failed one).  The same outcome everywhere -- all loaded, or all refused -- clears the flag.
    auto mixed = [&](const std::vector<int> &done) {
        int ok = 0;
        for (int v : done) ok += v == 1;
        return ok != 0 && ok != (int)done.size();
    };
    if (bk) pool->bk_mixed = mixed(bk_done);
    if (ks) pool->ks_mixed = mixed(ks_done);
    return status;
}
int tfhe_amd_pool_load_keys(tfhe_amd_pool *pool, const double *bkfft, const int32_t *ks) { return pool_load(pool, bkfft, true, ks); }
int tfhe_amd_pool_load_keys_torus(tfhe_amd_pool *pool, const void *bk_torus, const int32_t *ks) {
    return pool_load(pool, bk_torus, false, ks);
}

int tfhe_amd_pool_bootstrap_host(tfhe_amd_pool *pool, int32_t *out, int32_t mu, const int32_t *x, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    const size_t row = (size_t)pool->p.n + 1;
    // blind rotation + extraction into the member's own intermediate buffer, then the key switch: tfhe_amd_bootstrap's two
    // kernels without the context's scratch, so that two chunks can be in flight
    return pool_rows(pool, out, row, x, row, (size_t)pool->p.N + 1, NEEDS_BK | NEEDS_KS, count, [mu](tfhe_amd_ctx *c, int32_t *mid, int32_t *o, const int32_t *in, int rows) {
        const int rc = tfhe_amd_bootstrap_woks(c, mid, mu, in, rows);
        return rc ? rc : tfhe_amd_keyswitch(c, o, mid, rows);
    });
}
int tfhe_amd_pool_bootstrap_woks_host(tfhe_amd_pool *pool, int32_t *out, int32_t mu, const int32_t *x, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    return pool_rows(pool, out, (size_t)pool->p.N + 1, x, (size_t)pool->p.n + 1, 0, NEEDS_BK, count,
                     [mu](tfhe_amd_ctx *c, int32_t *, int32_t *o, const int32_t *in, int rows) { return tfhe_amd_bootstrap_woks(c, o, mu, in, rows); });
}
int tfhe_amd_pool_keyswitch_host(tfhe_amd_pool *pool, int32_t *out, const int32_t *x, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    return pool_rows(pool, out, (size_t)pool->p.ks_n_out + 1, x, (size_t)pool->p.N + 1, 0, NEEDS_KS, count,
                     [](tfhe_amd_ctx *c, int32_t *, int32_t *o, const int32_t *in, int rows) { return tfhe_amd_keyswitch(c, o, in, rows); });
}
// the same three operations on the CALLER's representation of the rows (e.g. an array of LweSample pointers): `get` fills a
// member's pinned staging buffer with rows [first, first + rows), `put` takes the results; no flat intermediate array
static int launch_bootstrap(tfhe_amd_ctx *c, int32_t mu, int32_t *mid, int32_t *o, const int32_t *in, int rows) {
    const int rc = tfhe_amd_bootstrap_woks(c, mid, mu, in, rows);
    return rc ? rc : tfhe_amd_keyswitch(c, o, mid, rows);
}
int tfhe_amd_pool_bootstrap_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int32_t mu, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    const size_t row = (size_t)pool->p.n + 1;
    return pool_rows_cb(pool, put, row, get, row, (size_t)pool->p.N + 1, user, NEEDS_BK | NEEDS_KS, count,
                        [mu](tfhe_amd_ctx *c, int32_t *mid, int32_t *o, const int32_t *in, int rows) { return launch_bootstrap(c, mu, mid, o, in, rows); });
}
int tfhe_amd_pool_bootstrap_woks_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int32_t mu, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    return pool_rows_cb(pool, put, (size_t)pool->p.N + 1, get, (size_t)pool->p.n + 1, 0, user, NEEDS_BK, count,
                        [mu](tfhe_amd_ctx *c, int32_t *, int32_t *o, const int32_t *in, int rows) { return tfhe_amd_bootstrap_woks(c, o, mu, in, rows); });
}
int tfhe_amd_pool_keyswitch_rows(tfhe_amd_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int count) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    return pool_rows_cb(pool, put, (size_t)pool->p.ks_n_out + 1, get, (size_t)pool->p.N + 1, 0, user, NEEDS_KS, count,
                        [](tfhe_amd_ctx *c, int32_t *, int32_t *o, const int32_t *in, int rows) { return tfhe_amd_keyswitch(c, o, in, rows); });
}
int tfhe_amd_pool_set_option(tfhe_amd_pool *pool, int option, int value) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    switch (option) {
        case TFHE_AMD_POOL_OPT_CHUNK_ROWS:
            if (value < 0) return TFHE_AMD_ERR_PARAM;
            pool->chunk_rows = value;
            return TFHE_AMD_OK;
        default:
            pool->err = "unknown pool option";
            return TFHE_AMD_ERR_PARAM;
    }
}
int tfhe_amd_pool_last_split(const tfhe_amd_pool *pool, int *counts, double *seconds) {
    if (!pool) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(const_cast<tfhe_amd_pool *>(pool)->call_mu);  // (the members write these during a call)
    for (size_t i = 0; i < pool->m.size(); i++) {
        if (counts) counts[i] = pool->m[i].last_count;
        if (seconds) seconds[i] = pool->m[i].last_seconds;
    }
    return TFHE_AMD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ circuit-bootstrap pool
struct tfhe_amd_cb_pool {
    tfhe_amd_cb_params p;
    struct Member {
        int device = 0;
        tfhe_amd_cb *cb = nullptr;
        tfhe_amd_ctx *ctx = nullptr;  // the member's level-2 context (memory helpers, streams, events)
        Staging st;  // one-piece form
        Pipe pipe;   // pipelined form (the pipeline's kernels all run on the handle's one stream, in order: its internal workspaces are safe)
        Worker *worker = nullptr;
        std::string err;
    };
    std::vector<Member> m;
    std::mutex call_mu;
    std::string err;
    int chunk_rows = 1024;  // one Torus64 / N2 = 2048 ciphertext per wave on every SIMD: the blind rotation's full-rate batch
    unsigned mixed = 0;     // bit per key component (0 preKS, 1 bk, 2 + u private key-switch plane u) whose last load failed on SOME members
};

namespace {
int cb_member_status(tfhe_amd_cb_pool::Member &mb, int rc) {
    if (rc) mb.err = mb.cb ? tfhe_amd_cb_last_error(mb.cb) : "no handle";
    return rc;
}
template <class Load>
int cb_pool_load(tfhe_amd_cb_pool *pool, const void *key, int component, Load load) {
    if (!pool || !key || component < 0 || component > 31) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    const int status = on_every_member(pool, [=](int i) { return cb_member_status(pool->m[i], load(pool->m[i].cb)); });
    if (status != TFHE_AMD_OK && pool->m.size() > 1)
        pool->mixed |= 1u << component;
    else if (status == TFHE_AMD_OK)
        pool->mixed &= ~(1u << component);
    return status;
}
}  // namespace

extern "C" {

int tfhe_amd_cb_pool_create(const tfhe_amd_cb_params *params, const int *devices, int n_devices, tfhe_amd_cb_pool **out) {
    if (!params || !devices || !out || n_devices < 1 || n_devices > 64) return TFHE_AMD_ERR_PARAM;
    *out = nullptr;
    tfhe_amd_cb_pool *pool = new tfhe_amd_cb_pool();
    pool->p = *params;
    pool->m.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        pool->m[i].device = devices[i];
        pool->m[i].worker = new Worker();
    }
    const int rc = on_every_member(pool, [pool](int i) {
        tfhe_amd_cb_pool::Member &mb = pool->m[i];
        const int r = tfhe_amd_cb_create(&pool->p, mb.device, &mb.cb);
        if (r) mb.err = "tfhe_amd_cb_create failed";
        else mb.ctx = tfhe_amd_cb_ctx_lvl2(mb.cb);
        return r;
    });
    if (rc) {
        tfhe_amd_cb_pool_destroy(pool);
        return rc;
    }
    *out = pool;
    return TFHE_AMD_OK;
}

void tfhe_amd_cb_pool_destroy(tfhe_amd_cb_pool *pool) {
    if (!pool) return;
    (void)on_every_member(pool, [pool](int i) {
        tfhe_amd_cb_pool::Member &mb = pool->m[i];
        if (mb.cb) {
            mb.st.release(mb.ctx);
            mb.pipe.release(mb.ctx);
            tfhe_amd_cb_destroy(mb.cb);
        }
        mb.cb = nullptr;
        mb.ctx = nullptr;
        return (int)TFHE_AMD_OK;
    });
    for (auto &mb : pool->m) delete mb.worker;
    delete pool;
}

const char *tfhe_amd_cb_pool_last_error(const tfhe_amd_cb_pool *pool) { return pool ? pool->err.c_str() : "null pool"; }
int tfhe_amd_cb_pool_size(const tfhe_amd_cb_pool *pool) { return pool ? (int)pool->m.size() : 0; }
tfhe_amd_cb *tfhe_amd_cb_pool_member(tfhe_amd_cb_pool *pool, int member) {
    return pool && member >= 0 && member < (int)pool->m.size() ? pool->m[member].cb : nullptr;
}
int tfhe_amd_cb_pool_load_preks(tfhe_amd_cb_pool *pool, const int32_t *preks) {
    return cb_pool_load(pool, preks, 0, [preks](tfhe_amd_cb *cb) { return tfhe_amd_cb_load_preks(cb, preks); });
}
int tfhe_amd_cb_pool_load_bk_fft(tfhe_amd_cb_pool *pool, const double *bkfft) {
    return cb_pool_load(pool, bkfft, 1, [bkfft](tfhe_amd_cb *cb) { return tfhe_amd_cb_load_bk_fft(cb, bkfft); });
}
int tfhe_amd_cb_pool_load_bk_torus(tfhe_amd_cb_pool *pool, const int64_t *bk) {
    return cb_pool_load(pool, bk, 1, [bk](tfhe_amd_cb *cb) { return tfhe_amd_cb_load_bk_torus(cb, bk); });
}
int tfhe_amd_cb_pool_load_privks_plane(tfhe_amd_cb_pool *pool, int u_plane, const int32_t *plane) {
    return cb_pool_load(pool, plane, 2 + (u_plane & 15), [u_plane, plane](tfhe_amd_cb *cb) { return tfhe_amd_cb_load_privks_plane(cb, u_plane, plane); });
}

}  // extern "C"

namespace {
template <class RowsIn, class RowsOut>
int cb_pool_rows_fn(tfhe_amd_cb_pool *pool, RowsOut rows_out, RowsIn rows_in, int count) {
    if (!pool || count < 0) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    if (pool->mixed) {
        pool->err = "the last load of a key failed on some members: the members hold different keys, load it again";
        return TFHE_AMD_ERR_STATE;
    }
    const int members = (int)pool->m.size();
    const size_t in_row = ((size_t)pool->p.N1 + 1) * 4, out_row = (size_t)2 * pool->p.l1 * 2 * pool->p.N1 * 4;
    return on_every_member(pool, [=](int i) {
        tfhe_amd_cb_pool::Member &mb = pool->m[i];
        int lo, hi;
        slice_of(count, i, members, &lo, &hi);
        if (hi == lo) return (int)TFHE_AMD_OK;
        const int chunk = pool->chunk_rows;
        int rc;
        if (chunk > 0 && hi - lo >= 2 * chunk)
            rc = pipelined_rows(mb.ctx, mb.pipe, rows_out, out_row, rows_in, in_row, 0, lo, hi, chunk, [&](Staging &, void *o, const void *in, int rows) {
                return tfhe_amd_circuit_bootstrap(mb.cb, (int32_t *)o, (const int32_t *)in, rows);
            });
        else
            rc = staged_rows(mb.ctx, mb.st, rows_out, out_row, rows_in, in_row, 0, lo, hi, [&](void *o, const void *in, int rows) {
                return tfhe_amd_circuit_bootstrap(mb.cb, (int32_t *)o, (const int32_t *)in, rows);
            });
        if (rc) {  // the pipeline's message, or the level-2 context's when a copy failed
            const char *e = tfhe_amd_cb_last_error(mb.cb);
            mb.err = (e && *e) ? e : tfhe_amd_last_error(mb.ctx);
        }
        return rc;
    });
}
}  // namespace

extern "C" {

int tfhe_amd_cb_pool_circuit_bootstrap_host(tfhe_amd_cb_pool *pool, int32_t *out, const int32_t *x, int count) {
    if (!pool || !out || !x) return TFHE_AMD_ERR_PARAM;
    const size_t in_row = ((size_t)pool->p.N1 + 1) * 4, out_row = (size_t)2 * pool->p.l1 * 2 * pool->p.N1 * 4;
    return cb_pool_rows_fn(
        pool, [=](int first, int rows, const void *src) { memcpy((char *)out + (size_t)first * out_row, src, (size_t)rows * out_row); },
        [=](int first, int rows, void *dst) { memcpy(dst, (const char *)x + (size_t)first * in_row, (size_t)rows * in_row); }, count);
}
int tfhe_amd_cb_pool_circuit_bootstrap_rows(tfhe_amd_cb_pool *pool, tfhe_amd_rows_out_fn put, tfhe_amd_rows_in_fn get, void *user, int count) {
    if (!pool || !put || !get) return TFHE_AMD_ERR_PARAM;
    return cb_pool_rows_fn(
        pool, [=](int first, int rows, const void *src) { put(user, first, rows, (const int32_t *)src); },
        [=](int first, int rows, void *dst) { get(user, first, rows, (int32_t *)dst); }, count);
}
int tfhe_amd_cb_pool_set_option(tfhe_amd_cb_pool *pool, int option, int value) {
    if (!pool || option != TFHE_AMD_POOL_OPT_CHUNK_ROWS || value < 0) return TFHE_AMD_ERR_PARAM;
    std::lock_guard<std::mutex> lk(pool->call_mu);
    pool->chunk_rows = value;
    return TFHE_AMD_OK;
}

}  // extern "C"
