// tfhe_amd_compat.hpp -- the reference's C++ entry points, served by the MI355X engine.
//
// Header-only C++11 shims over the C ABI (include/tfhe_amd.h).  They keep the NAMES,
// ARGUMENT ORDER and MEANING of the reference functions so that an existing driver links
// against this engine instead of the CPU code:
//
//   library form (CB/lwe_functions.cpp, CB/tgsw_functions.cpp; the reference ships these
//   without headers, struct fields are the ones those files use -- SURVEY 8b):
//       tfhe_bootstrap_FFT            CB/lwe_functions.cpp:434-446
//       tfhe_bootstrap_woKS_FFT       CB/lwe_functions.cpp:399-430
//       tfhe_blindRotateAndExtract_FFT  :366-395     tfhe_blindRotate_FFT  :337-361
//       tGswFFTExternMulToTLwe        CB/tgsw_functions.cpp:424-449
//       lweKeySwitch                  CB/lwe_functions.cpp:163-171
//   FFT plugin (CB/spqlios/lagrangehalfc_impl.h:8-36):
//       class FFT_Processor_AMD  (same five execute_* methods as FFT_Processor_Spqlios),
//       LagrangeHalfCPolynomialAddMulASM-compatible AddMul
//   PoC form (CB/poc_CircuitBootstrapping.cpp, types of CB/poc_types.h): class template
//       PocEngine<Globals>: tfhe_CircuitBootstrapFFT :823-873, CMux :877-879, circuitBootstrapWoKS :530-659,
//       circuitPrivKS :667-698, preKeySwitch :437-465, preModSwitch :472-484.  It is a template
//       so it compiles against the reference's own poc_types.h (include that first) or against
//       any structs with the same members.
//
// Like the reference these calls process ONE sample and are synchronous; they copy the sample
// to the GPU and back.  Called from ONE thread they are for drop-in correctness, not throughput; called from SEVERAL host
// threads at once -- the reference's own parallel construct is `#pragma omp parallel for` over independent items,
// parallel/src/test_parallel_multiplications.cpp:62 -- the bootstrap, key-switch and circuit-bootstrap calls are COALESCED:
// while one launch runs, the calls that arrive are gathered and go out together as one array launch (Coalescer below), so an
// unmodified OpenMP loop over one-sample calls gets batch launches (and, with set_devices, several GPUs).  The ARRAY FORMS
// (tfhe_bootstrap_FFT_array, tfhe_bootstrap_woKS_FFT_array, lweKeySwitch_array,
// PocEngine::tfhe_CircuitBootstrapFFT_array) take the caller's whole loop of samples -- still on the
// reference's struct types -- as one gather, one launch, one scatter: the batch engine's rate behind the
// reference names; code that keeps its data on the GPU calls the batch entry points of tfhe_amd.h directly.  Keys are uploaded once, the first time a
// key object is seen (keyed by its address), and stay resident until release().
// Errors: the reference returns void and asserts/aborts (lwe_functions.cpp:480-481,
// spqlios-fft-impl.cpp:92-97); the shims do the same (message on stderr, abort()).
// circuitBootstrapWoKS follows the LIBRARY rotation semantics, not the PoC's defective loop
// ("PoC defects").
#ifndef TFHE_AMD_COMPAT_HPP
#define TFHE_AMD_COMPAT_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "tfhe_amd.h"

// see tfhe_amd.h: the PoC's `#define k 1` must not rewrite the library-form field names below
#pragma push_macro("k")
#undef k

namespace tfhe_amd_compat {

typedef int32_t Torus32;
typedef int64_t Torus64;

inline void check(int rc, tfhe_amd_ctx *c, const char *what) {
    if (rc != TFHE_AMD_OK) {
        std::fprintf(stderr, "tfhe_amd: %s failed (%d): %s\n", what, rc, c ? tfhe_amd_last_error(c) : "");
        std::abort();
    }
}

// ---- library-form structs (fields evidenced by use in CB/*_functions.cpp) --------
#ifndef TFHE_AMD_COMPAT_NO_TYPES
#include "tfhe_amd_library_types.inc"
#endif

// ---- one resident engine per (key object, shape) -----------------------------------------------
// The reference's own call chain reaches the same TGswSampleFFT array through entry points that use
// different prefixes of it (tGswFFTExternMulToTLwe: 1 sample, tfhe_blindRotate_FFT: n samples) and with
// or without the key-switch key, so the registry is keyed on the pointer AND the shape it was attached with.
struct Resident {
    const void *owner = nullptr;  // the object the caller knows this engine by when it is not the registry key itself
                                  // (attach(bk) registers under bk->bkFFT; release(bk) must find it)
    tfhe_amd_ctx *ctx = nullptr;
    tfhe_amd_gsw *gsw = nullptr;
    int n = 0, N = 0, l = 0;
    void *d_in = nullptr, *d_out = nullptr, *d_aux = nullptr;  // one-sample staging buffers
    // array forms: pinned host staging + device buffers, grown on demand
    void *h_in = nullptr, *h_out = nullptr, *da_in = nullptr, *da_out = nullptr;
    size_t arr_in_bytes = 0, arr_out_bytes = 0;
    uint64_t fingerprint = 0;  // content sample of the host key the GPU copy was made from (see key_fingerprint)
};
struct ResidentKey {
    const void *obj, *ks;
    int n, N, l, Bgbit;
    bool operator<(const ResidentKey &o) const {
        if (obj != o.obj) return obj < o.obj;
        if (ks != o.ks) return ks < o.ks;
        if (n != o.n) return n < o.n;
        if (N != o.N) return N < o.N;
        if (l != o.l) return l < o.l;
        return Bgbit < o.Bgbit;
    }
};
// (The reference itself is single-threaded here: global scratch in fft_processor_spqlios.cpp:21-24,163-164, global RNG
// numeric_functions.cpp:14.  The registry and every resident engine's one set of staging buffers are protected by the lock below.)
inline std::map<ResidentKey, Resident> &registry() {
    static std::map<ResidentKey, Resident> r;
    return r;
}
// One lock around every public entry point of this header (the registry, its staging buffers and the PoC engines are
// shared state): launches issued through this header are SERIALISED, never corrupted.  On top of it the one-sample bootstrap,
// key-switch and circuit-bootstrap calls are coalesced across host threads (Coalescer): concurrency pays there, and through
// the array forms or the batch C ABI (one context per thread).
inline std::recursive_mutex &shim_mutex() {
    static std::recursive_mutex m;
    return m;
}
#define TFHE_AMD_SHIM_GUARD() std::lock_guard<std::recursive_mutex> tfhe_amd_shim_guard_(::tfhe_amd_compat::shim_mutex())
// One-sample calls from several host threads, coalesced.  The first caller to find the coalescer idle becomes the LEADER: it
// takes everything pending (its own request included), runs it as one batch, marks those requests done and -- if more have
// arrived meanwhile -- hands the lead to one of the waiting callers before it returns (no caller serves more than one batch: a
// call waits for at most the batch in flight plus its own).  A lone caller is a batch of one at the one-sample path's latency.
//
// LINGER.  A tight loop of T threads would otherwise settle into two alternating groups of T/2: the callers of the batch in
// flight come back microseconds AFTER the next leader has taken what was pending.  The coalescer keeps an estimate of the
// callers around (the last batch + what was pending when it finished); a leader that holds fewer requests than that waits for
// the stragglers -- at most an eighth of the last launch's duration, 0.5 ms at most -- and goes as soon as they are all there
// (32 callers are back within ~0.1 ms: 6.5 -> 12 k bootstraps/s on one MI355X; with 256 callers the host's turnaround is as long
// as the launch itself, waiting for everyone only idles the GPU, and launches of about half the callers, alternating, are as
// fast: 36-47 k/s with any bound, profiles/r05_array_form_stats.jsonl).  A lone caller never waits (estimate 1).  A wait that
// ran out while callers were still arriving lengthens the next one, a wait during which nobody came halves it (down to 1/16 of
// the bound; the full wait is tried again every 16th batch).  TFHE_AMD_COALESCE_LINGER_US in the environment fixes the bound
// (0: never wait).
// process-wide sums over coalescers that report into them (they outlive a coalescer dropped with its key)
struct CoalescerTotals {
    std::atomic<unsigned long> batches{0}, requests{0}, lingers{0}, linger_timeouts{0};
};
template <class Item>
class Coalescer {
   public:
    explicit Coalescer(CoalescerTotals *totals = nullptr) : totals_(totals) {}
    // run(items): executes the batch; called without the coalescer's lock, by exactly one thread at a time
    template <class RunBatch>
    void call(const Item &item, RunBatch run) {
        Req me;
        me.item = item;
        std::unique_lock<std::mutex> lk(mu_);
        pending_.push_back(&me);
        if (busy_) {
            if (lingering_ && pending_.size() >= crowd_) gather_cv_.notify_one();  // the leader waits for exactly this
            lk.unlock();
            // every request sleeps on its OWN mutex and condition variable: the callers of a finished batch wake side by side
            // instead of queueing, one after the other, for the one lock a shared condition variable would hand them
            std::unique_lock<std::mutex> mine(me.m);
            me.cv.wait(mine, [&] { return me.done || me.lead; });
            if (me.done) {  // another caller's batch carried this request
                if (me.wake_list) {  // ... and its leader made this caller a captain: pass the news on to a share of the batch
                    const std::shared_ptr<std::vector<Req *>> list = std::move(me.wake_list);
                    const size_t lo = me.wake_lo, hi = me.wake_hi;
                    mine.unlock();
                    for (size_t i = lo; i < hi; i++) wake((*list)[i], &Req::done);
                }
                return;
            }
            mine.unlock();
            lk.lock();
        }
        busy_ = true;  // (already true when the lead was handed over)
        if (pending_.size() < crowd_ && linger_bound_.count() > 0) {
            const int shift = (linger_fixed_us() >= 0 || batches_ % 16 == 15) ? 0 : shift_;
            const size_t had = pending_.size();
            lingers_++;
            if (totals_) totals_->lingers++;
            lingering_ = true;
#if defined(__SANITIZE_THREAD__)
            // (gcc 11's ThreadSanitizer runtime does not intercept pthread_cond_clockwait, what a wait on the steady clock
            // becomes, and then reports the mutex as held throughout the wait: its build waits on the system clock)
            const bool all_here = gather_cv_.wait_until(lk, std::chrono::system_clock::now() + linger_bound_ / (1 << shift),
                                                        [&] { return pending_.size() >= crowd_; });
#else
            const bool all_here = gather_cv_.wait_for(lk, linger_bound_ / (1 << shift), [&] { return pending_.size() >= crowd_; });
#endif
            lingering_ = false;
            // all there: full waits from now on.  Ran out while callers were still arriving: they are on their way, wait longer
            // next time.  Ran out and nobody came: the callers have work of their own, halve the wait.
            if (all_here)
                shift_ = 0;
            else if (pending_.size() > had)
                shift_ = shift_ > 0 ? shift_ - 1 : 0;
            else
                shift_ = shift_ < 4 ? shift_ + 1 : 4;
            if (!all_here) {
                linger_timeouts_++;
                if (totals_) totals_->linger_timeouts++;
            }
        }
        std::vector<Req *> batch;
        batch.swap(pending_);
        lk.unlock();
        std::vector<Item> items;
        items.reserve(batch.size());
        for (Req *r : batch) items.push_back(r->item);
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        run(items);
        const std::chrono::nanoseconds took = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0);
        lk.lock();
        crowd_ = batch.size() + pending_.size();
        requests_ += batch.size();
        if (totals_) {
            totals_->requests += batch.size();
            totals_->batches++;
        }
        linger_bound_ = linger_fixed_us() >= 0 ? std::chrono::nanoseconds(1000LL * linger_fixed_us())
                                                : std::min(took / 8, std::chrono::nanoseconds(500000));
        batches_++;
        Req *next = pending_.empty() ? nullptr : pending_.front();  // (stays in pending_: its own request rides in its batch)
        if (!next) busy_ = false;
        lk.unlock();
        // the lead goes out first -- the next leader gathers while this one is still waking its batch; a flag is set and its
        // owner notified under the REQUEST's mutex (the Req lives on its owner's stack until the owner has seen the flag)
        if (next) wake(next, &Req::lead);
        // a large batch is woken in two levels -- ~sqrt(n) captains by this thread, a contiguous share of the rest by each
        // captain -- so that the last caller of 256 hears of it after ~30 wake-ups, not 255
        std::shared_ptr<std::vector<Req *>> others = std::make_shared<std::vector<Req *>>();
        others->reserve(batch.size());
        for (Req *r : batch)
            if (r != &me) others->push_back(r);
        const size_t n = others->size();
        size_t captains = n;
        if (n > 16) {
            captains = 4;
            while (captains * captains < n) captains++;
        }
        const size_t rest = n - captains, share = captains ? (rest + captains - 1) / captains : 0;
        for (size_t j = 0; j < captains; j++) {
            Req *r = (*others)[j];
            std::lock_guard<std::mutex> g(r->m);
            const size_t lo = captains + j * share, hi = lo + share < n ? lo + share : n;
            if (rest && lo < hi) {
                r->wake_list = others;
                r->wake_lo = lo;
                r->wake_hi = hi;
            }
            r->done = true;
            r->cv.notify_one();
        }
    }

    // launches issued and requests carried so far (diagnostics: requests / batches = the mean launch size)
    void stats(unsigned long *batches, unsigned long *requests, unsigned long *lingers = nullptr, unsigned long *linger_timeouts = nullptr) {
        std::lock_guard<std::mutex> lk(mu_);
        *batches = batches_;
        *requests = requests_;
        if (lingers) *lingers = lingers_;
        if (linger_timeouts) *linger_timeouts = linger_timeouts_;
    }
    // TFHE_AMD_COALESCE_LINGER_US in the environment: a FIXED wait bound in microseconds (0: leaders never wait), read once
    static long linger_fixed_us() {
        static const long v = [] {
            const char *e = std::getenv("TFHE_AMD_COALESCE_LINGER_US");
            return e && *e ? std::atol(e) : -1L;
        }();
        return v;
    }

   private:
    struct Req {
        Item item;
        bool done = false, lead = false;
        std::mutex m;
        std::condition_variable cv;
        std::shared_ptr<std::vector<Req *>> wake_list;  // captain duty (set with `done`): wake [wake_lo, wake_hi) of this list
        size_t wake_lo = 0, wake_hi = 0;
    };
    static void wake(Req *r, bool Req::*flag) {
        std::lock_guard<std::mutex> g(r->m);
        r->*flag = true;
        r->cv.notify_one();
    }
    std::mutex mu_;
    std::condition_variable gather_cv_;
    std::vector<Req *> pending_;
    bool busy_ = false, lingering_ = false;
    size_t crowd_ = 1;                        // callers around: size of the last batch + what was pending when it finished
    std::chrono::nanoseconds linger_bound_{0};  // an eighth of the last launch, 0.5 ms at most
    int shift_ = 0;                           // the bound is halved `shift_` times after waits that ran out
    unsigned long batches_ = 0, requests_ = 0, lingers_ = 0, linger_timeouts_ = 0;
    CoalescerTotals *totals_;
};
struct LweCall {
    LweSample *result;
    const LweSample *x;
    Torus32 mu;  // the call's test-vector value (0 for the key switch): calls of one batch are run grouped by it
};
// One coalescer per (entry point, key object): calls on one key meet in it whatever their mu; the leader runs its batch as one
// launch per DISTINCT mu (for_each_mu_group) -- a caller that sweeps mu does not grow this registry.  Entries are shared_ptr: a
// caller holds its coalescer for the length of its call, so release() of the key (which drops the entry) never frees one in use.
typedef std::pair<int, const void *> CoalesceKey;  // (kind, key object)
struct LweCoalescers {
    std::mutex m;
    std::map<CoalesceKey, std::shared_ptr<Coalescer<LweCall>>> reg;
    CoalescerTotals totals;  // all keys, released ones included
};
inline LweCoalescers &lwe_coalescers() {
    static LweCoalescers r;
    return r;
}
inline std::shared_ptr<Coalescer<LweCall>> lwe_coalescer(int kind, const void *key) {
    LweCoalescers &r = lwe_coalescers();
    std::lock_guard<std::mutex> lk(r.m);
    std::shared_ptr<Coalescer<LweCall>> &c = r.reg[CoalesceKey(kind, key)];
    if (!c) c = std::make_shared<Coalescer<LweCall>>(&r.totals);
    return c;
}
inline size_t lwe_coalescer_count() {
    LweCoalescers &r = lwe_coalescers();
    std::lock_guard<std::mutex> lk(r.m);
    return r.reg.size();
}
// the coalescers of a key object that is going away (release()); key == nullptr: all of them (release_all())
inline void lwe_coalescers_drop(const void *key) {
    LweCoalescers &r = lwe_coalescers();
    std::lock_guard<std::mutex> lk(r.m);
    for (auto it = r.reg.begin(); it != r.reg.end();) {
        auto cur = it++;
        if (key && cur->first.second != key) continue;
        r.reg.erase(cur);  // (callers inside it hold their own reference; its counts are in r.totals)
    }
}
// f(mu, calls of that mu) for every distinct mu of a batch, in order of first appearance; calls keep their order inside a group
template <class F>
inline void for_each_mu_group(const std::vector<LweCall> &calls, F f) {
    std::vector<char> taken(calls.size(), 0);
    std::vector<LweCall> group;
    for (size_t i = 0; i < calls.size(); i++) {
        if (taken[i]) continue;
        group.clear();
        for (size_t j = i; j < calls.size(); j++)
            if (!taken[j] && calls[j].mu == calls[i].mu) {
                taken[j] = 1;
                group.push_back(calls[j]);
            }
        f(calls[i].mu, group);
    }
}
// launches and one-sample requests of the coalesced gate entry points so far, all keys (diagnostics)
inline void lwe_coalescer_totals(unsigned long *batches, unsigned long *requests, unsigned long *lingers = nullptr,
                                 unsigned long *linger_timeouts = nullptr) {
    LweCoalescers &r = lwe_coalescers();
    *batches = r.totals.batches.load();
    *requests = r.totals.requests.load();
    if (lingers) *lingers = r.totals.lingers.load();
    if (linger_timeouts) *linger_timeouts = r.totals.linger_timeouts.load();
}

inline int &device_ordinal() {
    static int d = 0;
    return d;
}
inline void set_device(int d) { TFHE_AMD_SHIM_GUARD(); device_ordinal() = d; }
// Several GPUs behind the ARRAY forms: with more than one device named here, tfhe_bootstrap_FFT_array,
// tfhe_bootstrap_woKS_FFT_array, lweKeySwitch_array and PocEngine::tfhe_CircuitBootstrapFFT_array cut the caller's loop into
// contiguous slices over a pool (tfhe_amd_pool / tfhe_amd_cb_pool: one context, host thread and pinned staging buffer per
// device, the caller's key uploaded once to each) -- the reference's `#pragma omp parallel for` over independent items
// (parallel/src/test_parallel_multiplications.cpp:62) with GPUs as the workers.  The one-sample entry points stay on devices[0].
inline std::vector<int> &device_list() {
    static std::vector<int> d;
    return d;
}
inline void set_devices(const int *devices, int n) {
    TFHE_AMD_SHIM_GUARD();
    device_list().assign(devices, devices + (n > 0 ? n : 0));
    if (n > 0) device_ordinal() = devices[0];
}

inline void staging(Resident &R, size_t bytes) {
    if (R.d_in) return;
    check(tfhe_amd_malloc(R.ctx, &R.d_in, bytes), R.ctx, "malloc");
    check(tfhe_amd_malloc(R.ctx, &R.d_out, bytes), R.ctx, "malloc");
    check(tfhe_amd_malloc(R.ctx, &R.d_aux, bytes), R.ctx, "malloc");
}

// Resident copies are found by the ADDRESS of the caller's key object.  The CONTRACT is release(): call it whenever a key
// object's contents change or its memory is reused.  As a guard against the commonest mistake -- a key freed and rebuilt at
// the same address without a release() in between, or one TGSW sample / key-switch row re-encrypted in place -- every lookup
// re-reads a SAMPLE of the host key and compares it with the sample taken at upload time; a difference drops the resident copy
// and uploads the key again.  The sample touches EVERY TGSW sample of the bootstrapping key (one value each: TLWE row, polynomial
// and position walk with the sample index, so all 2l rows, both polynomials and ~n positions are covered across the key; four
// values of the first, middle and last samples besides) and EVERY input coefficient's block of the key-switch key (one mask
// word and the body of one of its t x (base-1) rows): ~n + N loads, 14 us at the gate shape once they sit in the host's caches
// (0.6 % of a one-sample call).
// The detection stays probabilistic: a change elsewhere than at the sampled positions is NOT seen and the stale GPU copy is
// served ("Caveat").
inline uint64_t fp_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h;
}
template <class GswT>
inline uint64_t key_fingerprint(const GswT *bkFFT, int n, int N, int l, const LweKeySwitchKey *ks) {
    uint64_t h = 0x5446484500000001ull;
    auto value_bits = [&](int i, int r, int q, int pos) {
        uint64_t bits;
        std::memcpy(&bits, &bkFFT[i].all_samples[r].a[q].values[pos], 8);
        return bits;
    };
    const int rows[3] = {0, n / 2, n - 1}, pos[4] = {0, 1, N / 2, N - 1};
    for (int a = 0; a < 3 && bkFFT && n > 0; a++)
        for (int r = 0; r < 2 * l; r += (2 * l - 1 > 0 ? 2 * l - 1 : 1))
            for (int q = 0; q < 2; q++)
                for (int b = 0; b < 4; b++) h = fp_mix(h, value_bits(rows[a], r, q, pos[b]));
    for (int i = 0; bkFFT && i < n; i++) h = fp_mix(h, value_bits(i, i % (2 * l), (i / (2 * l)) & 1, (int)(((long long)i * 37 + 5) % N)));
    if (ks) {
        const int is[3] = {0, ks->n / 2, ks->n - 1};
        for (int a = 0; a < 3; a++)
            for (int j = 0; j < ks->t; j += (ks->t - 1 > 0 ? ks->t - 1 : 1)) {
                const LweSample &row = ks->ks[is[a]][j][ks->base - 1];
                h = fp_mix(h, (uint32_t)row.a[0]);
                h = fp_mix(h, (uint32_t)row.a[ks->out_params->n - 1]);
                h = fp_mix(h, (uint32_t)row.b);
            }
        const int nout = ks->out_params->n;
        for (int i = 0; i < ks->n; i++) {
            const LweSample &row = ks->ks[i][i % ks->t][ks->base > 1 ? 1 + i % (ks->base - 1) : 0];
            h = fp_mix(h, ((uint64_t)(uint32_t)row.a[(int)(((long long)i * 29 + 3) % nout)] << 32) | (uint32_t)row.b);
        }
    }
    return h;
}

inline void release_entry(std::map<ResidentKey, Resident>::iterator it);
// the pointer-rich host keys as the flat arrays of the C ABI: [n][2l][2][N] doubles, [ks->n][t][base][n_out+1] int32
template <class GswT>
inline std::vector<double> flatten_bk(const GswT *bkFFT, int n, int N, int l) {
    std::vector<double> flat((size_t)n * 2 * l * 2 * N);
    for (int i = 0; i < n; i++)
        for (int r = 0; r < 2 * l; r++)
            for (int q = 0; q < 2; q++)
                std::memcpy(&flat[(((size_t)i * 2 * l + r) * 2 + q) * N], bkFFT[i].all_samples[r].a[q].values, sizeof(double) * (size_t)N);
    return flat;
}
inline std::vector<int32_t> flatten_ks(const LweKeySwitchKey *ks) {
    const int rows = ks->base, row = ks->out_params->n + 1;
    std::vector<int32_t> kflat((size_t)ks->n * ks->t * rows * row);
    for (int i = 0; i < ks->n; i++)
        for (int j = 0; j < ks->t; j++)
            for (int h = 0; h < rows; h++) {
                int32_t *dst = &kflat[(((size_t)i * ks->t + j) * rows + h) * row];
                std::memcpy(dst, ks->ks[i][j][h].a, sizeof(int32_t) * (size_t)(row - 1));
                dst[row - 1] = ks->ks[i][j][h].b;
            }
    return kflat;
}
// the pools of the array forms (set_devices with more than one device): same keying and content sample as the one-device
// registry, plus the device list the pool was built for
struct PoolResident {
    const void *owner = nullptr;
    tfhe_amd_pool *pool = nullptr;
    std::vector<int> devices;
    uint64_t fingerprint = 0;
    int n = 0, N = 0;
};
inline std::map<ResidentKey, PoolResident> &pool_registry() {
    static std::map<ResidentKey, PoolResident> r;
    return r;
}
inline void pool_check(int rc, tfhe_amd_pool *p, const char *what) {
    if (rc != TFHE_AMD_OK) {
        std::fprintf(stderr, "tfhe_amd: %s failed (%d): %s\n", what, rc, p ? tfhe_amd_pool_last_error(p) : "");
        std::abort();
    }
}
template <class GswT>
inline PoolResident &attach_pool(const GswT *bkFFT, int n, int N, int l, int Bgbit, const LweKeySwitchKey *ks, const void *owner) {
    auto &reg = pool_registry();
    const ResidentKey key{bkFFT ? (const void *)bkFFT : (const void *)ks, (const void *)ks, n, N, l, Bgbit};
    const uint64_t fp = key_fingerprint(bkFFT, bkFFT ? n : 0, N, l, ks);
    auto it = reg.find(key);
    const std::vector<int> want = device_list().empty() ? std::vector<int>(1, device_ordinal()) : device_list();
    if (it != reg.end()) {
        if (it->second.fingerprint == fp && it->second.devices == want) return it->second;
        tfhe_amd_pool_destroy(it->second.pool);  // the key was rebuilt in place, or the device list changed
        reg.erase(it);
    }
    PoolResident R;
    R.owner = owner;
    R.fingerprint = fp;
    R.devices = want;
    R.n = n;
    R.N = N;
    tfhe_amd_params p;
    std::memset(&p, 0, sizeof(p));
    p.torus_bits = 32;
    p.n = n;
    p.N = N;
    p.k = 1;
    p.l = l;
    p.Bgbit = Bgbit;
    if (ks) {
        p.ks_t = ks->t;
        p.ks_basebit = ks->basebit;
        p.ks_n_out = ks->out_params->n;
    }
    pool_check(tfhe_amd_pool_create(&p, R.devices.data(), (int)R.devices.size(), &R.pool), nullptr, "tfhe_amd_pool_create");
    {
        std::vector<double> flat;
        if (bkFFT) flat = flatten_bk(bkFFT, n, N, l);
        std::vector<int32_t> kflat;
        if (ks) kflat = flatten_ks(ks);
        pool_check(tfhe_amd_pool_load_keys(R.pool, bkFFT ? flat.data() : nullptr, ks ? kflat.data() : nullptr), R.pool, "tfhe_amd_pool_load_keys");
    }
    return reg.emplace(key, std::move(R)).first->second;
}
// flatten n TGswSampleFFT (pointer-rich) into [n][2l][2][N] doubles and upload
template <class GswT>
inline Resident &attach_gsw(const GswT *bkFFT, int n, int N, int l, int Bgbit, const LweKeySwitchKey *ks = nullptr,
                            const void *owner = nullptr) {
    auto &reg = registry();
    const ResidentKey key{(const void *)bkFFT, (const void *)ks, n, N, l, Bgbit};
    const uint64_t fp = key_fingerprint(bkFFT, n, N, l, ks);
    auto it = reg.find(key);
    if (it != reg.end()) {
        if (it->second.fingerprint == fp) return it->second;
        release_entry(it);  // same address, other contents: the caller rebuilt the key in place
    }
    Resident R;
    R.owner = owner;
    R.fingerprint = fp;
    R.n = n;
    R.N = N;
    R.l = l;
    tfhe_amd_params p;
    std::memset(&p, 0, sizeof(p));
    p.torus_bits = 32;
    p.n = n;
    p.N = N;
    p.k = 1;
    p.l = l;
    p.Bgbit = Bgbit;
    if (ks) {
        p.ks_t = ks->t;
        p.ks_basebit = ks->basebit;
        p.ks_n_out = ks->out_params->n;
    }
    check(tfhe_amd_ctx_create(&p, device_ordinal(), &R.ctx), nullptr, "tfhe_amd_ctx_create");
    {
        const std::vector<double> flat = flatten_bk(bkFFT, n, N, l);
        check(tfhe_amd_gsw_from_fft(R.ctx, flat.data(), n, &R.gsw), R.ctx, "tfhe_amd_gsw_from_fft");
    }
    check(tfhe_amd_set_bootstrap_key(R.ctx, R.gsw), R.ctx, "tfhe_amd_set_bootstrap_key");
    if (ks) {
        const std::vector<int32_t> kflat = flatten_ks(ks);
        check(tfhe_amd_load_keyswitch_key(R.ctx, kflat.data()), R.ctx, "tfhe_amd_load_keyswitch_key");
    }
    staging(R, sizeof(int32_t) * (size_t)(2 * N + n + 2));
    return reg.emplace(key, R).first->second;
}
// staging of the array forms: one gather into PINNED host memory, one copy each way
inline void array_staging(tfhe_amd_ctx *ctx, void *&h_in, void *&da_in, size_t &have_in, size_t in_bytes, void *&h_out,
                          void *&da_out, size_t &have_out, size_t out_bytes) {
    // grown GEOMETRICALLY (at least 1 MiB, at least twice the previous size): the coalesced one-sample calls arrive in launches
    // of varying size, and a pinned + device reallocation at every new maximum costs milliseconds each (the frees synchronise)
    auto grown = [](size_t have, size_t need) {
        size_t want = have * 2 > need ? have * 2 : need;
        return want > ((size_t)1 << 20) ? want : ((size_t)1 << 20);
    };
    if (have_in < in_bytes) {
        const size_t want = grown(have_in, in_bytes);
        if (h_in) tfhe_amd_host_free(ctx, h_in);
        if (da_in) tfhe_amd_free(ctx, da_in);
        check(tfhe_amd_host_alloc(ctx, &h_in, want), ctx, "host_alloc");
        check(tfhe_amd_malloc(ctx, &da_in, want), ctx, "malloc");
        have_in = want;
    }
    if (have_out < out_bytes) {
        const size_t want = grown(have_out, out_bytes);
        if (h_out) tfhe_amd_host_free(ctx, h_out);
        if (da_out) tfhe_amd_free(ctx, da_out);
        check(tfhe_amd_host_alloc(ctx, &h_out, want), ctx, "host_alloc");
        check(tfhe_amd_malloc(ctx, &da_out, want), ctx, "malloc");
        have_out = want;
    }
}
inline void array_staging(Resident &R, size_t in_bytes, size_t out_bytes) {
    array_staging(R.ctx, R.h_in, R.da_in, R.arr_in_bytes, in_bytes, R.h_out, R.da_out, R.arr_out_bytes, out_bytes);
}
inline void release_entry(std::map<ResidentKey, Resident>::iterator it) {
    Resident &R = it->second;
    if (R.h_in) tfhe_amd_host_free(R.ctx, R.h_in);
    if (R.h_out) tfhe_amd_host_free(R.ctx, R.h_out);
    if (R.da_in) tfhe_amd_free(R.ctx, R.da_in);
    if (R.da_out) tfhe_amd_free(R.ctx, R.da_out);
    tfhe_amd_free(R.ctx, R.d_in);
    tfhe_amd_free(R.ctx, R.d_out);
    tfhe_amd_free(R.ctx, R.d_aux);
    if (R.gsw) tfhe_amd_gsw_free(R.gsw);
    tfhe_amd_ctx_destroy(R.ctx);
    registry().erase(it);
}
// every resident engine made from this key object, whatever shape it was attached with and whichever of its
// addresses the caller passes: the LweBootstrappingKeyFFT, its bkFFT array or its key-switch key.  (The copies are
// keyed by ADDRESS; the content sample of key_fingerprint catches most keys rebuilt in place, not all: release() is the contract.)
inline void release(const void *key_object) {
    TFHE_AMD_SHIM_GUARD();
    auto &reg = registry();
    std::vector<const void *> gone(1, key_object);  // every address this key was seen under: their coalescers go too
    for (auto it = reg.begin(); it != reg.end();) {
        auto cur = it++;
        if (cur->first.obj == key_object || cur->first.ks == key_object || cur->second.owner == key_object) {
            gone.push_back(cur->first.obj);
            gone.push_back(cur->first.ks);
            gone.push_back(cur->second.owner);
            release_entry(cur);
        }
    }
    for (const void *g : gone)
        if (g) lwe_coalescers_drop(g);
    auto &pools = pool_registry();
    for (auto it = pools.begin(); it != pools.end();) {
        auto cur = it++;
        if (cur->first.obj == key_object || cur->first.ks == key_object || cur->second.owner == key_object) {
            tfhe_amd_pool_destroy(cur->second.pool);
            pools.erase(cur);
        }
    }
}
inline void release_all() {
    TFHE_AMD_SHIM_GUARD();
    while (!registry().empty()) release_entry(registry().begin());
    for (auto &e : pool_registry()) tfhe_amd_pool_destroy(e.second.pool);
    pool_registry().clear();
    lwe_coalescers_drop(nullptr);
}

inline Resident &attach(const LweBootstrappingKeyFFT *bk) {
    return attach_gsw(bk->bkFFT, bk->in_out_params->n, bk->accum_params->N, bk->bk_params->l, bk->bk_params->Bgbit,
                      bk->ks, bk);
}

// ---- library-form entry points ----------------------------------------------------------------
inline void put_lwe(Resident &R, void *dst_d, const LweSample *x, int n) {
    std::vector<int32_t> f((size_t)n + 1);
    std::memcpy(f.data(), x->a, sizeof(int32_t) * (size_t)n);
    f[n] = x->b;
    check(tfhe_amd_memcpy_h2d(R.ctx, dst_d, f.data(), f.size() * 4), R.ctx, "h2d");
}
inline void get_lwe(Resident &R, LweSample *res, const void *src_d, int n) {
    std::vector<int32_t> f((size_t)n + 1);
    check(tfhe_amd_memcpy_d2h(R.ctx, f.data(), src_d, f.size() * 4), R.ctx, "d2h");
    std::memcpy(res->a, f.data(), sizeof(int32_t) * (size_t)n);
    res->b = f[n];
}
inline void put_tlwe(Resident &R, void *dst_d, const TLweSample *s, int N) {
    std::vector<int32_t> f((size_t)2 * N);
    for (int q = 0; q < 2; q++) std::memcpy(&f[(size_t)q * N], s->a[q].coefsT, sizeof(int32_t) * (size_t)N);
    check(tfhe_amd_memcpy_h2d(R.ctx, dst_d, f.data(), f.size() * 4), R.ctx, "h2d");
}
inline void get_tlwe(Resident &R, TLweSample *s, const void *src_d, int N) {
    std::vector<int32_t> f((size_t)2 * N);
    check(tfhe_amd_memcpy_d2h(R.ctx, f.data(), src_d, f.size() * 4), R.ctx, "d2h");
    for (int q = 0; q < 2; q++) std::memcpy(s->a[q].coefsT, &f[(size_t)q * N], sizeof(int32_t) * (size_t)N);
}

inline void tfhe_bootstrap_woKS_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                          const LweSample *const *xs, int count);
inline void tfhe_bootstrap_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                     const LweSample *const *xs, int count);
inline void lweKeySwitch_array(LweSample *const *results, const LweKeySwitchKey *ks, const LweSample *const *samples, int count);
// a coalesced batch of one-sample calls: one call -> `one` (the one-sample path), several -> `many` (the array form)
template <class One, class Many>
inline void run_lwe_calls(const std::vector<LweCall> &calls, One one, Many many) {
    if (calls.size() == 1) {
        one(calls[0].result, calls[0].x);
        return;
    }
    std::vector<LweSample *> rs(calls.size());
    std::vector<const LweSample *> xs(calls.size());
    for (size_t i = 0; i < calls.size(); i++) {
        rs[i] = calls[i].result;
        xs[i] = calls[i].x;
    }
    many(rs.data(), xs.data(), (int)calls.size());
}
inline void tfhe_bootstrap_woKS_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x) {
    lwe_coalescer(0, bk)->call(LweCall{result, x, mu}, [&](const std::vector<LweCall> &batch) {
      for_each_mu_group(batch, [&](Torus32 mu, const std::vector<LweCall> &calls) {  // (shadows the leader's own mu: a group's)
        run_lwe_calls(calls,
                      [&](LweSample *r, const LweSample *in) {
                          TFHE_AMD_SHIM_GUARD();
                          Resident &R = attach(bk);
                          put_lwe(R, R.d_in, in, R.n);
                          check(tfhe_amd_bootstrap_woks(R.ctx, (int32_t *)R.d_out, mu, (const int32_t *)R.d_in, 1), R.ctx, "bootstrap_woks");
                          get_lwe(R, r, R.d_out, R.N);
                      },
                      [&](LweSample *const *rs, const LweSample *const *xs, int n) { ::tfhe_amd_compat::tfhe_bootstrap_woKS_FFT_array(rs, bk, mu, xs, n); });
      });
    });
}
inline void tfhe_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, Torus32 mu, const LweSample *x) {
    lwe_coalescer(1, bk)->call(LweCall{result, x, mu}, [&](const std::vector<LweCall> &batch) {
      for_each_mu_group(batch, [&](Torus32 mu, const std::vector<LweCall> &calls) {
        run_lwe_calls(calls,
                      [&](LweSample *r, const LweSample *in) {
                          TFHE_AMD_SHIM_GUARD();
                          Resident &R = attach(bk);
                          put_lwe(R, R.d_in, in, R.n);
                          check(tfhe_amd_bootstrap(R.ctx, (int32_t *)R.d_out, mu, (const int32_t *)R.d_in, 1), R.ctx, "bootstrap");
                          get_lwe(R, r, R.d_out, R.n);
                      },
                      [&](LweSample *const *rs, const LweSample *const *xs, int n) { ::tfhe_amd_compat::tfhe_bootstrap_FFT_array(rs, bk, mu, xs, n); });
      });
    });
}
inline void tfhe_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bkFFT, const int *bara, const int n,
                                 const TGswParams *bk_params) {
    TFHE_AMD_SHIM_GUARD();
    Resident &R = attach_gsw(bkFFT, n, bk_params->tlwe_params->N, bk_params->l, bk_params->Bgbit);
    put_tlwe(R, R.d_in, accum, R.N);
    check(tfhe_amd_memcpy_h2d(R.ctx, R.d_aux, bara, sizeof(int) * (size_t)n), R.ctx, "h2d");
    check(tfhe_amd_blind_rotate(R.ctx, R.d_in, (const int32_t *)R.d_aux, 1), R.ctx, "blind_rotate");
    get_tlwe(R, accum, R.d_in, R.N);
}
inline void tfhe_blindRotateAndExtract_FFT(LweSample *result, const TorusPolynomial *v, const TGswSampleFFT *bk,
                                           const int barb, const int *bara, const int n, const TGswParams *bk_params) {
    TFHE_AMD_SHIM_GUARD();
    Resident &R = attach_gsw(bk, n, bk_params->tlwe_params->N, bk_params->l, bk_params->Bgbit);
    std::vector<int32_t> rot((size_t)n + 1);
    std::memcpy(rot.data(), bara, sizeof(int) * (size_t)n);
    rot[n] = barb;
    check(tfhe_amd_memcpy_h2d(R.ctx, R.d_aux, rot.data(), rot.size() * 4), R.ctx, "h2d");
    check(tfhe_amd_memcpy_h2d(R.ctx, R.d_in, v->coefsT, sizeof(int32_t) * (size_t)R.N), R.ctx, "h2d");
    check(tfhe_amd_blind_rotate_extract(R.ctx, R.d_out, R.d_in, 0, (const int32_t *)R.d_aux, 1), R.ctx, "bre");
    get_lwe(R, result, R.d_out, R.N);
}
inline void tGswFFTExternMulToTLwe(TLweSample *accum, const TGswSampleFFT *gsw, const TGswParams *params) {
    TFHE_AMD_SHIM_GUARD();
    Resident &R = attach_gsw(gsw, 1, params->tlwe_params->N, params->l, params->Bgbit);
    put_tlwe(R, R.d_in, accum, R.N);
    check(tfhe_amd_extern_mul(R.ctx, R.d_in, R.gsw, 0, 1), R.ctx, "extern_mul");
    get_tlwe(R, accum, R.d_in, R.N);
}
// tfhe_MuxRotate_FFT (CB/lwe_functions.cpp:328-333): result = bki (x) ((X^barai - 1) accum) + accum -- one CMux step of the blind
// rotation on its own; result and accum may be the same object (the reference's callers ping-pong two samples)
inline void tfhe_MuxRotate_FFT(TLweSample *result, const TLweSample *accum, const TGswSampleFFT *bki, const int barai,
                               const TGswParams *bk_params) {
    TFHE_AMD_SHIM_GUARD();
    Resident &R = attach_gsw(bki, 1, bk_params->tlwe_params->N, bk_params->l, bk_params->Bgbit);
    put_tlwe(R, R.d_in, accum, R.N);
    const int32_t a = (int32_t)barai;
    check(tfhe_amd_memcpy_h2d(R.ctx, R.d_aux, &a, sizeof(a)), R.ctx, "h2d");
    check(tfhe_amd_mux_rotate(R.ctx, R.d_in, R.gsw, 0, (const int32_t *)R.d_aux, 1), R.ctx, "mux_rotate");
    get_tlwe(R, result, R.d_in, R.N);
}
// lweKeySwitch(result, ks, sample): a key-switch key seen on its own gets its own resident engine
// (input dimension ks->n must be a ring degree the engine supports: a power of two in [16, 2^20])
inline Resident &attach_ks(const LweKeySwitchKey *ks) {
    auto &reg = registry();
    const ResidentKey key{(const void *)ks, (const void *)ks, ks->out_params->n, ks->n, 0, 0};
    const uint64_t fp = key_fingerprint((const TGswSampleFFT *)nullptr, 0, 0, 0, ks);
    auto it = reg.find(key);
    if (it != reg.end() && it->second.fingerprint != fp) {
        release_entry(it);  // rebuilt in place
        it = reg.end();
    }
    if (it == reg.end()) {
        Resident R;
        R.fingerprint = fp;
        R.n = ks->out_params->n;
        R.N = ks->n;
        tfhe_amd_params p;
        std::memset(&p, 0, sizeof(p));
        p.torus_bits = 32;
        p.n = R.n;
        p.N = R.N;
        p.k = 1;
        p.l = 1;
        p.Bgbit = 1;
        p.ks_t = ks->t;
        p.ks_basebit = ks->basebit;
        p.ks_n_out = R.n;
        check(tfhe_amd_ctx_create(&p, device_ordinal(), &R.ctx), nullptr, "tfhe_amd_ctx_create");
        const std::vector<int32_t> kflat = flatten_ks(ks);
        check(tfhe_amd_load_keyswitch_key(R.ctx, kflat.data()), R.ctx, "tfhe_amd_load_keyswitch_key");
        staging(R, sizeof(int32_t) * (size_t)(2 * R.N + R.n + 2));
        it = reg.emplace(key, R).first;
    }
    return it->second;
}
inline void lweKeySwitch(LweSample *result, const LweKeySwitchKey *ks, const LweSample *sample) {
    lwe_coalescer(2, ks)->call(LweCall{result, sample, 0}, [&](const std::vector<LweCall> &calls) {
        run_lwe_calls(calls,
                      [&](LweSample *r, const LweSample *in) {
                          TFHE_AMD_SHIM_GUARD();
                          Resident &R = attach_ks(ks);
                          put_lwe(R, R.d_in, in, R.N);
                          check(tfhe_amd_keyswitch(R.ctx, (int32_t *)R.d_out, (const int32_t *)R.d_in, 1), R.ctx, "keyswitch");
                          get_lwe(R, r, R.d_out, R.n);
                      },
                      [&](LweSample *const *rs, const LweSample *const *xs, int n) { ::tfhe_amd_compat::lweKeySwitch_array(rs, ks, xs, n); });
    });
}

// ---- array forms behind the reference names ----------------------------------------------------
// The reference's unit is ONE sample per call (lwe_functions.cpp:434-446) and its drivers loop over samples
// (poc:1009-1013; the only parallel construct: parallel/src/test_parallel_multiplications.cpp:62).  These take the
// loop's `count` independent samples at once: one gather into a pinned staging buffer, one copy in, ONE launch, one
// copy out, one scatter -- the same results as `count` one-sample calls, bit for bit, at the batch engine's rate.
// results[c] / xs[c] are the loop's own objects (caller-allocated, as for the one-sample functions).
inline void gather_lwe(int32_t *dst, const LweSample *const *xs, int count, int n) {
    for (int c = 0; c < count; c++) {
        std::memcpy(dst + (size_t)c * (n + 1), xs[c]->a, sizeof(int32_t) * (size_t)n);
        dst[(size_t)c * (n + 1) + n] = xs[c]->b;
    }
}
inline void scatter_lwe(LweSample *const *results, const int32_t *src, int count, int n) {
    for (int c = 0; c < count; c++) {
        std::memcpy(results[c]->a, src + (size_t)c * (n + 1), sizeof(int32_t) * (size_t)n);
        results[c]->b = src[(size_t)c * (n + 1) + n];
    }
}
template <class Launch>
inline void lwe_array_call(Resident &R, LweSample *const *results, int n_out, const LweSample *const *xs, int n_in, int count,
                           Launch launch, const char *what) {
    if (count <= 0) return;
    const size_t in_bytes = sizeof(int32_t) * (size_t)count * (n_in + 1), out_bytes = sizeof(int32_t) * (size_t)count * (n_out + 1);
    array_staging(R, in_bytes, out_bytes);
    gather_lwe((int32_t *)R.h_in, xs, count, n_in);
    check(tfhe_amd_memcpy_h2d(R.ctx, R.da_in, R.h_in, in_bytes), R.ctx, "h2d");
    check(launch((int32_t *)R.da_out, (const int32_t *)R.da_in), R.ctx, what);
    check(tfhe_amd_memcpy_d2h(R.ctx, R.h_out, R.da_out, out_bytes), R.ctx, "d2h");
    scatter_lwe(results, (const int32_t *)R.h_out, count, n_out);
}
// the same call through a pool: the members gather the caller's samples straight into their pinned staging buffers and scatter
// the results from them (tfhe_amd_pool_*_rows) -- no flat intermediate array, and for a long loop the gather of chunk k + 1 and
// the scatter of chunk k - 1 run while chunk k computes
struct LweRows {
    LweSample *const *results;
    const LweSample *const *xs;
    int n_in, n_out;
};
inline void lwe_rows_in(void *user, int first, int rows, int32_t *dst) {
    const LweRows *r = static_cast<const LweRows *>(user);
    gather_lwe(dst, r->xs + first, rows, r->n_in);
}
inline void lwe_rows_out(void *user, int first, int rows, const int32_t *src) {
    const LweRows *r = static_cast<const LweRows *>(user);
    scatter_lwe(r->results + first, src, rows, r->n_out);
}
template <class Call>
inline void lwe_array_pool_call(PoolResident &R, LweSample *const *results, int n_out, const LweSample *const *xs, int n_in, int /*count*/,
                                Call call, const char *what) {
    LweRows rows{results, xs, n_in, n_out};
    pool_check(call(&rows), R.pool, what);
}
inline PoolResident &attach_pool(const LweBootstrappingKeyFFT *bk) {
    return attach_pool(bk->bkFFT, bk->in_out_params->n, bk->accum_params->N, bk->bk_params->l, bk->bk_params->Bgbit, bk->ks, bk);
}
// Which array calls go through a pool: all of them once several devices are named; on ONE device those long enough for the pool's
// pipelined form (>= 4096 samples: copies, gather and scatter hidden behind the kernels) -- shorter ones keep the resident
// engine's one-piece path (no second key copy on the device for callers that never hand over long loops)
#ifndef TFHE_AMD_COMPAT_POOL_MIN
#define TFHE_AMD_COMPAT_POOL_MIN 4096
#endif
constexpr int POOL_MIN_COUNT = TFHE_AMD_COMPAT_POOL_MIN;
inline bool use_pool(int count = 0) { return device_list().size() > 1 || count >= POOL_MIN_COUNT; }
inline void tfhe_bootstrap_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                     const LweSample *const *xs, int count) {
    TFHE_AMD_SHIM_GUARD();
    if (count <= 0) return;
    if (use_pool(count)) {
        PoolResident &P = attach_pool(bk);
        lwe_array_pool_call(P, results, P.n, xs, P.n, count,
                            [&](LweRows *r) { return tfhe_amd_pool_bootstrap_rows(P.pool, lwe_rows_out, lwe_rows_in, r, mu, count); }, "bootstrap (array, pool)");
        return;
    }
    Resident &R = attach(bk);
    lwe_array_call(R, results, R.n, xs, R.n, count,
                   [&](int32_t *o, const int32_t *i) { return tfhe_amd_bootstrap(R.ctx, o, mu, i, count); }, "bootstrap (array)");
}
inline void tfhe_bootstrap_woKS_FFT_array(LweSample *const *results, const LweBootstrappingKeyFFT *bk, Torus32 mu,
                                          const LweSample *const *xs, int count) {
    TFHE_AMD_SHIM_GUARD();
    if (count <= 0) return;
    if (use_pool(count)) {
        PoolResident &P = attach_pool(bk);
        lwe_array_pool_call(P, results, P.N, xs, P.n, count,
                            [&](LweRows *r) { return tfhe_amd_pool_bootstrap_woks_rows(P.pool, lwe_rows_out, lwe_rows_in, r, mu, count); },
                            "bootstrap_woks (array, pool)");
        return;
    }
    Resident &R = attach(bk);
    lwe_array_call(R, results, R.N, xs, R.n, count,
                   [&](int32_t *o, const int32_t *i) { return tfhe_amd_bootstrap_woks(R.ctx, o, mu, i, count); }, "bootstrap_woks (array)");
}
inline void lweKeySwitch_array(LweSample *const *results, const LweKeySwitchKey *ks, const LweSample *const *samples, int count) {
    TFHE_AMD_SHIM_GUARD();
    if (count <= 0) return;
    if (use_pool(count)) {  // a key-switch key on its own: a pool of contexts that hold nothing else (gadget unused: l = Bgbit = 1)
        PoolResident &P = attach_pool((const TGswSampleFFT *)nullptr, ks->out_params->n, ks->n, 1, 1, ks, ks);
        lwe_array_pool_call(P, results, P.n, samples, P.N, count,
                            [&](LweRows *r) { return tfhe_amd_pool_keyswitch_rows(P.pool, lwe_rows_out, lwe_rows_in, r, count); }, "keyswitch (array, pool)");
        return;
    }
    Resident &R = attach_ks(ks);
    lwe_array_call(R, results, R.n, samples, R.N, count,
                   [&](int32_t *o, const int32_t *i) { return tfhe_amd_keyswitch(R.ctx, o, i, count); }, "keyswitch (array)");
}

// ---- FFT plugin look-alike (CB/spqlios/lagrangehalfc_impl.h:8-31) -----------------------------
class FFT_Processor_AMD {
   public:
    const int _2N, N, Ns2;
    explicit FFT_Processor_AMD(int N_) : _2N(2 * N_), N(N_), Ns2(N_ / 2), ctx_(nullptr), a_(nullptr), b_(nullptr), c_(nullptr) {
        tfhe_amd_params p;
        std::memset(&p, 0, sizeof(p));
        p.torus_bits = 64;  // the width only selects gadget defaults; all four conversions are available
        p.n = 1;
        p.N = N_;
        p.k = 1;
        p.l = 1;
        p.Bgbit = 1;
        check(tfhe_amd_ctx_create(&p, device_ordinal(), &ctx_), nullptr, "tfhe_amd_ctx_create");
        check(tfhe_amd_malloc(ctx_, &a_, 8 * (size_t)N), ctx_, "malloc");
        check(tfhe_amd_malloc(ctx_, &b_, 8 * (size_t)N), ctx_, "malloc");
        check(tfhe_amd_malloc(ctx_, &c_, 8 * (size_t)N), ctx_, "malloc");
    }
    ~FFT_Processor_AMD() {
        tfhe_amd_free(ctx_, a_);
        tfhe_amd_free(ctx_, b_);
        tfhe_amd_free(ctx_, c_);
        tfhe_amd_ctx_destroy(ctx_);
    }
    void execute_reverse_int(double *res, const int *a) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(ctx_, a_, a, 4 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_ifft_int32(ctx_, (double *)b_, (const int32_t *)a_, 1), ctx_, "ifft");
        check(tfhe_amd_memcpy_d2h(ctx_, res, b_, 8 * (size_t)N), ctx_, "d2h");
    }
    void execute_reverse_torus32(double *res, const int32_t *a) { execute_reverse_int(res, (const int *)a); }
    void execute_direct_torus32(int32_t *res, const double *a) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(ctx_, a_, a, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_fft_torus32(ctx_, (int32_t *)b_, (const double *)a_, 1), ctx_, "fft");
        check(tfhe_amd_memcpy_d2h(ctx_, res, b_, 4 * (size_t)N), ctx_, "d2h");
    }
    void execute_reverse_torus64(double *res, const int64_t *a) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(ctx_, a_, a, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_ifft_torus64(ctx_, (double *)b_, (const int64_t *)a_, 1), ctx_, "ifft");
        check(tfhe_amd_memcpy_d2h(ctx_, res, b_, 8 * (size_t)N), ctx_, "d2h");
    }
    void execute_direct_torus64(int64_t *res, const double *a) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(ctx_, a_, a, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_fft_torus64(ctx_, (int64_t *)b_, (const double *)a_, 1), ctx_, "fft");
        check(tfhe_amd_memcpy_d2h(ctx_, res, b_, 8 * (size_t)N), ctx_, "d2h");
    }
    // LagrangeHalfCPolynomialAddMulASM(res, a, b, Ns2)
    void AddMul(double *res, const double *a, const double *b) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(ctx_, c_, res, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_memcpy_h2d(ctx_, a_, a, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_memcpy_h2d(ctx_, b_, b, 8 * (size_t)N), ctx_, "h2d");
        check(tfhe_amd_lagrange_addmul(ctx_, (double *)c_, (const double *)a_, (const double *)b_, 1, 0), ctx_, "addmul");
        check(tfhe_amd_memcpy_d2h(ctx_, res, c_, 8 * (size_t)N), ctx_, "d2h");
    }

   private:
    FFT_Processor_AMD(const FFT_Processor_AMD &);
    FFT_Processor_AMD &operator=(const FFT_Processor_AMD &);
    tfhe_amd_ctx *ctx_;
    void *a_, *b_, *c_;
};

// ---- PoC form: works on the reference's own poc_types.h structs (duck-typed template) ----------
// GlobalsT needs: n_lvl0, n_lvl1, n_lvl2, bgbit_lvl1, ell_lvl1, bgbit_lvl2, ell_lvl2, kslength_lvl10,
// ksbasebit_lvl10, kslength_lvl21, ksbasebit_lvl21, preKS[i][j][u].a[h], bkFFT[i].allsamples[r].a[q].values,
// privKS[u][i][j][d].a[q].coefs[p]   (CB/poc_types.h:267-312)
template <class GlobalsT>
class PocEngine {
   public:
    explicit PocEngine(const GlobalsT *env, int device = 0)
        : env_(env), cb_(nullptr), own_pool_(nullptr), pool_(nullptr) {
        tfhe_amd_cb_params p;
        p.n0 = env->n_lvl0;
        p.N1 = env->n_lvl1;
        p.N2 = env->n_lvl2;
        p.l1 = env->ell_lvl1;
        p.Bgbit1 = env->bgbit_lvl1;
        p.l2 = env->ell_lvl2;
        p.Bgbit2 = env->bgbit_lvl2;
        p.t10 = env->kslength_lvl10;
        p.bb10 = env->ksbasebit_lvl10;
        p.t21 = env->kslength_lvl21;
        p.bb21 = env->ksbasebit_lvl21;
        p_ = p;
        // the engine's own handle is the one member of a pool of its own: the one-sample entry points use the member's handle
        // directly (under the shims' lock), the array form goes through the pool and gets its pipelined member -- one copy of the
        // 2.7 GB of keys serves both
        own_pool_ = make_pool(std::vector<int>(1, device));
        cb_ = tfhe_amd_cb_pool_member(own_pool_, 0);
        c2_ = tfhe_amd_cb_ctx_lvl2(cb_);
        c10_ = tfhe_amd_cb_ctx_lvl10(cb_);
        const size_t big = sizeof(int32_t) * (size_t)2 * p.l1 * 2 * p.N1 + sizeof(int64_t) * (size_t)(p.N2 + 1);
        check(tfhe_amd_malloc(c2_, &d_a_, big), c2_, "malloc");
        check(tfhe_amd_malloc(c2_, &d_b_, big), c2_, "malloc");
    }
    ~PocEngine() {
        if (pool_) tfhe_amd_cb_pool_destroy(pool_);
        if (cb_) {
            tfhe_amd_free(c2_, d_a_);
            tfhe_amd_free(c2_, d_b_);
        }
        if (own_pool_) tfhe_amd_cb_pool_destroy(own_pool_);  // destroys cb_, its member
    }
    // tfhe_CircuitBootstrapFFT(TGswSample32* result, const LweSample32* sample, env)   poc:823-873
    // Calls from several host threads are coalesced into one tfhe_CircuitBootstrapFFT_array launch (Coalescer above).
    template <class TGswSample32T, class LweSample32T>
    void tfhe_CircuitBootstrapFFT(TGswSample32T *result, const LweSample32T *sample) {
        cb_calls_.call(std::make_pair((void *)result, (const void *)sample), [&](const std::vector<std::pair<void *, const void *>> &calls) {
            if (calls.size() == 1) {
                tfhe_CircuitBootstrapFFT_one((TGswSample32T *)calls[0].first, (const LweSample32T *)calls[0].second);
                return;
            }
            std::vector<TGswSample32T *> rs(calls.size());
            std::vector<const LweSample32T *> xs(calls.size());
            for (size_t i = 0; i < calls.size(); i++) {
                rs[i] = (TGswSample32T *)calls[i].first;
                xs[i] = (const LweSample32T *)calls[i].second;
            }
            tfhe_CircuitBootstrapFFT_array(rs.data(), xs.data(), (int)calls.size());
        });
    }
    template <class TGswSample32T, class LweSample32T>
    void tfhe_CircuitBootstrapFFT_one(TGswSample32T *result, const LweSample32T *sample) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(c2_, d_a_, sample->a, sizeof(int32_t) * (size_t)(p_.N1 + 1)), c2_, "h2d");
        die(tfhe_amd_circuit_bootstrap(cb_, (int32_t *)d_b_, (const int32_t *)d_a_, 1), "circuit_bootstrap");
        std::vector<int32_t> f((size_t)2 * p_.l1 * 2 * p_.N1);
        check(tfhe_amd_memcpy_d2h(c2_, f.data(), d_b_, f.size() * 4), c2_, "d2h");
        for (int u = 0; u < 2; u++)
            for (int w = 0; w < p_.l1; w++)
                for (int q = 0; q < 2; q++)
                    std::memcpy(result->samples[u][w].a[q].coefs, &f[(((size_t)u * p_.l1 + w) * 2 + q) * p_.N1],
                                sizeof(int32_t) * (size_t)p_.N1);
    }
    // The driver loop of poc:1009-1013 over `count` samples as ONE call: the pool's member(s) gather the inputs from the caller's
    // LweSample32 objects into pinned staging, run tfhe_amd_circuit_bootstrap on the batch (in pipelined chunks of 1024 when it is
    // long) and scatter the results into the caller's TGswSample32 objects.  With several devices named (set_devices) the pool
    // spans them; otherwise it is the engine's own one-member pool.
    template <class TGswSample32T, class LweSample32T>
    void tfhe_CircuitBootstrapFFT_array(TGswSample32T *const *results, const LweSample32T *const *samples, int count) {
        TFHE_AMD_SHIM_GUARD();
        if (count <= 0) return;
        tfhe_amd_cb_pool *pool = own_pool_;
        if (use_pool()) {
            ensure_pool();
            pool = pool_;
        }
        CbRows<TGswSample32T, LweSample32T> rows = {this, results, samples};
        const int rc = tfhe_amd_cb_pool_circuit_bootstrap_rows(pool, &CbRows<TGswSample32T, LweSample32T>::put, &CbRows<TGswSample32T, LweSample32T>::get, &rows, count);
        if (rc != TFHE_AMD_OK) {
            std::fprintf(stderr, "tfhe_amd: circuit_bootstrap (array) failed (%d): %s\n", rc, tfhe_amd_cb_pool_last_error(pool));
            std::abort();
        }
    }
    // circuitBootstrapWoKS(LweSample64* result, Torus64 mu, const int* abar, env)   poc:530-659
    template <class LweSample64T>
    void circuitBootstrapWoKS(LweSample64T *result, const Torus64 mu, const int *abar) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(c2_, d_a_, abar, sizeof(int) * (size_t)(p_.n0 + 1)), c2_, "h2d");
        check(tfhe_amd_cb_bootstrap_woks(c2_, (int64_t *)d_b_, mu, (const int32_t *)d_a_, 1), c2_, "cb_bootstrap_woks");
        check(tfhe_amd_memcpy_d2h(c2_, result->a, d_b_, sizeof(int64_t) * (size_t)(p_.N2 + 1)), c2_, "d2h");
    }
    // circuitPrivKS(TLweSample32* result, int u, const LweSample64* x, env)   poc:667-698
    template <class TLweSample32T, class LweSample64T>
    void circuitPrivKS(TLweSample32T *result, const int u, const LweSample64T *x) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(c2_, d_a_, x->a, sizeof(int64_t) * (size_t)(p_.N2 + 1)), c2_, "h2d");
        die(tfhe_amd_privks(cb_, (int32_t *)d_b_, u, (const int64_t *)d_a_, 1), "privks");
        std::vector<int32_t> f((size_t)2 * p_.N1);
        check(tfhe_amd_memcpy_d2h(c2_, f.data(), d_b_, f.size() * 4), c2_, "d2h");
        for (int q = 0; q < 2; q++) std::memcpy(result->a[q].coefs, &f[(size_t)q * p_.N1], sizeof(int32_t) * (size_t)p_.N1);
    }
    // preKeySwitch(LweSample32* result, const LweSample32* x, env)   poc:437-465
    template <class LweSample32T>
    void preKeySwitch(LweSample32T *result, const LweSample32T *x) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(c10_, d_a_, x->a, sizeof(int32_t) * (size_t)(p_.N1 + 1)), c10_, "h2d");
        check(tfhe_amd_keyswitch(c10_, (int32_t *)d_b_, (const int32_t *)d_a_, 1), c10_, "keyswitch");
        check(tfhe_amd_memcpy_d2h(c10_, result->a, d_b_, sizeof(int32_t) * (size_t)(p_.n0 + 1)), c10_, "d2h");
    }
    // preModSwitch(int* result, const LweSample32* x, env)   poc:472-484
    template <class LweSample32T>
    void preModSwitch(int *result, const LweSample32T *x) {
        TFHE_AMD_SHIM_GUARD();
        check(tfhe_amd_memcpy_h2d(c2_, d_a_, x->a, sizeof(int32_t) * (size_t)(p_.n0 + 1)), c2_, "h2d");
        check(tfhe_amd_modswitch(c2_, (int32_t *)d_b_, (const int32_t *)d_a_, 1), c2_, "modswitch");
        check(tfhe_amd_memcpy_d2h(c2_, result, d_b_, sizeof(int) * (size_t)(p_.n0 + 1)), c2_, "d2h");
    }
    // CMux(TLweSample32* out, const TGswSample32* c, const TLweSample32* in0, const TLweSample32* in1, env)
    // -- declared but left empty by the reference (poc:877-879).  out = c ? in1 : in0, i.e.
    // c (x) (in1 - in0) + in0 with c converted to Lagrange form on the device (tGswToFFTConvert).
    template <class TLweSample32T, class TGswSample32T>
    void CMux(TLweSample32T *out, const TGswSample32T *c, const TLweSample32T *in0, const TLweSample32T *in1) {
        TFHE_AMD_SHIM_GUARD();
        const size_t N1 = (size_t)p_.N1, rows = (size_t)2 * p_.l1;
        std::vector<int32_t> g(rows * 2 * N1), d(4 * N1);
        for (int u = 0; u < 2; u++)
            for (int w = 0; w < p_.l1; w++)
                for (int q = 0; q < 2; q++)
                    std::memcpy(&g[(((size_t)u * p_.l1 + w) * 2 + q) * N1], c->samples[u][w].a[q].coefs, sizeof(int32_t) * N1);
        for (int q = 0; q < 2; q++) {
            std::memcpy(&d[(size_t)q * N1], in0->a[q].coefs, sizeof(int32_t) * N1);
            std::memcpy(&d[(2 + (size_t)q) * N1], in1->a[q].coefs, sizeof(int32_t) * N1);
        }
        tfhe_amd_gsw *sel = nullptr;
        check(tfhe_amd_gsw_from_torus(c10_, g.data(), 1, &sel), c10_, "gsw_from_torus");
        check(tfhe_amd_memcpy_h2d(c10_, d_a_, d.data(), sizeof(int32_t) * d.size()), c10_, "h2d");
        int32_t *dd = (int32_t *)d_a_;
        check(tfhe_amd_cmux(c10_, d_b_, sel, nullptr, dd, dd + 2 * N1, 1), c10_, "cmux");
        check(tfhe_amd_memcpy_d2h(c10_, d.data(), d_b_, sizeof(int32_t) * 2 * N1), c10_, "d2h");
        tfhe_amd_gsw_free(sel);
        for (int q = 0; q < 2; q++) std::memcpy(out->a[q].coefs, &d[(size_t)q * N1], sizeof(int32_t) * N1);
    }
    tfhe_amd_cb *handle() { return cb_; }

   private:
    void die(int rc, const char *what) {
        if (rc != TFHE_AMD_OK) {
            std::fprintf(stderr, "tfhe_amd: %s failed (%d): %s\n", what, rc, cb_ ? tfhe_amd_cb_last_error(cb_) : "");
            std::abort();
        }
    }
    template <class TGswSample32T>
    void scatter_tgsw(TGswSample32T *const *results, const int32_t *ho, int count) {
        const size_t rout = (size_t)2 * p_.l1 * 2 * p_.N1;
        for (int c = 0; c < count; c++)
            for (int u = 0; u < 2; u++)
                for (int w = 0; w < p_.l1; w++)
                    for (int q = 0; q < 2; q++)
                        std::memcpy(results[c]->samples[u][w].a[q].coefs,
                                    ho + (size_t)c * rout + (((size_t)u * p_.l1 + w) * 2 + q) * p_.N1, sizeof(int32_t) * (size_t)p_.N1);
    }
    // Globals::preKS -> [N1][t10][base10][n0+1]
    std::vector<int32_t> flat_preks() const {
        const tfhe_amd_cb_params &p = p_;
        const int base10 = 1 << p.bb10;
        std::vector<int32_t> f((size_t)p.N1 * p.t10 * base10 * (p.n0 + 1));
        size_t o = 0;
        for (int i = 0; i < p.N1; i++)
            for (int j = 0; j < p.t10; j++)
                for (int u = 0; u < base10; u++, o += (size_t)p.n0 + 1)
                    std::memcpy(&f[o], env_->preKS[i][j][u].a, sizeof(int32_t) * (size_t)(p.n0 + 1));
        return f;
    }
    // Globals::bkFFT -> [n0][2 l2][2][N2]
    std::vector<double> flat_bkfft() const {
        const tfhe_amd_cb_params &p = p_;
        std::vector<double> f((size_t)p.n0 * 2 * p.l2 * 2 * p.N2);
        size_t o = 0;
        for (int i = 0; i < p.n0; i++)
            for (int r = 0; r < 2 * p.l2; r++)
                for (int q = 0; q < 2; q++, o += (size_t)p.N2)
                    std::memcpy(&f[o], env_->bkFFT[i].allsamples[r].a[q].values, sizeof(double) * (size_t)p.N2);
        return f;
    }
    // Globals::privKS plane u -> [N2+1][t21][base21][2][N1]
    std::vector<int32_t> flat_privks(int u) const {
        const tfhe_amd_cb_params &p = p_;
        const int base21 = 1 << p.bb21;
        std::vector<int32_t> f((size_t)(p.N2 + 1) * p.t21 * base21 * 2 * p.N1);
        size_t o = 0;
        for (int i = 0; i <= p.N2; i++)
            for (int j = 0; j < p.t21; j++)
                for (int d = 0; d < base21; d++)
                    for (int q = 0; q < 2; q++, o += (size_t)p.N1)
                        std::memcpy(&f[o], env_->privKS[u][i][j][d].a[q].coefs, sizeof(int32_t) * (size_t)p.N1);
        return f;
    }
    // rows of the array form between the caller's objects and a member's pinned staging buffer (called on the members' threads)
    template <class TGswSample32T, class LweSample32T>
    struct CbRows {
        PocEngine *self;
        TGswSample32T *const *results;
        const LweSample32T *const *samples;
        static void get(void *user, int first, int rows, int32_t *dst) {
            const CbRows *r = static_cast<const CbRows *>(user);
            const size_t rin = (size_t)r->self->p_.N1 + 1;
            for (int c = 0; c < rows; c++) std::memcpy(dst + (size_t)c * rin, r->samples[first + c]->a, sizeof(int32_t) * rin);
        }
        static void put(void *user, int first, int rows, const int32_t *src) {
            const CbRows *r = static_cast<const CbRows *>(user);
            r->self->scatter_tgsw(r->results + first, src, rows);
        }
    };
    // a pool over `devices` with this engine's keys (Globals::preKS, bkFFT, privKS flattened and uploaded once per member)
    tfhe_amd_cb_pool *make_pool(const std::vector<int> &devices) {
        tfhe_amd_cb_pool *pool = nullptr;
        auto pdie = [&](int rc, const char *what) {
            if (rc != TFHE_AMD_OK) {
                std::fprintf(stderr, "tfhe_amd: %s failed (%d): %s\n", what, rc, pool ? tfhe_amd_cb_pool_last_error(pool) : "");
                std::abort();
            }
        };
        pdie(tfhe_amd_cb_pool_create(&p_, devices.data(), (int)devices.size(), &pool), "tfhe_amd_cb_pool_create");
        {
            const std::vector<int32_t> f = flat_preks();
            pdie(tfhe_amd_cb_pool_load_preks(pool, f.data()), "load preKS");
        }
        {
            const std::vector<double> f = flat_bkfft();
            pdie(tfhe_amd_cb_pool_load_bk_fft(pool, f.data()), "load bkFFT");
        }
        for (int u = 0; u < 2; u++) {
            const std::vector<int32_t> f = flat_privks(u);
            pdie(tfhe_amd_cb_pool_load_privks_plane(pool, u, f.data()), "load privKS");
        }
        return pool;
    }
    // the multi-device pool behind the array form: built for the current device list, rebuilt when that list changes
    void ensure_pool() {
        if (pool_ && pool_devices_ == device_list()) return;
        if (pool_) tfhe_amd_cb_pool_destroy(pool_);
        pool_ = nullptr;
        pool_devices_ = device_list();
        pool_ = make_pool(pool_devices_);
    }
    const GlobalsT *env_;
    Coalescer<std::pair<void *, const void *> > cb_calls_;
    tfhe_amd_cb *cb_;                  // = the one member of own_pool_
    tfhe_amd_cb_pool *own_pool_, *pool_;  // this engine's device; the devices named by set_devices (array form only)
    std::vector<int> pool_devices_;
    tfhe_amd_cb_params p_;
    tfhe_amd_ctx *c2_, *c10_;
    void *d_a_, *d_b_;
};

}  // namespace tfhe_amd_compat
#pragma pop_macro("k")
#endif
