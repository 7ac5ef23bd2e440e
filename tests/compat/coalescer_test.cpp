// coalescer_test.cpp -- tfhe_amd_compat::Coalescer (include/tfhe_amd_compat.hpp) on its own, no engine behind it: many host
// threads call at once; every request must be carried by exactly one batch, batches must never overlap, a call must return only
// after its own request ran, and the lead must be handed over (no caller leads more than one batch per call).
// Built twice by tests/test_coalescer.py: plain, and with -fsanitize=thread (the class is all mutex / condition-variable code).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "tfhe_amd_compat.hpp"

struct Item {
    int id;
    int *slot;  // where the "result" of this request goes
};

// Registry mode (argv[1] == "registry"): the shims' coalescer REGISTRY and their grouping by mu, no engine behind them.  T threads
// make 10^4 calls on ONE key object with 10^4 DISTINCT mu (a caller that sweeps its test-vector value): the registry must hold
// one coalescer for the key throughout, every call must come back with ITS mu's result (the leader runs a batch as one group per
// distinct mu), and dropping the key (release()) while callers are inside must be safe -- they hold their coalescer by shared_ptr.
static int registry_mode(int threads, int total) {
    using namespace tfhe_amd_compat;
    static int key_a, key_b;  // two "key objects"
    std::vector<LweSample> out((size_t)total);
    std::atomic<int> wrong(0), groups(0), batches(0), max_reg(0);
    std::atomic<bool> stop(false);
    auto worker = [&](int t) {
        for (int id = t; id < total; id += threads) {
            const Torus32 mu = 1000 + id;  // distinct for every call
            const void *key = (id & 1) ? (const void *)&key_a : (const void *)&key_b;
            lwe_coalescer(1, key)->call(LweCall{&out[(size_t)id], nullptr, mu}, [&](const std::vector<LweCall> &batch) {
                batches++;
                for_each_mu_group(batch, [&](Torus32 m, const std::vector<LweCall> &calls) {
                    groups++;
                    for (const LweCall &c : calls) {
                        if (c.mu != m) wrong++;
                        c.result->b = 3 * m + 1;  // the "launch" of this group
                    }
                });
                std::this_thread::sleep_for(std::chrono::microseconds(30));
            });
            if (out[(size_t)id].b != 3 * mu + 1) wrong++;
            const int n = (int)lwe_coalescer_count();
            int m = max_reg.load();
            while (n > m && !max_reg.compare_exchange_weak(m, n)) {
            }
        }
    };
    std::thread dropper([&] {  // release() of one of the keys, again and again, while its callers are inside
        while (!stop.load()) {
            lwe_coalescers_drop(&key_a);
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto &th : pool) th.join();
    stop = true;
    dropper.join();
    unsigned long b = 0, q = 0;
    lwe_coalescer_totals(&b, &q);  // (includes the dropped coalescers' counts)
    lwe_coalescers_drop(nullptr);
    printf("{\"mode\": \"registry\", \"calls\": %d, \"distinct_mu\": %d, \"batches\": %d, \"groups\": %d, \"wrong\": %d, \"max_registry\": %d, "
           "\"requests_counted\": %lu, \"registry_after_release_all\": %d}\n",
           total, total, batches.load(), groups.load(), wrong.load(), max_reg.load(), q, (int)lwe_coalescer_count());
    return (wrong.load() || max_reg.load() > 2 || q != (unsigned long)total || lwe_coalescer_count() != 0) ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "registry") return registry_mode(argc > 2 ? atoi(argv[2]) : 8, argc > 3 ? atoi(argv[3]) : 10000);
    const int threads = argc > 1 ? atoi(argv[1]) : 32, calls = argc > 2 ? atoi(argv[2]) : 300;
    // rate mode (argv[3] > 0): a fixed "launch" of argv[3] microseconds and argv[4] microseconds of caller work between calls --
    // what share of the callers one launch carries (a tight loop of T threads should fill batches of T, not alternate in halves)
    const int launch_us = argc > 3 ? atoi(argv[3]) : 0, think_us = argc > 4 ? atoi(argv[4]) : 0;
    const auto wall0 = std::chrono::steady_clock::now();
    tfhe_amd_compat::Coalescer<Item> co;
    std::atomic<int> running(0), overlaps(0), batches(0), carried(0), max_batch(0);
    std::vector<int> results((size_t)threads * calls, -1);
    std::vector<int> led((size_t)threads, 0);
    auto worker = [&](int t) {
        for (int c = 0; c < calls; c++) {
            const int id = t * calls + c;
            int batches_led = 0;
            co.call(Item{id, &results[(size_t)id]}, [&](const std::vector<Item> &items) {
                if (running.fetch_add(1) != 0) overlaps++;
                batches++;
                batches_led++;
                int m = max_batch.load();
                while ((int)items.size() > m && !max_batch.compare_exchange_weak(m, (int)items.size())) {
                }
                std::this_thread::sleep_for(std::chrono::microseconds(launch_us > 0 ? launch_us : 20 + (id % 7) * 10));  // the "launch"
                for (const Item &it : items) {
                    *it.slot = it.id * 3 + 1;
                    carried++;
                }
                running.fetch_sub(1);
            });
            if (results[(size_t)id] != id * 3 + 1) {
                fprintf(stderr, "call %d returned before its request ran\n", id);
                return;
            }
            if (batches_led > 1) led[(size_t)t]++;
            if (think_us > 0) {  // the caller's own work between two calls (busy: a sleeping thread would not model it)
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(think_us);
                while (std::chrono::steady_clock::now() < until) {
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
    for (auto &th : pool) th.join();
    int bad = 0, multi = 0;
    for (size_t i = 0; i < results.size(); i++) bad += results[i] != (int)i * 3 + 1;
    for (int v : led) multi += v;
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    if (launch_us > 0)
        printf("{\"launch_us\": %d, \"think_us\": %d, \"calls_per_s\": %.0f, \"mean_batch\": %.1f, \"ideal_calls_per_s\": %.0f}\n", launch_us, think_us,
               threads * calls / wall, (double)(threads * calls) / batches.load(), threads / ((launch_us + think_us) * 1e-6));
    printf("{\"threads\": %d, \"calls\": %d, \"batches\": %d, \"carried\": %d, \"max_batch\": %d, \"overlaps\": %d, \"wrong\": %d, \"led_more_than_one\": %d}\n",
           threads, threads * calls, batches.load(), carried.load(), max_batch.load(), overlaps.load(), bad, multi);
    return (bad || overlaps.load() || multi || carried.load() != threads * calls) ? 1 : 0;
}
