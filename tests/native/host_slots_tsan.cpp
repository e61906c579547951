// GPU-free exercise of the host-slot pipeline bookkeeping (a-nice-rag_amd/csrc/host_slots.hpp) with a FAKE device,
// built with `g++ -fsanitize=thread` by tests/test_host_slots_tsan.py.  The real Backend is api.hip's HybridHostQuery
// (HIP streams and events); this one keeps the same contract:
//   * enqueue(slot) runs under the index mutex, stages the request into the slot, hands the slot to a device thread
//     (a FIFO = a stream) and advances the pipeline's sequence number; like hybrid_enqueue_group it first waits -- still
//     under the mutex -- until the job that used the slot's DEVICE buffers N sequence numbers ago has finished;
//   * wait(slot) blocks outside the mutex until the device thread has marked the slot done;
//   * fetch(slot) reads the slot's result staging outside the mutex.
// The staging arrays are plain (non-atomic) memory on purpose: if the ring ever let two callers own a slot at once, or
// re-sized the staging under a reader, ThreadSanitizer reports the race and the value checks fail.
// -DBREAK_RING builds a deliberately wrong pipeline (the slot is released BEFORE its results are copied out): the
// Python test expects that build to be caught.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "host_slots.hpp"

using namespace anrag;

constexpr int N = 4;  // a short ring: callers collide on slots all the time

struct FakeIndex {
    std::mutex mu;            // the index mutex
    HostSlotRing<N> ring;
    uint64_t seq = 0;         // the pipeline's sequence number (hyb_seq)
    // staging, (re)sized by ensure_size: request words and result words per slot
    std::vector<uint64_t> in, out;
    // "device": one worker executes jobs in FIFO order
    std::mutex dev_mu;
    std::condition_variable dev_cv;
    std::deque<std::pair<int, uint64_t>> fifo;  // (slot, sequence number)
    uint64_t finished[N] = {};                  // highest sequence number + 1 completed per slot (under dev_mu)
    bool stop = false;
    std::atomic<long> reallocs{0}, waits_on_device{0};
};

static void device_thread(FakeIndex *ix) {
    std::mt19937 rng(7);
    for (;;) {
        std::pair<int, uint64_t> job;
        {
            std::unique_lock<std::mutex> l(ix->dev_mu);
            ix->dev_cv.wait(l, [&] { return ix->stop || !ix->fifo.empty(); });
            if (ix->fifo.empty()) return;
            job = ix->fifo.front();
            ix->fifo.pop_front();
        }
        if (rng() % 4 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 60));
        const int s = job.first;
        const size_t w = ix->in.size() / N;           // words per slot (stable: no realloc while a slot is busy)
        for (size_t i = 0; i < w; ++i) ix->out[s * w + i] = ix->in[s * w + i] * 3 + i;
        {
            std::lock_guard<std::mutex> l(ix->dev_mu);
            ix->finished[s] = job.second + 1;
        }
        ix->dev_cv.notify_all();
    }
}

struct Query {  // host_slots.hpp's Backend
    FakeIndex *ix;
    int32_t words;       // "dimension" of this caller's requests
    uint64_t payload;
    bool fail_enqueue;
    uint64_t my_seq = 0;
    bool ok = false;

    int prepare(std::unique_lock<std::mutex> &lock) {
        return ix->ring.ensure_size(lock, words, [&](int32_t want) {
            ix->in.assign((size_t)want * N, 0);   // would race with any reader if a slot were still busy
            ix->out.assign((size_t)want * N, 0);
            ix->reallocs++;
            return 0;
        });
    }
    uint64_t next_seq() const { return ix->seq; }
    int enqueue(int s) {
        // device-side backpressure (hybrid_enqueue_group): the job N sequence numbers back used the same slot
        if (ix->seq >= (uint64_t)N) {
            std::unique_lock<std::mutex> l(ix->dev_mu);
            if (ix->finished[s] + N <= ix->seq) ix->waits_on_device++;
            ix->dev_cv.wait(l, [&] { return ix->finished[s] + N > ix->seq || ix->finished[s] >= ix->seq + 1 - N; });
        }
        if (fail_enqueue) return -2;  // nothing was handed to the device
        for (int32_t i = 0; i < words; ++i) ix->in[(size_t)s * words + i] = payload + i;
        my_seq = ix->seq++;
        {
            std::lock_guard<std::mutex> l(ix->dev_mu);
            ix->fifo.emplace_back(s, my_seq);
        }
        ix->dev_cv.notify_all();
        return 0;
    }
    void drain() {}
    int wait(int s) {
        std::unique_lock<std::mutex> l(ix->dev_mu);
        ix->dev_cv.wait(l, [&] { return ix->finished[s] >= my_seq + 1; });
        return 0;
    }
    void fetch(int s) {
        ok = true;
        for (int32_t i = 0; i < words; ++i)
            if (ix->out[(size_t)s * words + i] != (payload + i) * 3 + i) ok = false;
    }
};

#ifdef BREAK_RING
// the wrong pipeline: the slot is given back before its results are read
template <int NN, class Backend>
int broken_query(std::mutex &mu, HostSlotRing<NN> &ring, Backend &be) {
    std::unique_lock<std::mutex> lock(mu);
    int rc;
    int s;
    do {
        if ((rc = be.prepare(lock))) return rc;
        s = ring.try_acquire(lock, be.next_seq());
    } while (s < 0);
    rc = be.enqueue(s);
    ring.release(lock, s, false);  // <- too early
    if (rc) return rc;
    rc = be.wait(s);
    if (rc) return rc;
    std::this_thread::yield();
    be.fetch(s);
    return 0;
}
#endif

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 12;
    const int per_thread = argc > 2 ? atoi(argv[2]) : 400;
    FakeIndex ix;
    std::thread dev(device_thread, &ix);
    std::atomic<long> done{0}, wrong{0}, failed{0};
    std::vector<std::thread> callers;
    for (int t = 0; t < threads; ++t)
        callers.emplace_back([&, t] {
            std::mt19937 rng(100 + t);
            for (int i = 0; i < per_thread; ++i) {
                Query q{&ix, (rng() % 64 == 0) ? 24 : 16, ((uint64_t)t << 32) | (uint64_t)i, rng() % 97 == 0};
#ifdef BREAK_RING
                const int rc = broken_query(ix.mu, ix.ring, q);
#else
                const int rc = host_slot_query(ix.mu, ix.ring, q);
#endif
                if (rc) failed++;
                else if (!q.ok) wrong++;
                else done++;
                if (rng() % 8 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 40));
            }
        });
    for (auto &c : callers) c.join();
    {
        std::lock_guard<std::mutex> l(ix.dev_mu);
        ix.stop = true;
    }
    ix.dev_cv.notify_all();
    dev.join();
    bool idle;
    {
        std::lock_guard<std::mutex> l(ix.mu);
        idle = ix.ring.idle();
    }
    printf("queries %ld ok, %ld wrong, %ld failed enqueues; %ld staging re-sizes, %ld device waits; ring idle at the end: %d\n",
           done.load(), wrong.load(), failed.load(), ix.reallocs.load(), ix.waits_on_device.load(), (int)idle);
    return (wrong.load() == 0 && idle && done.load() + failed.load() == (long)threads * per_thread) ? 0 : 1;
}
