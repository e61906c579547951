// Host-side bookkeeping of the host-pointer query pipeline (anrag_hybrid_search, api.hip) -- no HIP in this file.
//
// The reference shares one SearchEngine between Streamlit session threads (src/app.py:17-27); here callers on several
// threads share one index.  A caller holds the index mutex only while it ENQUEUES its query; it waits for its own
// result outside the lock, so the queries of several threads pipeline on the device like back-to-back device calls.
// What makes that safe is a ring of N staging slots (pinned host block + device operands + result records each):
//   * the slot of a query is the pipeline's next sequence number mod N -- the same slot the device pipeline gives
//     the query's list sets, so host staging and device buffers are reused in lockstep;
//   * a slot is BUSY from the moment its caller starts staging operands until the caller has copied its results
//     out; a caller whose slot is busy waits on the condition variable and then starts over (another caller may have
//     advanced the sequence number, or re-sized the staging, meanwhile);
//   * re-sizing the staging (another corpus dimension) waits until no slot is busy.
// The device side is behind a Backend (below), so this file builds with plain g++ and runs under ThreadSanitizer
// with a fake device: tests/native/host_slots_tsan.cpp, driven by tests/test_host_slots_tsan.py.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>

namespace anrag {

template <int N>
struct HostSlotRing {
    bool busy[N] = {};
    std::condition_variable cv;
    int32_t sized_for = 0;  // what the slots' staging is sized for (0: nothing allocated)

    bool idle() const {
        for (bool b : busy)
            if (b) return false;
        return true;
    }

    // `lock` held.  Make the staging fit `want` (realloc(want) -> 0 or an error code): waits until nobody reads a slot.
    template <class Realloc>
    int ensure_size(std::unique_lock<std::mutex> &lock, int32_t want, Realloc &&realloc) {
        if (sized_for == want) return 0;
        cv.wait(lock, [&] { return idle(); });
        if (sized_for == want) return 0;  // another caller did it while this one waited
        sized_for = 0;
        const int rc = realloc(want);
        if (rc == 0) sized_for = want;
        return rc;
    }

    // `lock` held.  Take the slot the pipeline's next sequence number maps to if its previous user has copied out
    // (marks it busy, returns it); otherwise wait for a release and return -1: the caller starts over (re-checks the
    // staging size, re-reads the sequence number -- both may have changed while the lock was given up).
    int try_acquire(std::unique_lock<std::mutex> &lock, uint64_t next_seq) {
        const int s = (int)(next_seq % (uint64_t)N);
        if (!busy[s]) {
            busy[s] = true;
            return s;
        }
        cv.wait(lock);
        return -1;
    }

    // Frees slot s and wakes the waiters; leaves `lock` unlocked.  relock: the caller does not hold it.
    void release(std::unique_lock<std::mutex> &lock, int s, bool relock) {
        if (relock) lock.lock();
        busy[s] = false;
        lock.unlock();
        cv.notify_all();
    }
};

// One host-synchronous query through the ring.  Backend:
//   int      prepare(std::unique_lock<std::mutex>&)  under the lock: argument checks, ring.ensure_size(...)
//   uint64_t next_seq()                              under the lock: the pipeline's next sequence number
//   int      enqueue(int slot)                       under the lock: stage operands into the slot, enqueue the query
//                                                    (advances the sequence number), mark the slot's completion
//   void     drain()                                 under the lock, after a failed enqueue: nothing of the query may
//                                                    still touch the slot
//   int      wait(int slot)                          NOT under the lock: block until the slot's query has finished
//   void     fetch(int slot)                         NOT under the lock: copy the results out of the slot's staging
template <int N, class Backend>
int host_slot_query(std::mutex &mu, HostSlotRing<N> &ring, Backend &be) {
    std::unique_lock<std::mutex> lock(mu);
    int s, rc;
    do {
        // every time round: a wait below gives the lock up, and the staging may have been re-sized meanwhile (a caller
        // that slept through a re-size and then staged by the old size would write past its slot -- the fake-device
        // harness found exactly that)
        if ((rc = be.prepare(lock))) return rc;
        s = ring.try_acquire(lock, be.next_seq());
    } while (s < 0);
    rc = be.enqueue(s);
    if (rc) {
        be.drain();
        ring.release(lock, s, false);
        return rc;
    }
    lock.unlock();
    rc = be.wait(s);
    if (rc) {
        ring.release(lock, s, true);
        return rc;
    }
    be.fetch(s);
    ring.release(lock, s, true);
    return 0;
}

}  // namespace anrag
