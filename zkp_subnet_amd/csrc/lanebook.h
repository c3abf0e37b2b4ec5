// lanebook.h -- the HOST-SIDE state machine of a context, free of HIP: who holds which lane, which MSM tickets are out,
// which pinned staging buffers are taken, which row-cache slots are in use.  One mutex + one condition variable.
//
// The reference's axon runs Miner.forward on worker threads (reference neurons/miner.py:106-135) and must never crash or
// hang (:133-135), so this bookkeeping is the part of the library that concurrent host threads actually contend on.  It
// lives apart from csrc/lanes.hip -- which only adds streams, buffers and kernels to the slots handed out here -- so that
// ThreadSanitizer can drive exactly this code from many threads with a fake back end on a box without a GPU
// (tests/lanebook_tsan.cpp, scripts/sanitize_cpu.sh tsan-lanes); the GPU stress test stays the functional check.
//
// Rules (the same the C-ABI documents, include/kzg_mi355x.h):
//   * a blocking call takes the lowest free lane for its duration and waits while lanes are merely busy with other calls;
//     it fails with BUSY when every lane is parked under an MSM ticket (only the ticket's owner can free those);
//   * a ticket (kzg_msm_submit / kzg_msm_sharded_begin) parks a lane until exactly one waiter collects or cancels it;
//   * whole-context operations (SRS load, slot upload, calibration, communicator set-up) take every lane and fail with
//     BUSY while any ticket is out;
//   * staging buffers and row-cache slots are handed to one holder at a time.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <mutex>

namespace kzg_book {

enum { LANE_FREE = 0, LANE_CALL, LANE_TICKET, LANE_WAITING };
enum { BOOK_OK = 0, BOOK_BUSY_TICKETS = 1, BOOK_BUSY_NO_TICKET_LANE = 2, BOOK_BAD_TICKET = 3, BOOK_NOT_HELD = 4 };

template <int N_LANES, int N_STAGE>
class LaneBook {
public:
    // ---- lanes
    // `serial`: profiling pins every call to lane 0 so that stage times stay attributable (kzg_set_profiling(1))
    int acquire(int state, int* out_li) {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            const int limit = serial_ ? 1 : N_LANES;
            bool any_call = false;
            for (int i = 0; i < limit; i++) {
                if (lane_[i] == LANE_FREE) {
                    lane_[i] = state;
                    *out_li = i;
                    return BOOK_OK;
                }
                any_call |= lane_[i] == LANE_CALL;
            }
            // nobody who could free a lane is running (every lane sits under a ticket), or the caller itself wants a
            // ticket and must not block: BUSY instead of a wait that only the caller's own kzg_msm_wait could end
            if (!any_call || state == LANE_TICKET) return state == LANE_TICKET ? BOOK_BUSY_NO_TICKET_LANE : BOOK_BUSY_TICKETS;
            cv_.wait(lk);
        }
    }
    int try_second(int first) {     // a second free lane for the two-lane form of a long commit+open, or -1
        std::lock_guard<std::mutex> lk(mu_);
        if (serial_) return -1;
        for (int i = 0; i < N_LANES; i++)
            if (i != first && lane_[i] == LANE_FREE) {
                lane_[i] = LANE_CALL;
                return i;
            }
        return -1;
    }
    void release(int li) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            lane_[li] = LANE_FREE;
        }
        cv_.notify_all();
    }
    int acquire_all() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            bool all_free = true, ticket = false;
            for (int i = 0; i < N_LANES; i++) {
                all_free &= lane_[i] == LANE_FREE;
                ticket |= lane_[i] == LANE_TICKET || lane_[i] == LANE_WAITING;
            }
            if (ticket) return BOOK_BUSY_TICKETS;
            if (all_free) break;
            cv_.wait(lk);
        }
        for (int i = 0; i < N_LANES; i++) lane_[i] = LANE_CALL;
        return BOOK_OK;
    }
    void release_all() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < N_LANES; i++) lane_[i] = LANE_FREE;
        }
        cv_.notify_all();
    }
    // exactly one waiter (or canceller) per ticket: a second one would read the lane's result after it has been reused
    int ticket_claim(int li) {
        std::lock_guard<std::mutex> lk(mu_);
        if (li < 0 || li >= N_LANES || lane_[li] != LANE_TICKET) return BOOK_BAD_TICKET;
        lane_[li] = LANE_WAITING;
        return BOOK_OK;
    }
    void set_serial(bool on) {
        std::lock_guard<std::mutex> lk(mu_);
        serial_ = on;
    }
    int lane_state(int li) {
        std::lock_guard<std::mutex> lk(mu_);
        return lane_[li];
    }

    // ---- pinned staging buffers: the bookkeeping only (the memory itself is the holder's business: `cap` is what the
    // holder last recorded with stage_set_cap while it held the slot)
    int stage_acquire(size_t bytes) {
        std::unique_lock<std::mutex> lk(mu_);
        int k = -1;
        for (;;) {
            for (int i = 0; i < N_STAGE && k < 0; i++)      // prefer a free buffer that is already large enough
                if (!stage_used_[i] && stage_cap_[i] >= bytes) k = i;
            for (int i = 0; i < N_STAGE && k < 0; i++)
                if (!stage_used_[i]) k = i;
            if (k >= 0) break;
            cv_.wait(lk);
        }
        stage_used_[k] = true;
        return k;
    }
    void stage_set_cap(int k, size_t cap) {
        std::lock_guard<std::mutex> lk(mu_);
        stage_cap_[k] = cap;
    }
    bool stage_held(int k) {
        std::lock_guard<std::mutex> lk(mu_);
        return k >= 0 && k < N_STAGE && stage_used_[k];
    }
    int stage_release(int k) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (k < 0 || k >= N_STAGE || !stage_used_[k]) return BOOK_NOT_HELD;
            stage_used_[k] = false;
        }
        cv_.notify_all();
        return BOOK_OK;
    }

    // ---- row cache: the coefficient vectors of the last N_LANES rows, keyed by the caller's 128-bit content tag.
    // lookup: >= 0 a hit on that slot (now busy); <= -2 a miss with slot (-2 - result) reserved for the caller to fill;
    // -1 a miss with every slot in use by concurrent requests (no caching for this call).
    int rcache_lookup(const uint8_t tag[16], uint64_t T, int eval_form) {
        std::lock_guard<std::mutex> lk(mu_);
        for (int k = 0; k < N_LANES; k++) {
            Row& e = row_[k];
            if (e.valid && !e.busy && e.T == T && e.eval_form == eval_form && !memcmp(e.tag, tag, 16)) {
                e.busy = true;
                e.stamp = ++rc_clock_;
                rc_hits_++;
                return k;
            }
        }
        rc_misses_++;
        int lru = -1;             // a slot to fill: an empty one, else the least recently used of those nobody is using
        for (int k = 0; k < N_LANES; k++) {
            const Row& e = row_[k];
            if (e.busy) continue;
            const uint64_t age_k = e.valid ? e.stamp : 0;
            if (lru < 0 || age_k < (row_[lru].valid ? row_[lru].stamp : 0)) lru = k;
        }
        if (lru >= 0) {
            row_[lru].busy = true;
            row_[lru].valid = false;
            return -2 - lru;
        }
        return -1;
    }
    void rcache_release(int k, bool valid, const uint8_t tag[16], uint64_t T, int eval_form) {
        std::lock_guard<std::mutex> lk(mu_);
        Row& e = row_[k];
        e.busy = false;
        e.valid = valid;
        if (valid) {
            memcpy(e.tag, tag, 16);
            e.T = T;
            e.eval_form = eval_form;
            e.stamp = ++rc_clock_;
        }
    }
    void rcache_collision() {     // a verified hit turned out to be another row under the same tag: recount it as a miss
        std::lock_guard<std::mutex> lk(mu_);
        rc_hits_--;
        rc_misses_++;
        rc_collisions_++;
    }
    void rcache_stats(uint64_t out[3]) {
        std::lock_guard<std::mutex> lk(mu_);
        out[0] = rc_hits_;
        out[1] = rc_misses_;
        out[2] = rc_collisions_;
    }

private:
    struct Row {
        uint8_t tag[16] = {0};
        uint64_t T = 0, stamp = 0;
        int eval_form = 0;
        bool valid = false, busy = false;
    };
    std::mutex mu_;
    std::condition_variable cv_;      // a lane or a staging buffer was released
    int lane_[N_LANES] = {};
    bool serial_ = false;
    bool stage_used_[N_STAGE] = {};
    size_t stage_cap_[N_STAGE] = {};
    Row row_[N_LANES];
    uint64_t rc_clock_ = 0, rc_hits_ = 0, rc_misses_ = 0, rc_collisions_ = 0;
};

}  // namespace kzg_book
