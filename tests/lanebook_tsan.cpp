// lanebook_tsan.cpp -- ThreadSanitizer drive of the context's host-side state machine (zkp_subnet_amd/csrc/lanebook.h: lanes,
// MSM tickets, the pinned staging pool, the row-cache slots) with a FAKE back end: what csrc/lanes.hip hangs on a slot
// (streams, buffers, kernels) is here a few plain, NON-atomic words per slot that the holder writes while its "kernels"
// (sleeps) run -- so if the book ever hands one slot to two threads, ThreadSanitizer reports a data race on those words,
// and the logical invariants are asserted on top.  Failures are injected: calls that bail out half-way, tickets that are
// cancelled, cache fills that fail, double claims, releases of buffers that are not held.
//
// The reference's axon runs Miner.forward on worker threads and must never crash (reference neurons/miner.py:106-135,
// :133-135): this is the bookkeeping those threads contend on.  The GPU box runs the same machine un-instrumented under
// tests/test_gpu_serving.py's lane stress test; no sanitizer is available there.
//
//   clang++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I zkp_subnet_amd/csrc tests/lanebook_tsan.cpp -o lanebook_tsan
//   ./lanebook_tsan [seconds=5] [threads=12]            (scripts/sanitize_cpu.sh tsan-lanes)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "lanebook.h"

namespace {
constexpr int NL = 4, NS = 4;
using Book = kzg_book::LaneBook<NL, NS>;
using namespace kzg_book;

struct Fake {                       // the "GPU side" of every slot: plain words, holder-only by contract
    uint64_t lane_word[NL] = {};
    int lane_owner[NL] = {};
    std::vector<uint8_t> stage_buf[NS];
    int stage_owner[NS] = {};
    uint64_t row_content[NL] = {};  // what the filler left in the slot ("coefficients"): must match the tag on every hit
};

Book book;
Fake fake;
std::atomic<bool> stop{false};
std::atomic<long> n_call{0}, n_two{0}, n_busy{0}, n_ticket{0}, n_claim_ok{0}, n_claim_dup{0}, n_cancel{0}, n_excl{0},
    n_excl_busy{0}, n_stage{0}, n_bad_release{0}, n_hit{0}, n_fill{0}, n_fill_fail{0}, n_nocache{0}, n_collision{0},
    n_serial{0}, failures{0};
std::atomic<int> serial_on{0};
std::mutex tq_mu;                   // tickets travel between threads (submit on one, wait on another): the test's own queue
std::deque<int> tickets;

void fail(const char* what) {
    failures++;
    fprintf(stderr, "INVARIANT BROKEN: %s\n", what);
}
void kernel(std::mt19937& g, int max_us) {   // a "kernel": the holder sleeps while it owns the slot
    std::this_thread::sleep_for(std::chrono::microseconds(g() % (max_us + 1)));
}
uint64_t content_of(const uint8_t tag[16]) {
    uint64_t v = 1469598103934665603ull;
    for (int i = 0; i < 16; i++) v = (v ^ tag[i]) * 1099511628211ull;
    return v;
}
void touch_lane(int li, int me, std::mt19937& g, int us) {
    fake.lane_owner[li] = me;
    fake.lane_word[li]++;
    kernel(g, us);
    if (fake.lane_owner[li] != me) fail("a lane changed hands while its holder was running");
}

void worker(int me, unsigned seed) {
    std::mt19937 g(seed);
    while (!stop.load(std::memory_order_relaxed)) {
        const unsigned op = g() % 100;
        if (op < 40) {                                        // a blocking call, sometimes on two lanes, sometimes failing
            int li = -1;
            const int rc = book.acquire(LANE_CALL, &li);
            if (rc != BOOK_OK) {
                if (rc != BOOK_BUSY_TICKETS) fail("a blocking call was refused for another reason than parked tickets");
                n_busy++;
                continue;
            }
            if (serial_on.load() == 2 && li != 0) fail("profiling is serial but a call got a lane other than 0");
            int li2 = -1;
            if (g() % 4 == 0) li2 = book.try_second(li);
            if (li2 == li) fail("try_second returned the first lane");
            touch_lane(li, me, g, 150);
            if (li2 >= 0) {
                touch_lane(li2, me, g, 50);
                n_two++;
            }
            // (a call that fails half-way drains its streams and then releases exactly like a clean one)
            if (li2 >= 0) book.release(li2);
            book.release(li);
            n_call++;
        } else if (op < 55) {                                 // submit a ticket (never blocks)
            int li = -1;
            const int rc = book.acquire(LANE_TICKET, &li);
            if (rc != BOOK_OK) {
                if (rc != BOOK_BUSY_NO_TICKET_LANE) fail("a ticket submit was refused with the wrong reason");
                n_busy++;
                continue;
            }
            fake.lane_owner[li] = me;                         // the submit queues its kernels ...
            fake.lane_word[li]++;
            {
                std::lock_guard<std::mutex> lk(tq_mu);        // ... and hands the ticket on (happens-before for the waiter)
                tickets.push_back(li);
            }
            n_ticket++;
        } else if (op < 72) {                                 // wait for / cancel somebody's ticket; a second claimant loses
            int li = -1;
            {
                std::lock_guard<std::mutex> lk(tq_mu);
                if (!tickets.empty()) {
                    li = tickets.front();
                    tickets.pop_front();
                }
            }
            if (li < 0) continue;
            if (book.ticket_claim(li) != BOOK_OK) {
                fail("the only claimant of a ticket was refused");
                continue;
            }
            if (book.ticket_claim(li) == BOOK_OK) fail("a ticket was claimed twice");
            else n_claim_dup++;
            touch_lane(li, me, g, 100);                       // the wait (or the drain of a cancel)
            if (g() % 5 == 0) n_cancel++;
            else n_claim_ok++;
            book.release(li);
        } else if (op < 76) {                                 // a whole-context operation
            const int rc = book.acquire_all();
            if (rc != BOOK_OK) {
                n_excl_busy++;                                // tickets are out: BUSY, never a wait
                continue;
            }
            for (int i = 0; i < NL; i++) {
                fake.lane_owner[i] = me;
                fake.lane_word[i]++;
            }
            kernel(g, 200);
            for (int i = 0; i < NL; i++)
                if (fake.lane_owner[i] != me) fail("a lane was used during a whole-context operation");
            book.release_all();
            n_excl++;
        } else if (op < 88) {                                 // a pinned staging buffer
            const size_t bytes = 32u << (g() % 8);
            const int k = book.stage_acquire(bytes);
            if (!book.stage_held(k)) fail("an acquired staging buffer is not held");
            fake.stage_owner[k] = me;
            if (fake.stage_buf[k].size() < bytes) {           // "hipHostMalloc": only the holder resizes
                fake.stage_buf[k].assign(bytes + bytes / 8, (uint8_t)me);
                book.stage_set_cap(k, fake.stage_buf[k].size());
            }
            fake.stage_buf[k][bytes - 1] = (uint8_t)me;
            kernel(g, 80);
            if (fake.stage_owner[k] != me || fake.stage_buf[k][bytes - 1] != (uint8_t)me) fail("a staging buffer changed hands");
            if (book.stage_release(k) != BOOK_OK) fail("the holder could not release its staging buffer");
            if (g() % 8 == 0) {                               // tokens that name no buffer are refused
                if (book.stage_release(-1) != BOOK_NOT_HELD || book.stage_release(NS) != BOOK_NOT_HELD || book.stage_held(NS))
                    fail("a staging token outside the pool was accepted");
                n_bad_release++;
            }
            n_stage++;
        } else if (op < 99) {                                 // the row cache
            uint8_t tag[16] = {0};
            tag[0] = (uint8_t)(g() % 7);                      // seven rows compete for four slots
            const uint64_t T = 1u << (10 + tag[0] % 2);
            const int look = book.rcache_lookup(tag, T, 1);
            if (look >= 0) {
                if (fake.row_content[look] != content_of(tag)) fail("a row-cache hit returned another row's coefficients");
                kernel(g, 60);
                if (g() % 16 == 0) {                          // "the verification found a different row under this tag"
                    book.rcache_collision();
                    book.rcache_release(look, false, tag, T, 1);
                    n_collision++;
                } else {
                    book.rcache_release(look, true, tag, T, 1);
                    n_hit++;
                }
            } else if (look <= -2) {
                const int slot = -2 - look;
                fake.row_content[slot] = 0;                   // the INTT writes the slot ...
                kernel(g, 60);
                const bool ok = g() % 10 != 0;                // ... unless the call fails half-way
                if (ok) fake.row_content[slot] = content_of(tag);
                book.rcache_release(slot, ok, tag, T, 1);
                (ok ? n_fill : n_fill_fail)++;
            } else {
                n_nocache++;
            }
        } else if (me == 0) {                                 // profiling on / off: calls then serialise on lane 0
            serial_on.store(1);                               // in transition: calls admitted before may still hold lanes
            book.set_serial(true);
            // wait until every lane but 0 has drained, then the invariant "only lane 0" is checkable
            for (int spin = 0; spin < 2000; spin++) {
                bool idle = true;
                for (int i = 1; i < NL; i++) idle &= book.lane_state(i) == LANE_FREE;
                if (idle) break;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            bool idle = true;
            for (int i = 1; i < NL; i++) idle &= book.lane_state(i) == LANE_FREE;
            if (idle) serial_on.store(2);                     // (tickets parked on lanes > 0 keep it at 1: nothing to check)
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            serial_on.store(0);
            book.set_serial(false);
            n_serial++;
        }
    }
}
}  // namespace

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "canary") {
        // the harness must SEE a bug of this kind: two threads write one lane's word WITHOUT the book -- expect a report
        std::thread a([] { for (int i = 0; i < 100000; i++) fake.lane_word[0]++; });
        std::thread b([] { for (int i = 0; i < 100000; i++) fake.lane_word[0]++; });
        a.join();
        b.join();
        printf("canary done (%llu)\n", (unsigned long long)fake.lane_word[0]);
        return 0;
    }
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int nthreads = argc > 2 ? atoi(argv[2]) : 12;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t, 1000u + 77u * (unsigned)t);
    std::this_thread::sleep_for(std::chrono::milliseconds((int)(seconds * 1000)));
    stop = true;
    for (auto& t : th) t.join();
    // collect what is still parked, as the last waiter would
    for (int li : tickets) {
        if (book.ticket_claim(li) != BOOK_OK) fail("a leftover ticket could not be claimed");
        book.release(li);
    }
    for (int i = 0; i < NL; i++)
        if (book.lane_state(i) != LANE_FREE) fail("a lane is still taken after every thread has finished");
    if (book.acquire_all() != BOOK_OK) fail("the idle context refused a whole-context operation");
    book.release_all();
    uint64_t rc[3];
    book.rcache_stats(rc);
    if ((long)rc[0] != n_hit.load() + 0 || (long)rc[2] != n_collision.load()) fail("row-cache hit / collision counters disagree with the drive");
    printf("lanebook drive: %d threads x %.1f s: %ld calls (%ld on two lanes), %ld tickets (%ld collected, %ld cancelled, %ld double "
           "claims refused), %ld busy answers, %ld whole-context operations (+%ld refused while tickets were out), %ld staging "
           "holds, %ld cache hits, %ld fills, %ld failed fills, %ld collisions, %ld uncached, %ld profiling toggles; "
           "invariant failures: %ld\n",
           nthreads, seconds, n_call.load(), n_two.load(), n_ticket.load(), n_claim_ok.load(), n_cancel.load(), n_claim_dup.load(),
           n_busy.load(), n_excl.load(), n_excl_busy.load(), n_stage.load(), n_hit.load(), n_fill.load(), n_fill_fail.load(),
           n_collision.load(), n_nocache.load(), n_serial.load(), failures.load());
    return failures.load() ? 1 : 0;
}
