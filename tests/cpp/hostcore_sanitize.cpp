// hostcore_sanitize.cpp -- the stream pipeline's host-side machinery (csrc/aeth_hostcore.h: the pinned-range registry,
// the object pool of src/pool.rs:43-221, the copy threads) under the thread and address sanitizers, on the CPU build
// SURVEY section 5 names.  malloc stands in for hipHostMalloc through the pool's one allocator hook; everything else
// is the code libaether_hip.so runs.  Built twice by tests/test_hostcore_sanitizers.py:
//     g++ -std=c++17 -O1 -g -fsanitize=thread            -I aether_primitives_amd/csrc  ... -lpthread
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I aether_primitives_amd/csrc  ... -lpthread
// Exit code 0 and "hostcore: ok" = every check passed and the sanitizer stayed silent.
#include "aeth_hostcore.h"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace aeth::hostcore;

static std::atomic<long> g_live{0};
static int hook_alloc(void *, void **out, size_t bytes)
{
    void *p = nullptr;
    if (posix_memalign(&p, 4096, bytes) != 0) return -5;
    g_live.fetch_add(1);
    *out = p;
    return 0;
}
static int hook_release(void *, void *p) { free(p); g_live.fetch_sub(1); return 0; }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "hostcore: CHECK failed at line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

int main()
{
    const int T = 8;
    PinHooks hooks; hooks.alloc = hook_alloc; hooks.release = hook_release;

    // ---- the reference's three pool tests (src/pool.rs:228-296) on one thread: take / give back, len / cap, growth
    {
        PoolCore p(4096, true, hooks);
        CHECK(p.prefill(2) == 0 && p.len() == 2 && p.cap() == 2);
        void *a = p.take(), *b = p.take();
        CHECK(a && b && a != b && p.take() == nullptr && p.len() == 0);
        void *c = nullptr;
        CHECK(p.take_or_make(&c) == 0 && c && p.cap() == 3);
        memset(a, 0x5a, 4096);
        CHECK(p.give_back(a) == PoolCore::OK && p.give_back(a) == PoolCore::GIVEN_TWICE);
        int other = 0;
        CHECK(p.give_back(&other) == PoolCore::NOT_ELEMENT);
        CHECK(((unsigned char *)a)[17] == 0);                       // the resetter ran
        CHECK(p.checked_out() == 2 && ranges().contains(b, 4096) && ranges().contains((char *)b + 8, 100));
        CHECK(!ranges().contains((char *)b + 8, 4096));             // runs past the element
        CHECK(p.give_back(b) == PoolCore::OK && p.give_back(c) == PoolCore::OK && p.len() == 3);
    }
    CHECK(g_live.load() == 0 && ranges().size() == 0);              // the destructor freed and unregistered everything

    // ---- 8 threads on one pool and one copy team: take_or_make / fill / submit a copy / wait / check / give back
    {
        PoolCore p(1 << 20, false, hooks);
        CHECK(p.prefill(3) == 0);
        CopyTeam team(4);
        std::atomic<int> errors{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                std::vector<unsigned char> dst(3 << 20);
                for (int it = 0; it < 40; it++) {
                    void *e = nullptr;
                    if (p.take_or_make(&e) != 0 || !e) { errors++; return; }
                    if (!ranges().contains(e, 1 << 20)) errors++;
                    memset(e, (t * 41 + it) & 0xff, 1 << 20);
                    std::atomic<int> pending{0};
                    const size_t bytes = (size_t)(1 << 20) - (size_t)(it * 1000);
                    team.submit(dst.data() + it * 100, e, bytes, &pending);       // one slice, or several for the big ones
                    std::atomic<int> pending2{0};
                    team.submit(dst.data() + (2 << 20), e, 0, &pending2);         // an empty copy completes at once
                    while (pending.load(std::memory_order_acquire) != 0) std::this_thread::yield();
                    if (dst[it * 100] != ((t * 41 + it) & 0xff) || dst[it * 100 + bytes - 1] != ((t * 41 + it) & 0xff)) errors++;
                    if (p.give_back(e) != PoolCore::OK) errors++;
                    (void)p.len(); (void)p.cap();
                }
            });
        for (auto &x : th) x.join();
        CHECK(errors.load() == 0);
        CHECK(p.checked_out() == 0 && p.cap() >= 3 && p.cap() <= (size_t)T + 3 && p.len() == p.cap());
        // a large copy is cut into slices served by several threads
        std::vector<unsigned char> big(40 << 20, 7), out(40 << 20, 0);
        std::atomic<int> pending{0};
        team.submit(out.data(), big.data(), big.size(), &pending);
        while (pending.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        CHECK(memcmp(out.data(), big.data(), big.size()) == 0);
    }
    CHECK(g_live.load() == 0 && ranges().size() == 0);

    // ---- the range registry from 8 threads: overlapping claims -- exactly one of every overlapping set may win
    {
        std::vector<unsigned char> arena(64 * 4096);
        std::atomic<int> won{0}, lost{0}, errors{0};
        for (int round = 0; round < 50; round++) {
            won = 0; lost = 0;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    // thread t claims pages [t, t + 3): neighbours overlap, threads three apart do not
                    unsigned char *p = arena.data() + (size_t)t * 4096;
                    if (ranges().try_claim(p, 3 * 4096, PIN_REGISTERED)) {
                        if (ranges().contains(p, 4096)) errors++;                 // pending: not yet visible
                        ranges().confirm(p);
                        if (!ranges().contains(p + 100, 4096) || ranges().kind_at(p) != PIN_REGISTERED) errors++;
                        won++;
                    } else lost++;
                    (void)ranges().contains(arena.data(), 10);                    // queries race with the claims
                });
            for (auto &x : th) x.join();
            CHECK(won.load() + lost.load() == T && won.load() >= 2 && won.load() <= 3);   // at most pages 0-2, 3-5, 6-8 ...
            // no two confirmed ranges overlap
            int live = 0;
            for (int t = 0; t < T; t++) {
                unsigned char *p = arena.data() + (size_t)t * 4096;
                if (ranges().kind_at(p) == PIN_REGISTERED) {
                    live++;
                    for (int u = t + 1; u < t + 3 && u < T; u++) CHECK(ranges().kind_at(arena.data() + (size_t)u * 4096) == 0);
                    ranges().remove(p);
                }
            }
            CHECK(live == won.load() && ranges().size() == 0);
        }
        CHECK(errors.load() == 0);
    }
    printf("hostcore: ok\n");
    return 0;
}
