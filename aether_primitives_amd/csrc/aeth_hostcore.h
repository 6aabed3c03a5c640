// aeth_hostcore.h -- the host-only machinery of the stream pipeline, free of HIP types and calls so that it also
// builds with plain g++ under -fsanitize=thread / address,undefined (tests/cpp/hostcore_sanitize.cpp, driven by
// tests/test_hostcore_sanitizers.py):
//   RangeMap   the registry of host ranges known to be page-locked (pool elements, explicit registrations)
//   PoolCore   the reference's object pool (src/pool.rs:43-221: make, take, take_or_make, give_back, len, cap) over
//              elements that come from a PinHooks allocator -- hipHostMalloc in the library, malloc in the sanitizer build
//   CopyTeam   the host threads that serve the pipeline's copy-in / copy-out stages (src/pipeline.rs:52-119 runs one
//              thread per stage; a PCIe link outruns one core's memcpy several times over)
// aeth_pool.hip and aeth_pipeline.hip wrap these with the HIP runtime and the C ABI's error texts.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace aeth {
namespace hostcore {

enum { PIN_POOL = 1, PIN_REGISTERED = 2, PIN_PENDING = 4 /* claimed, the page lock not yet confirmed */ };

// how pinned memory comes and goes (0 = success, anything else is handed back to the caller as is)
struct PinHooks {
    int (*alloc)(void *user, void **out, size_t bytes) = nullptr;
    int (*release)(void *user, void *p) = nullptr;
    void *user = nullptr;
};

// [lo, hi) ranges keyed by lo; thread-safe
class RangeMap {
public:
    void add(const void *p, size_t bytes, int kind)
    {
        std::lock_guard<std::mutex> l(mu_);
        m_[(uintptr_t)p] = R{(uintptr_t)p + bytes, kind};
    }
    void remove(const void *p)
    {
        std::lock_guard<std::mutex> l(mu_);
        m_.erase((uintptr_t)p);
    }
    // is [p, p + bytes) wholly inside ONE confirmed range?
    bool contains(const void *p, size_t bytes)
    {
        if (!p || !bytes) return false;
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        std::lock_guard<std::mutex> l(mu_);
        auto it = m_.upper_bound(lo);
        if (it == m_.begin()) return false;
        --it;
        return !(it->second.kind & PIN_PENDING) && it->first <= lo && hi <= it->second.hi;
    }
    // Claim [p, p + bytes) unless it touches a known range: the check and the insert happen under ONE lock, so two
    // threads cannot both claim overlapping ranges.  The entry is PENDING until confirm(); drop it with remove().
    bool try_claim(const void *p, size_t bytes, int kind)
    {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        std::lock_guard<std::mutex> l(mu_);
        auto it = m_.upper_bound(lo);                       // first range starting after lo
        if (it != m_.end() && it->first < hi) return false;
        if (it != m_.begin()) { --it; if (it->second.hi > lo) return false; }
        m_[lo] = R{hi, kind | PIN_PENDING};
        return true;
    }
    void confirm(const void *p)
    {
        std::lock_guard<std::mutex> l(mu_);
        auto it = m_.find((uintptr_t)p);
        if (it != m_.end()) it->second.kind &= ~PIN_PENDING;
    }
    // kind of the confirmed range that STARTS at p, 0 if none
    int kind_at(const void *p)
    {
        std::lock_guard<std::mutex> l(mu_);
        auto it = m_.find((uintptr_t)p);
        return (it == m_.end() || (it->second.kind & PIN_PENDING)) ? 0 : it->second.kind;
    }
    size_t size() { std::lock_guard<std::mutex> l(mu_); return m_.size(); }

private:
    struct R { uintptr_t hi; int kind; };
    std::mutex mu_;
    std::map<uintptr_t, R> m_;
};

inline RangeMap &ranges() { static RangeMap r; return r; }

// Pool<T> = Arc<Mutex<PoolInner<T>>> (pool.rs:71-73) with T = one pinned buffer of elem_bytes
class PoolCore {
public:
    enum { OK = 0, NOT_ELEMENT = 1, GIVEN_TWICE = 2, CHECKED_OUT = 3 };
    PoolCore(size_t elem_bytes, bool zero_on_return, PinHooks hooks) : elem_bytes_(elem_bytes), zero_(zero_on_return), hooks_(hooks) {}
    PoolCore(const PoolCore &) = delete;
    PoolCore &operator=(const PoolCore &) = delete;
    ~PoolCore() { (void)release_all(); }

    size_t elem_bytes() const { return elem_bytes_; }
    // pool::make's initial elements (:53-56: the resetter runs on them too); hook error code or 0
    int prefill(size_t n)
    {
        std::lock_guard<std::mutex> l(mu_);
        for (size_t i = 0; i < n; i++) {
            void *h = nullptr;
            const int rc = make_locked(&h);
            if (rc) return rc;
            if (zero_) memset(h, 0, elem_bytes_);
            elems_.push_back(h);
        }
        return 0;
    }
    void *take()                                          // Pool::take :78-97 (nullptr: the pool is empty)
    {
        std::lock_guard<std::mutex> l(mu_);
        if (elems_.empty()) return nullptr;
        void *h = elems_.back(); elems_.pop_back();
        return h;
    }
    int take_or_make(void **out)                          // Pool::take_or_make :115-132; hook error code or 0
    {
        *out = nullptr;
        std::lock_guard<std::mutex> l(mu_);
        if (elems_.empty()) return make_locked(out);
        *out = elems_.back(); elems_.pop_back();
        return 0;
    }
    int give_back(void *buf)                              // Elem::drop -> PoolInner::give_back :175-208
    {
        std::lock_guard<std::mutex> l(mu_);
        if (std::find(owned_.begin(), owned_.end(), buf) == owned_.end()) return NOT_ELEMENT;
        if (std::find(elems_.begin(), elems_.end(), buf) != elems_.end()) return GIVEN_TWICE;
        if (zero_) memset(buf, 0, elem_bytes_);
        elems_.push_back(buf);
        return OK;
    }
    size_t len() { std::lock_guard<std::mutex> l(mu_); return elems_.size(); }       // Pool::len :138-140
    size_t cap() { std::lock_guard<std::mutex> l(mu_); return owned_.size(); }       // Pool::cap :157-159
    size_t checked_out() { std::lock_guard<std::mutex> l(mu_); return owned_.size() - elems_.size(); }
    // frees every element, checked out or not; first hook error or 0
    int release_all()
    {
        std::lock_guard<std::mutex> l(mu_);
        int rc = 0;
        for (void *h : owned_) {
            ranges().remove(h);
            const int r = hooks_.release(hooks_.user, h);
            if (r && !rc) rc = r;
        }
        owned_.clear(); elems_.clear();
        return rc;
    }

private:
    int make_locked(void **out)                           // the `maker` (pool.rs:46,117): one pinned element
    {
        void *h = nullptr;
        const int rc = hooks_.alloc(hooks_.user, &h, elem_bytes_);
        if (rc) return rc;
        ranges().add(h, elem_bytes_, PIN_POOL);
        owned_.push_back(h);
        *out = h;
        return 0;
    }
    const size_t elem_bytes_;
    const bool zero_;
    const PinHooks hooks_;
    std::mutex mu_;
    std::vector<void *> elems_;          // checked-in elements (PoolInner.elems, :163)
    std::vector<void *> owned_;          // every element made so far; cap = owned.size() (PoolInner.cap, :171)
};

class CopyTeam {
public:
    explicit CopyTeam(int nthreads)
    {
        if (nthreads < 1) nthreads = 1;
        for (int i = 0; i < nthreads; i++) th_.emplace_back([this] { run(); });
    }
    ~CopyTeam()
    {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    CopyTeam(const CopyTeam &) = delete;
    CopyTeam &operator=(const CopyTeam &) = delete;
    // dst <- src in slices; *pending is raised by the number of slices now and lowered (release) as each completes
    void submit(void *dst, const void *src, size_t bytes, std::atomic<int> *pending)
    {
        if (bytes == 0) return;
        // slices of 1-4 MiB: enough of them for every thread, each long enough to amortise the hand-over
        size_t slice = bytes / (size_t)(2 * th_.size());
        const size_t lo = (size_t)1 << 20, hi = (size_t)4 << 20;
        slice = slice < lo ? lo : (slice > hi ? hi : slice);
        slice = (slice + 4095) & ~(size_t)4095;
        const int n = (int)((bytes + slice - 1) / slice);
        pending->fetch_add(n, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> l(mu_);
            for (size_t off = 0; off < bytes; off += slice)
                q_.push_back(Job{(char *)dst + off, (const char *)src + off, bytes - off < slice ? bytes - off : slice, pending});
        }
        if (n > 1) cv_.notify_all(); else cv_.notify_one();
    }
    int threads() const { return (int)th_.size(); }

private:
    struct Job { void *dst; const void *src; size_t bytes; std::atomic<int> *pending; };
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;             // stop_ and nothing left
                j = q_.front();
                q_.pop_front();
            }
            memcpy(j.dst, j.src, j.bytes);
            j.pending->fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> q_;
    bool stop_ = false;
};

}  // namespace hostcore
}  // namespace aeth
