// host_util.hpp -- the host-side pieces of libhmcgibbs.so that touch neither HIP nor the C ABI: the scatter helper
// threads, the chunk schedule of a chain and the static partition of windows over devices.  Plain C++17, so that the
// sanitizer builds of tests/sanitize/ (g++ -fsanitize=address,undefined / -fsanitize=thread, CPU only) compile exactly
// the code the library runs.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

namespace hmcg_hostutil {

// A few persistent host threads that share the scatter of a chunk (pinned staging -> the caller's arrays: 41 MB per call
// at the headline shape, as many small memcpys) with the calling thread.  On hosts whose single-thread copy rate is below
// the device's draw rate the scatter, not the GPU, would otherwise set the pace of the host entry.
class ScatterPool {
public:
    ~ScatterPool() { if (!th_.empty()) stop(); }
    void start(int workers)
    {
        if (!th_.empty() || workers <= 0) return;
        // a worker born after a stop() (hmcg_shutdown, then a new context) must not mistake the generations that went by
        // before its birth for a job: it starts from the generation current NOW (taken here, not in the thread, which may
        // first run after the first run() has already posted its job) and counts itself done only for a job it ran
        unsigned long g0;
        { std::lock_guard<std::mutex> lk(m_); g0 = gen_; }
        for (int i = 0; i < workers; ++i) th_.emplace_back([this, i, g0] { loop(i, g0); });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m_); quit_ = true; ++gen_; }
        go_.notify_all();
        for (auto& t : th_) t.join();
        th_.clear();
        quit_ = false;
    }
    // f(part, nparts) for part = 0..nparts-1, nparts = workers + 1; returns when every part is done
    void run(const std::function<void(int, int)>& f)
    {
        const int np = (int)th_.size() + 1;
        if (np == 1) { f(0, 1); return; }
        { std::lock_guard<std::mutex> lk(m_); job_ = &f; pending_ = np - 1; ++gen_; }
        go_.notify_all();
        f(np - 1, np);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
private:
    void loop(int id, unsigned long seen)
    {
        for (;;) {
            const std::function<void(int, int)>* f;
            int np;
            {
                std::unique_lock<std::mutex> lk(m_);
                go_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
                f = job_;
                np = (int)th_.size() + 1;
            }
            if (!f) continue;
            (*f)(id, np);
            { std::lock_guard<std::mutex> lk(m_); --pending_; }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable go_, done_;
    const std::function<void(int, int)>* job_ = nullptr;
    unsigned long gen_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

// kept draws produced by the sweeps [0, g) of the sampling schedule (n_samples blocks of burnin + nrun sweeps)
inline long long kept_before(long long g, int per, int burnin, int nrun)
{
    const long long smp = g / per, i = g - smp * per;
    return smp * nrun + std::max(0LL, std::min((long long)nrun, i - burnin));
}
// global sweep index just after kept draw number d - 1 (d >= 1) has been produced
inline long long sweep_after_kept(long long d, int per, int burnin, int nrun)
{
    const long long smp = (d - 1) / nrun, i = (d - 1) - smp * nrun;
    return smp * per + burnin + i + 1;
}

struct Chunk { int s0, s1; long long d0, d1; };   // sweeps [s0, s1) produce the kept draws [d0, d1)

// Chunks of the sweep range [sb, se): draw counts halve from chunk to chunk down to ~1/32 of the run (the last chunk's
// copy-out is the only one not hidden behind sampling), never more than `cap` draws in a chunk.  fdiv_env / keep_env:
// diagnostic overrides of the 32 and of the share a chunk takes ("num/den"), nullptr in production (hmcg.hip, diag_env).
inline std::vector<Chunk> plan_chunks(int sb, int se, int per, int burnin, int nrun, long long cap, bool stream_draws,
                                      const char* fdiv_env = nullptr, const char* keep_env = nullptr)
{
    std::vector<Chunk> out;
    const long long dB = kept_before(sb, per, burnin, nrun), dE = kept_before(se, per, burnin, nrun);
    const long long nd = dE - dB;
    if (!stream_draws || nd <= 0 || se <= sb) { out.push_back({sb, se, dB, dE}); return out; }
    long long fdiv = 32;       // (measured at the headline shape: 1/8 5.34 ms, 1/16 5.30, 1/32 5.23 per call)
    if (const char* e = fdiv_env) { const long long v = atoll(e); if (v >= 2 && v <= 1024) fdiv = v; }
    const long long floor_sz = std::max(16LL, nd / fdiv);
    // share of the remaining draws a chunk takes: 2/3 (four launches at the headline shape: 667, 222, 74, 37 draws).  Until
    // the chunk copies set out on time (round 4, KernelParams::skip_host) 1/2 measured better -- its smaller first chunk hid
    // the late first copy; now every relaunch saved is ~30 us (profiles/r04/trace_host_entry_skip_words.txt)
    long long keep_num = 2, keep_den = 3;
    if (const char* e = keep_env) {
        long long a = 0, b = 0;
        if (sscanf(e, "%lld/%lld", &a, &b) == 2 && a >= 1 && b > a && b <= 64) { keep_num = a; keep_den = b; }
    }
    long long d = dB;
    int s = sb;
    while (d < dE) {
        const long long rem = dE - d;
        long long take = std::min(cap, std::max((rem * keep_num + keep_den - 1) / keep_den, floor_sz));
        if (rem - take < floor_sz / 2) take = std::min(cap, rem);       // no crumbs
        take = std::min(take, rem);
        const long long d1 = d + take;
        const int s1 = d1 == dE ? se : (int)sweep_after_kept(d1, per, burnin, nrun);
        out.push_back({s, s1, d, d1});
        d = d1; s = s1;
    }
    if (out.back().s1 != se) out.back().s1 = se;
    return out;
}

// Static LPT partition (as hmc.jl_amd/shard.py partition_windows): windows by length, longest first (stable), each to
// the lightest device that still has room under the count cap ceil(W / G).
inline std::vector<std::vector<int32_t>> partition_windows(const int32_t* T, int W, int G)
{
    std::vector<int32_t> order((size_t)W);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return T[a] > T[b]; });
    std::vector<long long> load((size_t)G, 0);
    std::vector<int> count((size_t)G, 0);
    std::vector<std::vector<int32_t>> parts((size_t)G);
    const int cap = (W + G - 1) / G;
    for (int32_t w : order) {
        int best = -1;
        for (int r = 0; r < G; ++r)
            if (count[(size_t)r] < cap && (best < 0 || load[(size_t)r] < load[(size_t)best])) best = r;
        parts[(size_t)best].push_back(w);
        load[(size_t)best] += T[w];
        ++count[(size_t)best];
    }
    for (auto& p : parts) std::sort(p.begin(), p.end());
    return parts;
}

}  // namespace hmcg_hostutil
