// wave_emul.hpp -- TEST-ONLY emulation of one 64-lane wavefront on the CPU.
//
// The flow kernel (asyncflow_amd/csrc/af_flow.hpp) is written against a small wave interface
// (lane / ballot / any / shfl / sync / LDS atomic add).  On the GPU that is WaveHip (engine.hip); here
// it is 64 cooperative fibres (ucontext) scheduled round-robin by ONE thread: a cross-lane operation
// deposits the lane's value, yields until every lane has deposited, reads, and yields again before
// anyone may overwrite the exchange buffer.  Deterministic, no data races, and it checks the
// discipline the GPU needs as well: every lane must reach the SAME sequence of cross-lane operations
// (a lane that takes a different path trips the `site` assertion instead of hanging the wave).
// Built into tests/hostcheck/libaf_hostcheck.so only; never part of the product.
#pragma once

#include <execinfo.h>
#include <stdint.h>
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

namespace emu {

struct WaveState {
    static constexpr int kLanes = 64;
    ucontext_t main_ctx;
    ucontext_t ctx[kLanes];
    std::vector<char> stacks;
    bool done[kLanes];
    int n_done = 0;
    int current = -1;
    uint64_t xbuf[kLanes];
    uint32_t site[kLanes];
    std::function<void()> body;
};

inline WaveState*& cur() {
    static thread_local WaveState* w = nullptr;
    return w;
}

inline void yield_lane() {
    WaveState* w = cur();
    swapcontext(&w->ctx[w->current], &w->main_ctx);
}

inline void fibre_entry() {
    WaveState* w = cur();
    w->body();
    w->done[w->current] = true;
    w->n_done += 1;
    swapcontext(&w->ctx[w->current], &w->main_ctx);
}

// run `body` on 64 fibres until all of them return
inline void run_wave(const std::function<void()>& body) {
    WaveState w;
    constexpr size_t kStack = 512u * 1024u;
    w.stacks.resize(kStack * WaveState::kLanes);
    w.body = body;
    WaveState* prev = cur();
    cur() = &w;
    for (int i = 0; i < WaveState::kLanes; ++i) {
        w.done[i] = false;
        getcontext(&w.ctx[i]);
        w.ctx[i].uc_stack.ss_sp = w.stacks.data() + kStack * (size_t)i;
        w.ctx[i].uc_stack.ss_size = kStack;
        w.ctx[i].uc_link = &w.main_ctx;
        makecontext(&w.ctx[i], fibre_entry, 0);
    }
    while (w.n_done < WaveState::kLanes) {
        for (int i = 0; i < WaveState::kLanes; ++i) {
            if (w.done[i]) continue;
            w.current = i;
            swapcontext(&w.main_ctx, &w.ctx[i]);
        }
    }
    cur() = prev;
}

// deposit -> (everyone deposited) -> caller reads xbuf -> (everyone has read)
struct Exchange {
    WaveState* w;
    explicit Exchange(uint64_t v, uint32_t site) : w(cur()) {
        w->xbuf[w->current] = v;
        w->site[w->current] = site;
        yield_lane();
        if (w->n_done != 0) {
            std::fprintf(stderr, "wave_emul: a lane finished while others wait in a cross-lane operation (site %u)\n", site);
            std::abort();
        }
        for (int i = 0; i < WaveState::kLanes; ++i)
            if (w->site[i] != site) {
                std::fprintf(stderr, "wave_emul: divergent cross-lane operations: lane %d at site %u, lane %d at site %u\n",
                             w->current, site, i, w->site[i]);
                void* frames[24];   // (where THIS lane stands: `c++filt` the names)
                backtrace_symbols_fd(frames, backtrace(frames, 24), 2);
                std::abort();
            }
    }
    uint64_t of(uint32_t lane) const { return w->xbuf[lane & 63u]; }
    ~Exchange() { yield_lane(); }
};

// The wave backend handed to aff::Flow<W, IPL>.
struct WaveEmu {
    static uint32_t lane() { return (uint32_t)cur()->current; }
    static uint64_t ballot(bool p) {
        Exchange x(p ? 1u : 0u, 1u);
        uint64_t m = 0;
        for (uint32_t i = 0; i < 64u; ++i) m |= (x.of(i) & 1ull) << i;
        return m;
    }
    static bool any(bool p) { return ballot(p) != 0ull; }
    static uint32_t shfl32(uint32_t v, uint32_t src) {
        Exchange x(v, 2u);
        return (uint32_t)x.of(src);
    }
    static uint64_t shfl64(uint64_t v, uint32_t src) {
        Exchange x(v, 3u);
        return x.of(src);
    }
    static void sync() { Exchange x(0u, 4u); }
    static uint32_t lds_add(uint32_t* p, uint32_t v) {   // fibres never preempt each other
        const uint32_t old = *p;
        *p = old + v;
        return old;
    }
    static void global_add(uint32_t* p, uint32_t v) { *p += v; }
    static uint32_t global_load(const uint32_t* p) { return *p; }
    static void global_fence() {}
    static double rcp(double x) { return 1.0 / x; }
    static double fract(double x) { return x - __builtin_floor(x); }   // (x >= 0: exact, like v_fract_f64)
    static uint32_t bcast32(uint32_t v, uint32_t src) { return shfl32(v, src); }
    static uint64_t bcast64(uint64_t v, uint32_t src) { return shfl64(v, src); }
    static uint32_t scan_incl_u32(uint32_t v) {
        Exchange x(v, 5u);
        uint32_t s = 0;
        for (uint32_t i = 0; i <= lane(); ++i) s += (uint32_t)x.of(i);
        return s;
    }
    static uint32_t mbcnt(uint64_t m) { return (uint32_t)__builtin_popcountll(m & ((1ull << lane()) - 1ull)); }
    static unsigned long long clock() { return 0ull; }   // (FEAT_PROF is a device-only measurement build)
};

}  // namespace emu
