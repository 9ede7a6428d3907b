// Batched analyzer on the device: per-scenario latency statistics, 1-s throughput buckets,
// an optional latency histogram and per-series mean/max of the sampled metrics.
//
// Replaces, for n scenarios at once, ResultsAnalyzer._process_event_metrics
// (/root/reference src/asyncflow/metrics/analyzer.py:83-126):
//   latencies = finish - start;  total, np.mean, np.median, np.std, np.percentile(95 / 99),
//   np.min, np.max;  RPS windows (k-1, k] for k = 1..floor(total_simulation_time).
// EVERY statistic is bit-equal to numpy's: the order statistics (partition + linear interpolation, including its "t >= 0.5"
// lerp branch) since round 2, mean and standard deviation since round 6 -- the kernel adds in numpy's own order:
//   np.add.reduce over a contiguous f64 array calls DOUBLE_add's reduce loop on pieces of at most 8 192 elements (the ufunc
//   buffer size) and adds the pieces' sums one after the other onto the identity; a piece is summed by DOUBLE_pairwise_sum
//   (numpy/_core/src/umath/loops_utils.h.src): n < 8: one after the other; n <= 128: eight running sums r[j] over the elements
//   j, j + 8, j + 16, ..., then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 elements left over one by one;
//   larger: the first n2 = (n / 2) - (n / 2) % 8 elements and the rest, recursively, added.
//   np.mean = that sum / n; np.std = sqrt(the same sum over (x - mean)^2, / n)   (_methods.py::_mean / _var).
// A full piece is therefore a PERFECT tree over 64 leaves of 128 elements (16 rows of 8): one piece per step of the
// workgroup, an 8-lane group per leaf, a lane per running sum -- every load instruction of a group is one full 128-byte line --
// and the tree is six butterfly exchanges plus eight values through LDS.  The last, partial piece has leaves of 64 .. 128
// elements found by walking the recursion down from every 64-th element; its sums meet in 128 LDS slots (a leaf of depth d
// with path p sits at p << (7 - d); absent slots are skipped, so the additions are exactly the recursion's).
//
// HBM-bound: one workgroup per scenario streams that scenario's rqs_clock rows (16 B per completed request) TWICE:
//   pass 1  sum / min / max / RPS buckets / histogram of the f64 exponent field -- and, for the (<= 3)
//           exponent bins in which the first 512 latencies put the wanted ranks, already the next 10 key
//           bits: when the guess covers every wanted rank (latencies of one scenario span 2-3 binades)
//           the first radix level costs no pass of its own
//   pass 2+ (only while a wanted rank still has > kCand candidates) 10 more key bits per pass,
//           MSB-first radix select, all wanted ranks at once
//   last    the squared deviations about the mean (which only exists after pass 1: numpy's two-pass variance cannot be had
//           in one), and on the way the <= kCand candidates of every wanted rank into LDS; then select by counting
// (Round 3 - 5 had a scratch array of 16-bit codes instead of the second read, and a one-pass shifted variance that agreed
// with numpy to 1e-13: 1.25 reads of the clock's bytes instead of 2.  Bit-equality costs the other 0.75.)
// No sort, no atomics on floating point (results are run-to-run deterministic).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace afs {

constexpr int kThreads = 512;   // (1 024 threads, two scenarios per CU, in the hope of Infinity Cache hits in the second pass: 6.4 -> 6.7 ms)
constexpr int kWaves = kThreads / 64;
constexpr int kRanks = 6;       // median lo/hi, p95 lo/hi, p99 lo/hi
constexpr int kCand = 512;      // candidates per rank resolved in LDS
constexpr uint32_t kPiece = 8192;   // numpy's reduction buffer: elements per call of the pairwise sum = per step of a workgroup
constexpr int kLeafRows = 16;       // a full piece's leaf: 128 elements = 16 rows of 8
constexpr int kTailSlots = 128;     // leaves of the partial piece (64 .. 128 elements each, < 8 192 together)
constexpr unsigned long long kAbsent = 0xFFF8DEADBEEF0000ull;   // an LDS slot no leaf was written to (a NaN no sum can be)
constexpr int kExpBins = 2048;  // level 0: bits 62..52 (latencies are >= +0.0, the sign bit is clear)
constexpr int kDigBits = 10;    // deeper levels: 10 key bits each
constexpr int kDigBins = 1 << kDigBits;

struct SumArgs {
    const double* clock;     // [n][clock_cap][2]  (start, finish)
    const uint32_t* counts;  // [n][8]
    uint32_t clock_cap;
    uint32_t cnt_completed_slot;
    double* stats;  // [n][8]  total, mean, median, std, p95, p99, min, max
    float* rps;     // [n][rps_buckets] or null
    uint32_t rps_buckets;
    uint32_t* hist;  // [n][hist_bins] or null
    uint32_t hist_bins;
    double hist_scale;  // hist_bins / hist_max
    // af_engine_run_summarized (engine.hip): the analyzer running BESIDE the simulation kernel's last residency round
    const uint32_t* done_flags;  // [n] or null: scenario's outputs are in memory (set by its wave, device-scope release)
    uint32_t* retry;             // [1 + n]: count, then the scenarios this kernel met unfinished (a later launch takes them: scen_map)
    const uint32_t* scen_map;    // null, or workgroup j analyses scenario scen_map[j]
};

// Which scenario is this workgroup's, and may it be analysed yet?  (message passing: the simulation wave stores its outputs,
// fences at device scope and sets the flag with release semantics; an acquire load of the flag followed by a device-scope
// acquire fence in every thread makes those outputs visible here, also across the XCDs' L2 caches)
__device__ inline bool claim_scenario(const uint32_t* done_flags, uint32_t* retry, const uint32_t* scen_map, uint32_t& sc) {
    sc = scen_map ? scen_map[blockIdx.x] : blockIdx.x;
    if (done_flags == nullptr) return true;
    __shared__ uint32_t go;
    if (threadIdx.x == 0) {
        const uint32_t f = __hip_atomic_load(done_flags + sc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (!f) retry[1u + atomicAdd(retry, 1u)] = sc;
        go = f;
    }
    __syncthreads();
    if (!go) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

__device__ inline double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ inline double wave_min(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// One LDS atomic per DISTINCT index of a wave instead of one per lane, for counters whose index is nearly the same in all
// 64 lanes (the exponent bin of a latency: 2-3 bins per scenario; the 1-s window of a completion: consecutive completions):
// 64 lanes on one LDS word serialise, and two such atomics per element made pass 1 LDS-bound rather than HBM-bound.
// Called by all lanes of the wave from uniform control flow; after 4 distinct values the rest go one by one.
__device__ inline void wave_agg_add(uint32_t* base, uint32_t idx, bool active) {
    unsigned long long todo = __ballot(active);
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < 4 && todo; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)idx, leader);   // (the leader is wave-uniform: v_readlane, not an LDS shuffle)
        const unsigned long long same = __ballot(active && idx == v) & todo;
        if (lane == leader) atomicAdd(&base[v], (uint32_t)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&base[idx], 1u);
}

// Deterministic block sum: per-thread partials -> wave tree -> the 8 wave results in lane order.
__device__ inline double block_sum(double v, double* scratch /* [kWaves] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < kWaves; ++w) r += scratch[w];
    return r;
}

// The 8 lanes of a group hold 8 consecutive completions -- nearly always one 1-s window: one LDS atomic per group instead
// of one per lane (64 lanes on one LDS word serialise; round 3 found pass 1 LDS-bound on exactly that).  Called by every
// lane of the wave from uniform control flow; a group's first lane is active whenever any of its lanes is.
__device__ inline void group_agg_add(uint32_t* base, uint32_t idx, bool active) {
    const int lane = threadIdx.x & 63;
    const uint32_t lead = (uint32_t)__shfl((int)idx, lane & ~7, 64);
    const bool same = active && idx == lead;
    const uint32_t grp = (uint32_t)(__ballot(same) >> (lane & ~7)) & 0xFFu;
    if ((lane & 7) == 0 && grp != 0u) atomicAdd(&base[lead], (uint32_t)__popc(grp));
    if (active && !same) atomicAdd(&base[idx], 1u);
}

__device__ inline double xor_add(double v, int m) { return v + __shfl_xor(v, m, 64); }   // (a + b == b + a bit for bit)

// numpy's np.add.reduce over the n values elem(row i) of a scenario, in numpy's own order (the head of this file).  `elem`
// is called once per completion by the lane that owns it -- and by every other lane of the wave with act = false (it may
// hold wave-wide operations) -- and returns the value to be added.  wsum: [2][kWaves] doubles, slots: [kTailSlots] doubles.
template <int kLoads, class Elem>
__device__ __forceinline__ double numpy_sum(const double2* ck, uint32_t n, double* wsum, double* slots, Elem&& elem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t j = (uint32_t)tid & 7u, g = (uint32_t)tid >> 3;
    double tot = 0.0;   // the reduction starts from the identity
    const uint32_t n_full = n / kPiece;
    for (uint32_t c = 0; c < n_full; ++c) {   // a full piece: leaf g = elements [128 g, 128 g + 128), lane j its running sum r[j]
        const double2* p = ck + (size_t)c * kPiece + g * 128u + j;
        double acc = -0.0;   // (-0.0 + x == x for every x: the first row needs no case of its own)
#pragma unroll 1
        for (int r0 = 0; r0 < kLeafRows; r0 += kLoads) {
            double2 cc[kLoads];
#pragma unroll
            for (int v = 0; v < kLoads; ++v) cc[v] = p[(r0 + v) * 8];
#pragma unroll
            for (int v = 0; v < kLoads; ++v) acc = acc + elem(cc[v], true);
        }
        acc = xor_add(acc, 1); acc = xor_add(acc, 2); acc = xor_add(acc, 4);      // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
        acc = xor_add(acc, 8); acc = xor_add(acc, 16); acc = xor_add(acc, 32);    // the wave's eight leaves, pairwise
        double* ws = wsum + (c & 1u) * kWaves;   // (two buffers: one barrier per piece)
        if (lane == 0) ws[wave] = acc;
        __syncthreads();
        tot = tot + (((ws[0] + ws[1]) + (ws[2] + ws[3])) + ((ws[4] + ws[5]) + (ws[6] + ws[7])));
    }
    const uint32_t m = n - n_full * kPiece;
    if (m == 0u) return tot;
    // the partial piece: the recursion's leaves hold 64 .. 128 elements (or all m <= 128), so every leaf holds a multiple of
    // 64; group g walks the recursion down to the leaf of element 64 g (and 64 (g + 64)) and takes it if it is the first there
    const double2* q = ck + (size_t)n_full * kPiece;
    if (tid < kTailSlots) slots[tid] = __longlong_as_double((long long)kAbsent);
    __syncthreads();
    const uint32_t n_cand = (m + 63u) / 64u;
    for (uint32_t k0 = 0; k0 < n_cand; k0 += (uint32_t)kThreads / 8u) {
        const uint32_t k = k0 + g, e = 64u * k;
        uint32_t o = 0u, len = m, path = 0u, depth = 0u;
        bool mine = k < n_cand;
        if (mine) {
            while (len > 128u) {
                const uint32_t n2 = (len >> 1) & ~7u;
                if (e - o < n2) { len = n2; path <<= 1; }
                else { o += n2; len -= n2; path = (path << 1) | 1u; }
                depth += 1u;
            }
            if (k > 0u && e - 64u >= o) mine = false;   // (the candidate before this one lies in the same leaf)
        }
        const uint32_t rows = mine ? len >> 3 : 0u, rem = mine ? len & 7u : 0u;
        double acc = -0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < kLeafRows; r0 += kLoads) {   // (<= 16 rows; the trip count is uniform, the rows are not)
            double2 cc[kLoads];
#pragma unroll
            for (int v = 0; v < kLoads; ++v) cc[v] = (uint32_t)(r0 + v) < rows ? q[o + (uint32_t)(r0 + v) * 8u + j] : double2{0.0, 0.0};
#pragma unroll
            for (int v = 0; v < kLoads; ++v) {
                const bool act = (uint32_t)(r0 + v) < rows;
                const double x = elem(cc[v], act);
                if (act) acc = acc + x;
            }
        }
        double leaf = xor_add(acc, 1);
        leaf = xor_add(leaf, 2);
        leaf = xor_add(leaf, 4);
        if (rows == 0u) leaf = -0.0;   // n < 8: one after the other, from -0.0
        {   // the len % 8 elements left over (the piece's last leaf only), one by one
            const bool act = j < rem;
            const double2 cc = act ? q[o + 8u * rows + j] : double2{0.0, 0.0};
            const double x = elem(cc, act);
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const double xt = __shfl(x, (lane & ~7) + t, 64);
                if ((uint32_t)t < rem) leaf = leaf + xt;
            }
        }
        if (mine && j == 0u) slots[path << (7u - depth)] = leaf;   // (depth <= 7: a node of depth 7 holds < 8 192 / 128 + 15 elements)
    }
    __syncthreads();
    for (uint32_t st = 1u; st < (uint32_t)kTailSlots; st <<= 1) {   // the recursion's additions, bottom up; an absent right half = a leaf higher up
        if ((uint32_t)tid < (uint32_t)kTailSlots && ((uint32_t)tid & (2u * st - 1u)) == 0u) {
            const double r = slots[(uint32_t)tid + st];
            if ((unsigned long long)__double_as_longlong(r) != kAbsent) slots[tid] = slots[tid] + r;
        }
        __syncthreads();
    }
    tot = tot + slots[0];
    __syncthreads();   // (the slots are the next call's)
    return tot;
}

// One wave finds the bin holding rank k of a histogram: bin, #elements below it, its count.
__device__ inline void wave_select(const uint32_t* hist, int nbins, uint32_t k, uint32_t& bin, uint32_t& below,
                                   uint32_t& count) {
    const int lane = threadIdx.x & 63;
    const int per = nbins / 64;
    uint32_t mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[lane * per + j];
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const uint32_t excl = incl - mine;
    const bool owner = excl <= k && k < incl;
    uint32_t b = 0, bl = 0, c = 0;
    if (owner) {
        uint32_t run = excl;
        for (int j = 0; j < per; ++j) {
            const uint32_t h = hist[lane * per + j];
            if (k < run + h) {
                b = (uint32_t)(lane * per + j);
                bl = run;
                c = h;
                break;
            }
            run += h;
        }
    }
    const unsigned long long m = __ballot(owner);
    const int src = m ? __ffsll((long long)m) - 1 : 0;
    bin = __shfl(b, src, 64);
    below = __shfl(bl, src, 64);
    count = __shfl(c, src, 64);
}

// kWpe: waves per SIMD the register allocation aims at -- 4: 115 registers, two scenarios (workgroups) per CU; 8: 64 registers,
// four, which is what the 35 KB of LDS admit (51 spilled registers, none of them inside the pass loops).  The kernel waits
// most of its time (61 % of its wave cycles in s_waitcnt at two scenarios per CU: profiles/r05/binding_c2.json), residency hides
// its HBM and LDS round trips: measured on BASELINE config 2 (profiles/r05/summary_wpe_ab.txt) 4.67 -> 4.28 ms for the
// analyzer's two kernels (6 waves, three scenarios per CU: 4.6).  af_engine_summarize launches <8>; AF_SUMMARY_WPE=4 the other.
template <int kWpe>
__global__ __launch_bounds__(kThreads, kWpe) void af_summary_kernel(SumArgs a) {
    extern __shared__ uint32_t dyn[];  // [rps_buckets] then [hist_bins]
    __shared__ uint32_t exp_hist[kExpBins];
    __shared__ __attribute__((aligned(16))) uint32_t dig_hist[kRanks][kDigBins];
    static_assert(sizeof(double) * kCand == sizeof(uint32_t) * kDigBins, "the candidates take the digit histograms' place");
    double (*cand)[kCand] = reinterpret_cast<double (*)[kCand]>(&dig_hist[0][0]);   // (dead by the last pass: 35 instead of 59 KB, 4 scenarios per CU)
    __shared__ uint32_t cand_n[kRanks];
    __shared__ double scratch[kWaves];
    __shared__ unsigned long long pfx[kRanks];       // key >> shift of the bin holding rank r
    __shared__ unsigned long long slot_pfx[kRanks];  // distinct prefixes
    __shared__ uint32_t rank_in[kRanks];             // rank r relative to its bin
    __shared__ uint32_t cnt[kRanks];                 // elements in that bin
    __shared__ uint32_t slot_of[kRanks];
    __shared__ uint32_t n_slots, more;
    __shared__ double val[kRanks];
    __shared__ uint32_t g_pfx[3];   // exponent bins guessed from the first 512 latencies
    __shared__ uint32_t g_n, g_hit;
    __shared__ double wsum[2 * kWaves];
    __shared__ double tail_slots[kTailSlots];

    constexpr int kLoads = kWpe > 4 ? 4 : 8;   // (16-byte loads in flight per thread: eight at 128 registers, four at 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t sc;
    if (!claim_scenario(a.done_flags, a.retry, a.scen_map, sc)) return;
    uint32_t n = a.counts[(size_t)sc * 8u + a.cnt_completed_slot];
    if (n > a.clock_cap) n = a.clock_cap;
    const double2* ck = reinterpret_cast<const double2*>(a.clock) + (size_t)sc * a.clock_cap;
    uint32_t* rps_l = dyn;
    uint32_t* hist_l = dyn + a.rps_buckets;

    for (int i = tid; i < kExpBins; i += kThreads) exp_hist[i] = 0u;
    for (uint32_t i = tid; i < a.rps_buckets + (a.hist ? a.hist_bins : 0u); i += kThreads) dyn[i] = 0u;
    for (uint32_t i = tid; i < 3u * (uint32_t)kDigBins; i += kThreads) (&dig_hist[0][0])[i] = 0u;
    __syncthreads();

    // ---- guess: in which exponent bins do the first 512 latencies put the median / p95 / p99? ------------
    {
        const uint32_t m = n < (uint32_t)kThreads ? n : (uint32_t)kThreads;
        if ((uint32_t)tid < m) {
            const double2 c = ck[tid];
            atomicAdd(&exp_hist[((unsigned long long)__double_as_longlong(c.y - c.x) >> 52) & (kExpBins - 1)], 1u);
        }
        __syncthreads();
        if (wave < 3 && m > 0u) {
            const uint32_t k = wave == 0 ? m / 2u : wave == 1 ? (uint32_t)((double)(m - 1u) * 0.95) : (uint32_t)((double)(m - 1u) * 0.99);
            uint32_t bin, below, count;
            wave_select(exp_hist, kExpBins, k, bin, below, count);
            if (lane == 0) g_pfx[wave] = bin;
        }
        __syncthreads();
        if (tid == 0) {   // distinct guesses first
            uint32_t k = 0;
            if (m > 0u)
                for (uint32_t r = 0; r < 3u; ++r) {
                    bool seen = false;
                    for (uint32_t q = 0; q < k; ++q) seen = seen || g_pfx[q] == g_pfx[r];
                    if (!seen) g_pfx[k++] = g_pfx[r];
                }
            g_n = k;
        }
        for (int i = tid; i < kExpBins; i += kThreads) exp_hist[i] = 0u;
        __syncthreads();
    }
    const uint32_t gn = g_n, gp0 = g_pfx[0], gp1 = g_pfx[1], gp2 = g_pfx[2];

    // ---- pass 1 -----------------------------------------------------------------------------
    double mn = __builtin_inf(), mx = -__builtin_inf();
    const double total = numpy_sum<kLoads>(ck, n, wsum, tail_slots, [&](const double2 c, const bool act) -> double {
        const double lat = c.y - c.x;
        const unsigned long long key = (unsigned long long)__double_as_longlong(lat);
        const uint32_t ebin = (uint32_t)(key >> 52) & (kExpBins - 1);
        if (act) {
            mn = fmin(mn, lat);
            mx = fmax(mx, lat);
            const uint32_t dig = (uint32_t)(key >> (52 - kDigBits)) & (kDigBins - 1);
            if (gn > 0u && ebin == gp0) atomicAdd(&dig_hist[0][dig], 1u);
            else if (gn > 1u && ebin == gp1) atomicAdd(&dig_hist[1][dig], 1u);
            else if (gn > 2u && ebin == gp2) atomicAdd(&dig_hist[2][dig], 1u);
            else atomicAdd(&exp_hist[ebin], 1u);   // (rare: the guessed bins hold nearly everything, and THEIR counts are the sums of their digit bins, below)
        }
        if (a.rps) {
            // window (k-1, k]; a finish at exactly 0 belongs to the first window (analyzer.py:112-121)
            const double kf = ceil(c.y);
            const uint32_t k = kf < 1.0 ? 1u : (kf > 4.0e9 ? 0xFFFFFFFFu : (uint32_t)kf);
            group_agg_add(rps_l, k - 1u, act && k <= a.rps_buckets);
        }
        if (a.hist && act) {
            const double bf = lat * a.hist_scale;
            const uint32_t b = bf >= (double)(a.hist_bins - 1u) ? a.hist_bins - 1u : (uint32_t)bf;
            atomicAdd(&hist_l[b], 1u);
        }
        return lat;
    });
    __syncthreads();   // (closes pass 1's histogram updates)
    if ((uint32_t)wave < gn) {   // exponent bin q of the guess: as many as its digit bins hold together
        uint32_t c = 0;
        for (int j = lane; j < kDigBins; j += 64) c += dig_hist[wave][j];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (lane == 0) exp_hist[g_pfx[wave]] = c;
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) scratch[wave] = mn;
    __syncthreads();
    double vmin = scratch[0];
    for (int w = 1; w < kWaves; ++w) vmin = fmin(vmin, scratch[w]);
    __syncthreads();
    if (lane == 0) scratch[wave] = mx;
    __syncthreads();
    double vmax = scratch[0];
    for (int w = 1; w < kWaves; ++w) vmax = fmax(vmax, scratch[w]);
    __syncthreads();

    if (a.rps)
        for (uint32_t i = tid; i < a.rps_buckets; i += kThreads) a.rps[(size_t)sc * a.rps_buckets + i] = (float)rps_l[i];
    if (a.hist)
        for (uint32_t i = tid; i < a.hist_bins; i += kThreads) a.hist[(size_t)sc * a.hist_bins + i] = hist_l[i];

    double* st = a.stats + (size_t)sc * 8u;
    if (n == 0u) {  // the reference leaves latency_stats empty (analyzer.py:105-106)
        if (tid < 8) st[tid] = tid == 0 ? 0.0 : __builtin_nan("");
        return;
    }
    const double mean = total / (double)n;

    // ---- wanted ranks (numpy: median = mean of the middle pair; percentile 'linear') ---------
    __shared__ uint32_t want[kRanks];
    __shared__ double tfrac[2];
    if (tid == 0) {
        want[0] = (n & 1u) ? n / 2u : n / 2u - 1u;
        want[1] = n / 2u;
        const double q[2] = {95.0 / 100.0, 99.0 / 100.0};
        for (int p = 0; p < 2; ++p) {
            const double v = (double)(n - 1u) * q[p];
            if (v >= (double)(n - 1u)) {
                want[2 + 2 * p] = want[3 + 2 * p] = n - 1u;
                tfrac[p] = 0.0;
            } else {
                const double f = floor(v);
                want[2 + 2 * p] = (uint32_t)f;
                want[3 + 2 * p] = (uint32_t)f + 1u;
                tfrac[p] = v - f;
            }
        }
    }
    __syncthreads();

    // ---- level 0: exponent bin of every wanted rank ---------------------------------------------
    if (wave < kRanks) {
        uint32_t bin, below, count;
        wave_select(exp_hist, kExpBins, want[wave], bin, below, count);
        if (lane == 0) {
            pfx[wave] = bin;
            rank_in[wave] = want[wave] - below;
            cnt[wave] = count;
        }
    }
    int shift = 52;
    // ---- the guessed bins already carry their next 10 key bits: if they cover every wanted rank that still has
    // too many candidates ... (a rank that is already narrow enough keeps its exponent bin: prefixes of different
    // lengths cannot be mixed, so the shortcut is taken only when ALL ranks can take it)
    __syncthreads();
    if (tid == 0) {
        uint32_t hit = 1u;
        for (int r = 0; r < kRanks; ++r) {
            bool ok = false;
            for (uint32_t q = 0; q < g_n; ++q) ok = ok || pfx[r] == (unsigned long long)g_pfx[q];
            if (!ok) hit = 0u;
        }
        g_hit = hit;
    }
    __syncthreads();
    if (g_hit) {
        if (wave < kRanks) {
            uint32_t q = 0;
            for (uint32_t j = 0; j < g_n; ++j)
                if (pfx[wave] == (unsigned long long)g_pfx[j]) q = j;
            uint32_t bin, below, count;
            wave_select(dig_hist[q], kDigBins, rank_in[wave], bin, below, count);
            __syncthreads();
            if (lane == 0) {
                pfx[wave] = (pfx[wave] << kDigBits) | bin;
                rank_in[wave] -= below;
                cnt[wave] = count;
            }
        } else {
            __syncthreads();
        }
        shift = 52 - kDigBits;
    }

    // ---- deeper levels while some rank still has too many candidates ------------------------------
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            uint32_t ns = 0, m = 0;
            for (int r = 0; r < kRanks; ++r) {
                uint32_t sidx = ns;
                for (uint32_t q = 0; q < ns; ++q)
                    if (slot_pfx[q] == pfx[r]) sidx = q;
                if (sidx == ns) slot_pfx[ns++] = pfx[r];
                slot_of[r] = sidx;
                if (cnt[r] > (uint32_t)kCand && shift > 0) m = 1u;
            }
            n_slots = ns;
            more = m;
        }
        __syncthreads();
        if (!more) break;
        const int bits = shift >= kDigBits ? kDigBits : shift;
        const int new_shift = shift - bits;
        const uint32_t ns = n_slots;
        for (uint32_t i = tid; i < ns * (uint32_t)kDigBins; i += kThreads) (&dig_hist[0][0])[i] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kThreads) {
            const double2 c = ck[i];
            const unsigned long long key = (unsigned long long)__double_as_longlong(c.y - c.x);
            const unsigned long long hi = key >> shift;
            for (uint32_t q = 0; q < ns; ++q)
                if (hi == slot_pfx[q]) atomicAdd(&dig_hist[q][(key >> new_shift) & ((1u << bits) - 1u)], 1u);
        }
        __syncthreads();
        if (wave < kRanks) {
            uint32_t bin, below, count;
            wave_select(dig_hist[slot_of[wave]], kDigBins, rank_in[wave], bin, below, count);
            if (lane == 0) {
                pfx[wave] = (pfx[wave] << bits) | bin;
                rank_in[wave] -= below;
                cnt[wave] = count;
            }
        }
        shift = new_shift;
    }

    // ---- last pass: the squared deviations in numpy's order (x - mean, squared, summed like the latencies), candidates on the way
    if (tid < kRanks) cand_n[tid] = 0u;
    __syncthreads();
    const uint32_t ns = n_slots;
    unsigned long long sp[kRanks];
#pragma unroll
    for (int q = 0; q < kRanks; ++q) sp[q] = (uint32_t)q < ns ? slot_pfx[q] : ~0ull;   // (a prefix no key >> shift can equal: latencies are >= +0.0)
    const double sq_total = numpy_sum<kLoads>(ck, n, wsum, tail_slots, [&](const double2 c, const bool act) -> double {
        const double lat = c.y - c.x;
        const double d = lat - mean;
        if (act && shift > 0) {
            const unsigned long long hi = (unsigned long long)__double_as_longlong(lat) >> shift;
#pragma unroll
            for (int q = 0; q < kRanks; ++q)
                if (hi == sp[q]) {
                    const uint32_t pos = atomicAdd(&cand_n[q], 1u);
                    if (pos < (uint32_t)kCand) cand[q][pos] = lat;
                }
        }
        return d * d;
    });
    __syncthreads();
    for (int r = 0; r < kRanks; ++r) {
        const uint32_t q = slot_of[r];
        if (shift == 0) {  // the whole key is known: every candidate has this value
            if (tid == 0) val[r] = __longlong_as_double((long long)pfx[r]);
            continue;
        }
        const uint32_t m = cand_n[q] < (uint32_t)kCand ? cand_n[q] : (uint32_t)kCand;
        const uint32_t k = rank_in[r];
        if ((uint32_t)tid < m) {
            const double x = cand[q][tid];
            uint32_t less = 0, leq = 0;
            for (uint32_t j = 0; j < m; ++j) {
                const double y = cand[q][j];
                less += y < x ? 1u : 0u;
                leq += y <= x ? 1u : 0u;
            }
            if (less <= k && k < leq) val[r] = x;
        }
    }
    __syncthreads();

    if (tid == 0) {
        auto lerp = [](double lo, double hi, double t) {  // numpy _lerp
            const double d = hi - lo;
            return t >= 0.5 ? hi - d * (1.0 - t) : lo + d * t;
        };
        st[0] = (double)n;
        st[1] = mean;
        st[2] = (n & 1u) ? val[1] : (val[0] + val[1]) / 2.0;
        st[3] = sqrt(sq_total / (double)n);
        st[4] = lerp(val[2], val[3], tfrac[0]);
        st[5] = lerp(val[4], val[5], tfrac[1]);
        st[6] = vmin;
        st[7] = vmax;
    }
}

// Per-series mean and maximum of the sampled metrics of every scenario
// (samples [n][tick_cap][pitch] 4-byte words; only the first counts[CNT_TICKS] rows are valid).
// Column j < n_edges and the ready / io columns of a server hold int32 counts; the ram_in_use
// column (j = n_edges + 3 s + 2) holds float32 values (include/asyncflow_hip.h): its mean is the
// f64 sum of the float values / ticks, its maximum the float maximum, returned as float32 bits
// (for non-negative floats the bit patterns order like the values).
struct SeriesArgs {
    const uint32_t* samples;
    const uint32_t* counts;
    uint32_t tick_cap, pitch, n_series, cnt_ticks_slot, n_edges;
    double* mean;    // [n][n_series]
    uint32_t* maxv;  // [n][n_series]
    const uint32_t* done_flags;  // (as in SumArgs)
    uint32_t* retry;
    const uint32_t* scen_map;
};

constexpr int kSeriesThreads = 256;

__device__ __forceinline__ bool series_is_float(uint32_t j, uint32_t n_edges, uint32_t n_series) {
    return j >= n_edges && j < n_series && (j - n_edges) % 3u == 2u;
}

__global__ __launch_bounds__(kSeriesThreads) void af_series_kernel(SeriesArgs a) {
    // per-thread partial sums [thread][4] (u64 integer sum or f64 bits) and maxima, reduced per column
    // in thread order: the result does not depend on scheduling
    __shared__ unsigned long long part_sum[kSeriesThreads][4];
    __shared__ uint32_t part_max[kSeriesThreads][4];
    const int tid = threadIdx.x;
    uint32_t sc;
    if (!claim_scenario(a.done_flags, a.retry, a.scen_map, sc)) return;
    uint32_t ticks = a.counts[(size_t)sc * 8u + a.cnt_ticks_slot];
    if (ticks > a.tick_cap) ticks = a.tick_cap;
    const uint32_t pq = a.pitch / 4u;                       // 16-byte groups per row
    const uint32_t stride = (kSeriesThreads / pq) * pq;     // keeps a thread on one column group
    const uint4* rows = reinterpret_cast<const uint4*>(a.samples + (size_t)sc * a.tick_cap * a.pitch);
    const uint32_t total = ticks * pq;
    const uint32_t col = ((uint32_t)tid % pq) * 4u;
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    if ((uint32_t)tid < stride) {
        for (uint32_t i = tid; i < total; i += stride) {
            const uint4 v = rows[i];
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
            f0 += (double)__uint_as_float(v.x); f1 += (double)__uint_as_float(v.y);
            f2 += (double)__uint_as_float(v.z); f3 += (double)__uint_as_float(v.w);
            m0 = v.x > m0 ? v.x : m0; m1 = v.y > m1 ? v.y : m1;
            m2 = v.z > m2 ? v.z : m2; m3 = v.w > m3 ? v.w : m3;
        }
    }
    part_sum[tid][0] = series_is_float(col + 0u, a.n_edges, a.n_series) ? (unsigned long long)__double_as_longlong(f0) : s0;
    part_sum[tid][1] = series_is_float(col + 1u, a.n_edges, a.n_series) ? (unsigned long long)__double_as_longlong(f1) : s1;
    part_sum[tid][2] = series_is_float(col + 2u, a.n_edges, a.n_series) ? (unsigned long long)__double_as_longlong(f2) : s2;
    part_sum[tid][3] = series_is_float(col + 3u, a.n_edges, a.n_series) ? (unsigned long long)__double_as_longlong(f3) : s3;
    part_max[tid][0] = m0; part_max[tid][1] = m1; part_max[tid][2] = m2; part_max[tid][3] = m3;
    __syncthreads();
    for (uint32_t j = tid; j < a.n_series; j += kSeriesThreads) {
        const uint32_t g = j / 4u, k = j % 4u;
        const bool is_f = series_is_float(j, a.n_edges, a.n_series);
        unsigned long long si = 0;
        double sf = 0.0;
        uint32_t mx = 0;
        for (uint32_t t = g; t < stride; t += pq) {  // the threads of this column group, in order
            if (is_f) sf += __longlong_as_double((long long)part_sum[t][k]);
            else si += part_sum[t][k];
            mx = part_max[t][k] > mx ? part_max[t][k] : mx;
        }
        if (a.mean) a.mean[(size_t)sc * a.n_series + j] = ticks ? (is_f ? sf : (double)si) / (double)ticks : __builtin_nan("");
        if (a.maxv) a.maxv[(size_t)sc * a.n_series + j] = mx;
    }
}

}  // namespace afs
