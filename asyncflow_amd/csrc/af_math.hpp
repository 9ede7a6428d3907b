// af_math.hpp -- counter-based random streams and deterministic elementary
// functions of the MI355X engine (gfx950 device code; also compiles on the host
// for the test-only instantiation under tests/hostcheck/).
//
// Spec (DESIGN.md "Random streams"): Philox4x32-10, key = scenario seed,
// counter = (logical draw index, sub-block, stream id, 0); two 53-bit uniforms
// per block; exponential / normal / log-normal / Poisson variates built only
// from IEEE-754 f64 + - * / sqrt in a fixed order (compile with
// -ffp-contract=off), so device results are reproducible bit for bit on any
// conforming CPU.  Replaces numpy.random.Generator as used by the reference at
// runtime/actors/edge.py:78-90, samplers/poisson_poisson.py:60-70,
// samplers/common_helpers.py:10-89 and runtime/actors/server.py:101.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define AF_HD __host__ __device__ __forceinline__
#else
#define AF_HD inline
#endif

namespace af {

constexpr uint32_t STREAM_GENERATOR = 0u;
AF_HD uint32_t stream_edge(uint32_t e) { return 1u + e; }
AF_HD uint32_t stream_server(uint32_t s) { return 0x1000u + s; }

struct U4 {
    uint32_t x, y, z, w;
};

AF_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

AF_HD double u53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// Block (index, sub) of a stream: uniforms 2*sub and 2*sub+1 of logical draw `index`.
AF_HD U4 draw_block(uint64_t seed, uint32_t stream, uint32_t index, uint32_t sub) {
    return philox4x32_10(index, sub, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

AF_HD double uniform_j(uint64_t seed, uint32_t stream, uint32_t index, uint32_t j) {
    const U4 r = draw_block(seed, stream, index, j >> 1);
    return (j & 1u) ? u53(r.z, r.w) : u53(r.x, r.y);
}

AF_HD uint64_t f64_bits(double x) { return __builtin_bit_cast(uint64_t, x); }
AF_HD double bits_f64(uint64_t u) { return __builtin_bit_cast(double, u); }

AF_HD double af_sqrt(double x) { return __builtin_sqrt(x); }

// Natural logarithm: argument reduction to [sqrt(2)/2, sqrt(2)) and the
// classic degree-14 odd series in s = f/(2+f) (error < 1 ulp).
AF_HD double af_log(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                     L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                     L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                     L7 = 1.479819860511658591e-01;
    uint64_t u = f64_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -__builtin_inf();
        if (hx >> 31) return __builtin_nan("");
        k -= 54;
        x *= 18014398509481984.0;
        u = f64_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = bits_f64(((uint64_t)hx << 32) | (u & 0xffffffffull));

    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (L2 + w * (L4 + w * L6));
    const double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// af_log for a NORMAL x in (0, 1] -- one minus a uniform of [0, 1), the argument of every exponential variate --: the same
// operations without the checks for zero / negative / subnormal / infinite arguments (and without the x == 1 shortcut, whose
// result the general path gives too: f = 0 makes every term +0).  Same bits as af_log on that domain
// (tests/test_gpu_parity.py::test_device_math_matches_oracle_bit_for_bit, kind 7).
AF_HD double af_log_unit(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                     L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                     L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                     L7 = 1.479819860511658591e-01;
    const uint64_t u = f64_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = bits_f64(((uint64_t)hx << 32) | (u & 0xffffffffull));
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (L2 + w * (L4 + w * L6));
    const double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

AF_HD double af_exp(double x) {
    constexpr double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                     invln2 = 1.44269504088896338700e+00;
    constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                     P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                     P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_inf();
    if (x < -745.13321910194110842) return 0.0;
    const double ax = x < 0 ? -x : x;
    double hi, lo;
    int k;
    if (ax > 0.34657359027997264) {
        if (ax >= 1.0397207708399179) {
            k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        } else {
            k = x < 0 ? -1 : 1;
        }
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax > 3.725290298461914e-09) {
        k = 0;
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    const double xx = x * x;
    const double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    const double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    const int k1 = k / 2, k2 = k - k1;
    const double f1 = bits_f64((uint64_t)(0x3ff + k1) << 52);
    const double f2 = bits_f64((uint64_t)(0x3ff + k2) << 52);
    return y * f1 * f2;
}

// Standard normal quantile: Wichura's algorithm AS 241 (PPND16).
AF_HD double af_norminv(double p) {
    const double q = p - 0.5;
    double r, val;
    if ((q < 0 ? -q : q) <= 0.425) {
        r = 0.180625 - q * q;
        return q *
               (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r +
                    45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r +
                 133.14166789178437745) * r + 3.387132872796366608) /
               (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r +
                    21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r +
                 42.313330701600911252) * r + 1.0);
    }
    r = q < 0 ? p : 1.0 - p;
    if (r <= 0.0) return q < 0 ? -__builtin_inf() : __builtin_inf();
    r = af_sqrt(-af_log(r));
    if (r <= 5.0) {
        r -= 1.6;
        val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r +
                   1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r +
                4.6303378461565452959) * r + 1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r +
                   0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.0);
    } else {
        r -= 5.0;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r +
                   0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r +
                5.4637849111641143699) * r + 6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r +
                   7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r +
                0.59983220655588793769) * r + 1.0);
    }
    return q < 0 ? -val : val;
}

// Poisson(mean): mean split into ceil(mean/16) equal parts, each inverted by
// sequential search with one uniform (uniforms j0, j0+1, ... of the draw).
AF_HD int64_t af_poisson(double mean, uint64_t seed, uint32_t stream, uint32_t index, uint32_t j0) {
    if (!(mean > 0.0)) return 0;
    const double nd = mean / 16.0;
    uint32_t nchunks = (uint32_t)nd;
    if ((double)nchunks < nd) nchunks += 1u;
    if (nchunks == 0u) nchunks = 1u;
    const double chunk = mean / (double)nchunks;
    const double p0 = af_exp(-chunk);
    int64_t total = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const double u = uniform_j(seed, stream, index, j0 + c);
        double p = p0, s = p0;
        int k = 0;
        while (u > s && k < 256) {
            k += 1;
            p = p * chunk / (double)k;
            s += p;
        }
        total += k;
    }
    return total;
}

// TEST-ONLY hook, host builds of this header only (g++: tests/hostcheck; never under hipcc, host or device pass):
// latencies and arrival gaps rounded down to a multiple of 2^-bits seconds, so that exact timestamp ties --
// probability ~1e-6 per scenario with continuous laws -- happen thousands of times per scenario.  The oracle has
// the same hook (oracle/des_oracle.c: orc_set_test_quantum); tests/test_flow_hostcheck.py uses both to pin the
// tie handling of the stage-parallel kernel on SimPy's event order.
#if !defined(__HIPCC__)
inline int g_test_quantum_bits = 0;
inline double test_quant(double x) {
    if (g_test_quantum_bits <= 0 || !(x > 0.0)) return x;
    const double s = (double)(1ull << g_test_quantum_bits);
    return __builtin_floor(x * s) / s;
}
#else
AF_HD double test_quant(double x) { return x; }
#endif

enum Dist : uint32_t { DIST_POISSON = 0, DIST_NORMAL = 1, DIST_LOG_NORMAL = 2, DIST_EXPONENTIAL = 3, DIST_UNIFORM = 4 };

// general_sampler (samplers/common_helpers.py:49-89) on the stream spec; `u1`
// is uniform number 1 of the draw (number 0 was the dropout test, edge.py:78).
AF_HD double variate_from_u1(uint32_t dist, double mean, double sigma, double u1, uint64_t seed, uint32_t stream,
                             uint32_t index) {
    switch (dist) {
        case DIST_EXPONENTIAL:
            return -(mean * af_log_unit(1.0 - u1));
        case DIST_UNIFORM:
            return u1;
        case DIST_NORMAL: {
            const double v = mean + sigma * af_norminv(u1);
            return v > 0.0 ? v : 0.0;
        }
        case DIST_LOG_NORMAL:
            return af_exp(mean + sigma * af_norminv(u1));
        default:  // DIST_POISSON
            return (double)af_poisson(mean, seed, stream, index, 1u);
    }
}

}  // namespace af
