/* oracle_rng.h -- ORACLE / TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Counter-based random streams + deterministic elementary functions used by the
 * CPU oracle.  This is the oracle's own statement of the "engine RNG spec"
 * (DESIGN.md section "Random streams"); the HIP engine carries an independent
 * implementation of the same spec in asyncflow_amd/csrc/af_rng.hpp and the two
 * must agree bit for bit (tests/test_rng_spec.py, tests/test_gpu_parity.py).
 *
 * Why not numpy's PCG64 + ziggurat (reference: simulation_runner.py:77,
 * samplers/common_helpers.py:10-89)?  One sequential 128-bit generator shared
 * by every actor cannot be advanced independently by 10^4 scenarios x entities
 * on a GPU.  The reference's own tests inject fake generators through the same
 * seam (tests/unit/runtime/actors/test_edge.py:93-99,
 * tests/integration/single_server/test_int_single_server.py:36); the oracle
 * injects adapters implementing THIS spec (oracle/rng_adapter.py).
 *
 * All functions use only IEEE-754 binary64 + - * / sqrt and integer ops, in a
 * fixed order, compiled with -ffp-contract=off: the result is a pure function
 * of the inputs on any conforming CPU or GPU.
 */
#ifndef AF_ORACLE_RNG_H
#define AF_ORACLE_RNG_H

#include <stdint.h>
#include <string.h>

/* ---- stream ids (counter word 2) -------------------------------------- */
#define ORC_STREAM_GENERATOR 0u
#define ORC_STREAM_EDGE(e) (1u + (uint32_t)(e))
#define ORC_STREAM_SERVER(s) (0x1000u + (uint32_t)(s))

/* ---- Philox4x32-10 (Salmon et al., SC'11) ------------------------------ */
static inline void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* 53-bit uniform in [0,1) from two words (numpy-legacy construction). */
static inline double orc_u53(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

/* Uniform number j (j = 0,1,2,...) of logical draw `index` on `stream` for
 * scenario `seed`.  Block (index, sub=j>>1) yields two uniforms. */
static inline double orc_uniform(uint64_t seed, uint32_t stream, uint32_t index, uint32_t j) {
    uint32_t r[4];
    orc_philox4x32_10(index, j >> 1, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (j & 1u) ? orc_u53(r[2], r[3]) : orc_u53(r[0], r[1]);
}

static inline uint32_t orc_word0(uint64_t seed, uint32_t stream, uint32_t index) {
    uint32_t r[4];
    orc_philox4x32_10(index, 0u, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return r[0];
}

/* ---- deterministic elementary functions -------------------------------- */
static inline uint64_t orc_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double orc_from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* Natural log, FreeBSD/fdlibm e_log.c algorithm (error < 1 ulp). */
static inline double orc_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t u = orc_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -1.0 / 0.0;       /* log(+-0) = -inf */
        if (hx >> 31) return 0.0 / 0.0;             /* log(-#) = NaN   */
        k -= 54;                                     /* subnormal: scale up */
        x *= 18014398509481984.0;                    /* 2^54 */
        u = orc_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;                                    /* inf / nan */
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = orc_from_bits(u);

    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

/* exp, FreeBSD/fdlibm e_exp.c algorithm (error < 1 ulp). */
static inline double orc_exp(double x) {
    static const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return 1.0 / 0.0;
    if (x < -745.13321910194110842) return 0.0;
    double ax = x < 0 ? -x : x;
    double hi, lo;
    int k;
    if (ax > 0.34657359027997264) {                  /* |x| > 0.5 ln2 */
        if (ax >= 1.0397207708399179) {              /* |x| >= 1.5 ln2 */
            k = (int)(invln2 * x + (x < 0 ? -0.5 : 0.5));
        } else {
            k = x < 0 ? -1 : 1;
        }
        hi = x - (double)k * ln2hi;
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (ax > 3.725290298461914e-09) {         /* |x| > 2^-28 */
        k = 0;
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    double xx = x * x;
    double c = x - xx * (P1 + xx * (P2 + xx * (P3 + xx * (P4 + xx * P5))));
    double y = 1.0 + (x * c / (2.0 - c) - lo + hi);
    if (k == 0) return y;
    /* scalbn(y, k) with k in [-1075, 1024], two exact power-of-two factors */
    int k1 = k / 2, k2 = k - k1;
    double f1 = orc_from_bits((uint64_t)(0x3ff + k1) << 52);
    double f2 = orc_from_bits((uint64_t)(0x3ff + k2) << 52);
    return y * f1 * f2;
}

double sqrt(double);

/* Standard normal quantile, Wichura's AS 241 PPND16 (rel. error ~1e-16). */
static inline double orc_norminv(double p) {
    double q = p - 0.5, r, val;
    if ((q < 0 ? -q : q) <= 0.425) {
        r = 0.180625 - q * q;
        val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r +
                        67265.770927008700853) * r + 45921.953931549871457) * r +
                      13731.693765509461125) * r + 1971.5909503065514427) * r +
                    133.14166789178437745) * r + 3.387132872796366608) /
              (((((((r * 5226.495278852545925 + 28729.085735721942674) * r +
                    39307.89580009271061) * r + 21213.794301586595867) * r +
                  5394.1960214247511077) * r + 687.1870074920579083) * r +
                42.313330701600911252) * r + 1.0);
        return val;
    }
    r = q < 0 ? p : 1.0 - p;
    if (r <= 0.0) return q < 0 ? -1.0 / 0.0 : 1.0 / 0.0;
    r = sqrt(-orc_log(r));
    if (r <= 5.0) {
        r -= 1.6;
        val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r +
                    0.24178072517745061177) * r + 1.27045825245236838258) * r +
                  3.64784832476320460504) * r + 5.7694972214606914055) * r +
                4.6303378461565452959) * r + 1.42343711074968357734) /
              (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r +
                    0.0151986665636164571966) * r + 0.14810397642748007459) * r +
                  0.68976733498510000455) * r + 1.6763848301838038494) * r +
                2.05319162663775882187) * r + 1.0);
    } else {
        r -= 5.0;
        val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r +
                    0.0012426609473880784386) * r + 0.026532189526576123093) * r +
                  0.29656057182850489123) * r + 1.7848265399172913358) * r +
                5.4637849111641143699) * r + 6.6579046435011037772) /
              (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r +
                    1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
                  0.0148753612908506148525) * r + 0.13692988092273580531) * r +
                0.59983220655588793769) * r + 1.0);
    }
    return q < 0 ? -val : val;
}

/* ---- variates ----------------------------------------------------------- */
/* Poisson(mean) by chunked inversion: mean is split into n = ceil(mean/16)
 * equal parts; each part is inverted by sequential search with ONE uniform.
 * Uniforms j0, j0+1, ... of logical draw (stream, index) are consumed. */
static inline int64_t orc_poisson(double mean, uint64_t seed, uint32_t stream, uint32_t index,
                                  uint32_t j0) {
    if (!(mean > 0.0)) return 0;
    double nchunks_d = mean / 16.0;
    uint32_t nchunks = (uint32_t)nchunks_d;
    if ((double)nchunks < nchunks_d) nchunks += 1u;
    if (nchunks == 0u) nchunks = 1u;
    double chunk = mean / (double)nchunks;
    double p0 = orc_exp(-chunk);
    int64_t total = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        double u = orc_uniform(seed, stream, index, j0 + c);
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

#define ORC_DIST_POISSON 0
#define ORC_DIST_NORMAL 1
#define ORC_DIST_LOG_NORMAL 2
#define ORC_DIST_EXPONENTIAL 3
#define ORC_DIST_UNIFORM 4

/* Restates general_sampler (samplers/common_helpers.py:49-89) on the spec:
 * the FIRST uniform consumed is number j0 of logical draw (stream,index). */
static inline double orc_variate(int dist, double mean, double sigma, uint64_t seed,
                                 uint32_t stream, uint32_t index, uint32_t j0) {
    switch (dist) {
        case ORC_DIST_UNIFORM:      /* common_helpers.py:62-65: U[0,1), mean ignored */
            return orc_uniform(seed, stream, index, j0);
        case ORC_DIST_POISSON:      /* :67-70 */
            return (double)orc_poisson(mean, seed, stream, index, j0);
        case ORC_DIST_EXPONENTIAL:  /* :72-75: scale == mean */
            return -(mean * orc_log(1.0 - orc_uniform(seed, stream, index, j0)));
        case ORC_DIST_NORMAL: {     /* :78-80 -> truncated_gaussian_generator :23-33 (sigma=variance) */
            double v = mean + sigma * orc_norminv(orc_uniform(seed, stream, index, j0));
            return v > 0.0 ? v : 0.0;
        }
        case ORC_DIST_LOG_NORMAL:   /* :82-84 -> rng.lognormal(mean, sigma) */
            return orc_exp(mean + sigma * orc_norminv(orc_uniform(seed, stream, index, j0)));
        default:
            return 0.0 / 0.0;
    }
}

#endif /* AF_ORACLE_RNG_H */
