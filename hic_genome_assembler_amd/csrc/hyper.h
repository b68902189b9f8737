// hyper.h - upper-tail hypergeometric probability P[X >= x], X ~ Hypergeom(M, n, N), in fp64,
// compiled for BOTH the gfx950 kernels and the host side of libhicmi.so.
//
// Replaces scipy.stats.hypergeom.sf(x-1, M, n, N) as called by the reference's hyper_geom
// (scaffoldToChromosomes.py:352-368; SciPy delegates to Boost's hypergeometric_distribution).
// Argument checking follows SciPy (_discrete_distns.py: hypergeom._argcheck):
// invalid arguments give NaN, below the support gives 1, at/above its top gives 0.
//
// The point mass is evaluated with the saddle-point expansion of C. Loader, "Fast and accurate
// computation of binomial probabilities" (2000) - log-gamma differences of order 1e5..1e6 would
// lose ~6 digits at M ~ 64000 - and the tail is summed from x outwards with the exact term ratio.
// Only the comparison against psig reaches the outputs (S2C:466-469, 633-636, 669); tests compare
// both the value (1e-10 relative) and every decision against SciPy.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HICMI_HD __host__ __device__ inline
#else
#define HICMI_HD inline
#endif

namespace hicmi {

HICMI_HD double stirlerr(double n)   // ln(n!) - ln( sqrt(2 pi n) (n/e)^n ), integer n >= 0
{
    const double S0 = 0.083333333333333333333;        // 1/12
    const double S1 = 0.00277777777777777777778;      // 1/360
    const double S2 = 0.00079365079365079365079365;   // 1/1260
    const double S3 = 0.000595238095238095238095238;  // 1/1680
    const double S4 = 0.0008417508417508417508417508; // 1/1188
    if (n <= 15.0) {
        switch ((int)n) {
        case 0: return 0.0;
        case 1: return 0.08106146679532725821967026;
        case 2: return 0.04134069595540929409382208;
        case 3: return 0.02767792568499833914878929;
        case 4: return 0.02079067210376509311152277;
        case 5: return 0.01664469118982119216319487;
        case 6: return 0.01387612882307074799874573;
        case 7: return 0.01189670994589177009505572;
        case 8: return 0.01041126526197209649747857;
        case 9: return 0.009255462182712732917728637;
        case 10: return 0.008330563433362871256469319;
        case 11: return 0.007573675487951840794972024;
        case 12: return 0.006942840107209529865664153;
        case 13: return 0.006408994188004207068439631;
        case 14: return 0.005951370112758847735624416;
        default: return 0.00555473355196280137103869;
        }
    }
    double nn = n * n;
    if (n > 500.0) return (S0 - S1 / nn) / n;
    if (n > 80.0) return (S0 - (S1 - S2 / nn) / nn) / n;
    if (n > 35.0) return (S0 - (S1 - (S2 - S3 / nn) / nn) / nn) / n;
    return (S0 - (S1 - (S2 - (S3 - S4 / nn) / nn) / nn) / nn) / n;
}

HICMI_HD double bd0(double x, double np)   // x ln(x/np) + np - x, stable near x == np
{
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        double ej = 2.0 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; j++) {
            ej *= v;
            double s1 = s + ej / (double)((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
        return s;
    }
    return x * log(x / np) + np - x;
}

HICMI_HD double dbinom_raw(double x, double n, double p, double q)
{
    if (p == 0.0) return x == 0.0 ? 1.0 : 0.0;
    if (q == 0.0) return x == n ? 1.0 : 0.0;
    if (x == 0.0) {
        if (n == 0.0) return 1.0;
        double lc = (p < 0.1) ? -bd0(n, n * q) - n * p : n * log(q);
        return exp(lc);
    }
    if (x == n) {
        double lc = (q < 0.1) ? -bd0(n, n * p) - n * q : n * log(p);
        return exp(lc);
    }
    if (x < 0.0 || x > n) return 0.0;
    double lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q);
    double lf = 1.837877066409345483560659 /* ln(2 pi) */ + log(x) + log1p(-x / n);
    return exp(lc - 0.5 * lf);
}

// P[X = k]; r = marked items (SciPy n), b = unmarked (M - n), d = draws (SciPy N)
HICMI_HD double dhyper(double k, double r, double b, double d)
{
    if (d < k || r < k || d - k > b || k < 0.0) return 0.0;
    if (d == 0.0) return k == 0.0 ? 1.0 : 0.0;
    double p = d / (r + b), q = (r + b - d) / (r + b);
    double p1 = dbinom_raw(k, r, p, q);
    double p2 = dbinom_raw(d - k, b, p, q);
    double p3 = dbinom_raw(d, r + b, p, q);
    return p1 * p2 / p3;
}

// hyper_geom(x, M, n, N) of the reference: P[X >= x].
HICMI_HD double hypergeom_sf_ge(int64_t x, int64_t M, int64_t n, int64_t N)
{
    if (!(M > 0 && n >= 0 && N >= 0 && n <= M && N <= M)) return NAN;
    int64_t lo = N - (M - n); if (lo < 0) lo = 0;
    int64_t hi = n < N ? n : N;
    if (x <= lo) return 1.0;           // sf(k) with k = x-1 below the support
    if (x > hi) return 0.0;            // k >= top of the support
    const double r = (double)n, b = (double)(M - n), d = (double)N;
    // mode of the distribution
    int64_t mode = (int64_t)floor(((double)(n + 1) * (double)(N + 1)) / (double)(M + 2));
    if (x > mode) {
        // upper tail: terms decrease monotonically
        double term = dhyper((double)x, r, b, d);
        double sum = term;
        for (int64_t k = x; k < hi; k++) {
            // pmf(k+1)/pmf(k) = (n-k)(N-k) / ((k+1)(M-n-N+k+1))
            double num = (double)(n - k) * (double)(N - k);
            double den = (double)(k + 1) * (double)(M - n - N + k + 1);
            term *= num / den;
            double s1 = sum + term;
            if (s1 == sum) break;
            sum = s1;
        }
        return sum > 1.0 ? 1.0 : sum;
    }
    // x <= mode: 1 - P[X <= x-1], lower tail summed downwards from x-1
    double term = dhyper((double)(x - 1), r, b, d);
    double sum = term;
    for (int64_t k = x - 1; k > lo; k--) {
        // pmf(k-1)/pmf(k) = k (M-n-N+k) / ((n-k+1)(N-k+1))
        double num = (double)k * (double)(M - n - N + k);
        double den = (double)(n - k + 1) * (double)(N - k + 1);
        term *= num / den;
        double s1 = sum + term;
        if (s1 == sum) break;
        sum = s1;
    }
    double sf = 1.0 - sum;
    return sf < 0.0 ? 0.0 : sf;
}

// The only thing the reference does with hyper_geom is compare it with psig (S2C:466-469, 633-636, 669).
// hypergeom_decide returns 1 if hyper_geom(x, M, n, N) < psig, 0 if >= psig, -1 if it is NaN - the same decision as
// comparing hypergeom_sf_ge's value, but the tail sum stops as soon as the comparison is settled:
//   * the partial sums are monotone, so a partial sum beyond the threshold settles it exactly;
//   * the pmf is log-concave: walking away from the mode the term ratio only falls, so with the current ratio r < 1
//     the rest of the tail is below the geometric series term * r / (1 - r); when even that cannot reach the threshold
//     (with a 1e-9 relative margin) it is settled the other way.  Written without the division:
//     rest < gap  <=>  term * r < gap * (1 - r).  Inside the margin the loop just goes on to the full sum, i.e. to
//     hypergeom_sf_ge's own value.
// A row well inside a cluster (x far above the mode) needs a few terms instead of ~50, a row near the mode (p ~ 0.5)
// tens instead of hundreds - and the slowest row of a scan is what a scan waits for.
HICMI_HD int hypergeom_decide(int64_t x, int64_t M, int64_t n, int64_t N, double psig)
{
    if (!(M > 0 && n >= 0 && N >= 0 && n <= M && N <= M)) return -1;
    int64_t lo = N - (M - n); if (lo < 0) lo = 0;
    int64_t hi = n < N ? n : N;
    if (x <= lo) return 1.0 < psig ? 1 : 0;
    if (x > hi) return 0.0 < psig ? 1 : 0;
    const double r = (double)n, b = (double)(M - n), d = (double)N;
    const double below = psig * (1.0 - 1e-9), above = psig * (1.0 + 1e-9);
    int64_t mode = (int64_t)floor(((double)(n + 1) * (double)(N + 1)) / (double)(M + 2));
    if (x > mode) {
        double term = dhyper((double)x, r, b, d);
        double sum = term;
        if (!(sum < psig)) return sum != sum ? -1 : 0;
        for (int64_t k = x; k < hi; k++) {
            double num = (double)(n - k) * (double)(N - k);
            double den = (double)(k + 1) * (double)(M - n - N + k + 1);
            const double ratio = num / den;
            term *= ratio;
            double s1 = sum + term;
            if (s1 == sum) break;
            sum = s1;
            if (!(sum < psig)) return 0;                          // only grows from here
            if (ratio < 1.0 && term * ratio < (below - sum) * (1.0 - ratio)) return 1;   // the rest cannot reach psig
        }
        return (sum > 1.0 ? 1.0 : sum) < psig ? 1 : 0;
    }
    double term = dhyper((double)(x - 1), r, b, d);
    double sum = term;
    if (1.0 - sum < psig) return 1;                               // sf = 1 - sum only shrinks from here
    for (int64_t k = x - 1; k > lo; k--) {
        double num = (double)k * (double)(M - n - N + k);
        double den = (double)(n - k + 1) * (double)(N - k + 1);
        const double ratio = num / den;
        term *= ratio;
        double s1 = sum + term;
        if (s1 == sum) break;
        sum = s1;
        if (1.0 - sum < psig) return 1;
        if (ratio <= 0.5 && 1.0 - (sum + term) > above) return 0;
    }
    double sf = 1.0 - sum;
    if (sf < 0.0) sf = 0.0;
    return sf < psig ? 1 : 0;
}

}  // namespace hicmi
