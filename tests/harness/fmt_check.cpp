// tests/harness/fmt_check.cpp — TEST INFRASTRUCTURE: pdh::fmt2_to (exact "%.2f" without printf) against snprintf.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <random>
#include "../../pandepth_amd/host/report.h"

static long bad = 0, total = 0;
static void check(double v)
{
    char a[64], b[64];
    pdh::fmt2_to(a, v);
    snprintf(b, sizeof b, "%.2f", v);
    ++total;
    if (strcmp(a, b)) { if (bad < 10) printf("MISMATCH %a: %s vs %s\n", v, a, b); ++bad; }
}

int main()
{
    std::mt19937_64 rng(12345);
    // ratios as the tables compute them
    for (int i = 0; i < 4000000; ++i) {
        const uint64_t L = 1 + rng() % (i % 3 ? 20000 : 4000000000ull);
        const uint64_t C = rng() % (L + 1), D = rng() % (i % 5 ? L * 300 + 1 : (uint64_t)1 << (rng() % 63));
        check(C * 100.0 / L);
        check(D * 1.0 / L);
    }
    // exact ties and their neighbours: odd multiples of 1/8, +-1 ulp
    for (uint64_t k = 1; k < 2000000; k += 2) {
        const double t = (double)k / 8.0;
        check(t); check(nextafter(t, 0)); check(nextafter(t, 1e300));
        const double u = (double)(k * 1000003) / 8.0;
        check(u); check(nextafter(u, 0)); check(nextafter(u, 1e300));
    }
    // x.xx5 decimals (never exact in binary), powers of two, the edges of the fast path
    for (int i = 0; i < 1000000; ++i) check((double)(rng() % 100000000) / 1000.0);
    for (int e = -1080; e < 80; ++e) { const double p = ldexp(1.0, e); check(p); check(p * 1.5); check(nextafter(p, 0)); }
    const double specials[] = {0.0, -0.0, 0.004999999999999999, 0.005, 0.0050000000000000001, 0.995, 0.994999999999, 99.995, 4503599627370495.5,
                               4503599627370496.0, 9007199254740992.0, 1e300, -1.5, -0.004, 1.0 / 0.0, -1.0 / 0.0, 0.0 / 0.0};
    for (double s : specials) check(s);
    // random bit patterns
    for (int i = 0; i < 3000000; ++i) { uint64_t b = rng(); double v; memcpy(&v, &b, 8); check(v); }
    printf("%ld values, %ld mismatches\n", total, bad);
    return bad != 0;
}
