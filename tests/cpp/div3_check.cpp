// Host check of the division-free a/3 that k_lex_wg's border body uses (ccp_grid_lex.hpp, lex_div3): the same
// three IEEE operations — multiply by RN(1/3), exact residual in an fma, correction in an fma — against the
// machine's correctly rounded division, on random significands over EVERY binade (subnormals and the largest
// finite numbers included), on the neighbours of every multiple of 3 near binade edges, on results that are
// exactly representable and on signed zeros.  (Infinities and NaNs are the only inputs the kernel divides.)
// Prints "ok <cases>" or the first mismatch.  Built with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

static double div3(double a)
{
    const double y = 0x1.5555555555555p-2;
    const double q0 = a * y;
    const double r = std::fma(-3.0, q0, a);
    const double q1 = std::fma(r, y, q0);
    return r == 0.0 ? q0 : q1;
}

static bool same(double a, double b)
{
    uint64_t x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    return x == y;
}

static long checked = 0;
static bool check(double a)
{
    ++checked;
    volatile double d = 3.0;
    const double want = a / d;
    const double got = div3(a);
    if (same(want, got)) return true;
    std::printf("MISMATCH a=%a want=%a got=%a\n", a, want, got);
    return false;
}

int main(int argc, char **argv)
{
    long n = argc > 1 ? std::atol(argv[1]) : 20000000;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    };
    for (long i = 0; i < n; ++i) {
        const uint64_t m = next() & ((1ull << 52) - 1);
        const int e = (int)(next() % 2047);                            // every exponent field: subnormals .. 2^1023
        uint64_t bits = ((uint64_t)(next() & 1) << 63) | ((uint64_t)e << 52) | m;
        double a;
        std::memcpy(&a, &bits, 8);
        if (!check(a)) return 1;
    }
    for (int e = -60; e <= 60; ++e)
        for (long k = -2000; k <= 2000; ++k) {
            const double base = std::ldexp(1.0, e);
            for (double a : {base, 3.0 * base, base * 1.5}) {                   // binade edges, exact quotients
                double v = a;
                for (long q = 0; q < (k < 0 ? -k : k); ++q) v = std::nextafter(v, k < 0 ? -INFINITY : INFINITY);
                if (!check(v) || !check(-v)) return 1;
                if (k < -40 || k > 40) break;
            }
        }
    for (long k = 0; k < 3000000; ++k)                                           // small integers and thirds of them
        if (!check((double)k) || !check((double)k * 0.25) || !check(-(double)k / 7.0)) return 1;
    for (int e = 0; e < 2047; ++e)                                               // both ends of every binade
        for (uint64_t m : {0ull, 1ull, 2ull, 3ull, (1ull << 52) - 1, (1ull << 52) - 2, (1ull << 51), (1ull << 51) + 1, 0x5555555555555ull, 0xAAAAAAAAAAAAAull}) {
            const uint64_t bits = ((uint64_t)e << 52) | m;
            double a;
            std::memcpy(&a, &bits, 8);
            if (!check(a) || !check(-a)) return 1;
        }
    if (!check(0.0) || !check(-0.0)) return 1;
    std::printf("ok %ld\n", checked);
    return 0;
}
