// Rows of text on the device (mc_rowtext.hip): the pieces of a .diffs row that are numbers, for a host and a device compiler alike
// (tests/test_numerics.py runs the host build of these against repr() -- mc_repr_double_rowtext).
//
// repr(float) is the shortest decimal that reads back as the same double, of several shortest ones the closest (half-way: the even digit)
// (CPython: David Gay's dtoa, mode 0; the reference writes its slot means with it, extract_contexts.py:207-216 via str()).
// The host formatter gets those digits from std::to_chars (mc_format.cpp); here they come from the free-format digit generation
// of Steele & White / Burger & Dybvig in exact integer arithmetic: v = f * 2^e, the gaps to its neighbours m+ and m-, all
// scaled to integers r / s; digits are peeled off r while the rest is still further from v than the gap.  For the doubles a row
// holds (slot means, read qualities: RT_REPR_MIN <= |v| < RT_REPR_MAX -- a slot mean that should be zero is 1e-17 or 1e-20 when its
// sum left a rounding residue) every quantity stays below 2^116: two 64-bit words, no tables.  Anything outside that range is refused (-> the row, and with it the shard, goes to the host formatter).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP__)
#define RT_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define RT_HD inline
#endif

typedef unsigned __int128 rt_u128;

#define RT_REPR_MIN 1e-29
#define RT_REPR_MAX 1e9

// Up to 17 decimal digits, four bits each, first digit in the highest place: no array (on the device an array indexed by a running
// count lives in registers and every access is a chain of selects)
struct RtDigits {
    uint64_t lo = 0;            // the last 16 digits pushed
    unsigned hi = 0;            // what was pushed out of lo (the first digit of 17)
    int n = 0;
    RT_HD void push(int d) { hi = (unsigned)(lo >> 60); lo = (lo << 4) | (uint64_t)(unsigned)d; ++n; }
    RT_HD int at(int i) const {                                       // digit i, 0 = the first
        const int back = n - 1 - i;                                   // places from the last
        return back >= 16 ? (int)hi : (int)((lo >> (4 * back)) & 15u);
    }
    RT_HD int last() const { return (int)(lo & 15u); }
    RT_HD void pop() { lo = (lo >> 4) | ((uint64_t)hi << 60); hi = 0; --n; }
    RT_HD void raise_last() { lo += 1; }
};

// floor(r / s) for r < 10 s: an estimate from the leading bits, put right (64-bit integers: the quotient of two floats is off by
// less than one either way)
RT_HD int rt_small_quotient(uint64_t r, uint64_t s) {
    int d = (int)((float)r / (float)s);
    d = d > 9 ? 9 : d;
    uint64_t p = (uint64_t)(unsigned)d * s;
    if (p > r) { --d; p -= s; }
    if (r - p >= s) ++d;
    return d;
}

// The digit generation proper: r / s = v / 10^k in [0.1, 1), mp / s and mm / s the gaps to the neighbours' mid-points above and below,
// in integers of type U (128 bits wide, or 64 when s < 2^60: r * 10, r + mp * 10 and 2 r all stay below 11 s).
template <class U>
RT_HD void rt_generate(U r, U s, U mp, U mm, bool even, RtDigits &dig, int *decpt) {
    for (;;) {
        r *= 10u; mp *= 10u; mm *= 10u;
        int d = 0;
        if (sizeof(U) == 8) { d = rt_small_quotient((uint64_t)r, (uint64_t)s); r -= (U)((uint64_t)(unsigned)d * (uint64_t)s); }
        else { while (r >= s && d < 10) { r -= s; ++d; } }            // d = floor(r / s) <= 9
        const bool tc1 = even ? (r <= mm) : (r < mm);                 // the digits so far, as they are, read back as v
        const bool tc2 = even ? (r + mp >= s) : (r + mp > s);         // ... and so they do with the last one raised by one
        if (!tc1 && !tc2 && dig.n < 16) { dig.push(d); continue; }
        if (tc2 && (!tc1 || (r << 1) > s || ((r << 1) == s && (d & 1)))) ++d;      // (both: the closer one, half-way to the even digit -- dtoa.c)
        if (d < 10) dig.push(d);
        else {                                                        // (a raised 9: the generation is known not to need this)
            while (dig.n > 0 && dig.last() == 9) dig.pop();
            if (dig.n == 0) { dig.push(1); *decpt += 1; }
            else dig.raise_last();
        }
        break;
    }
    while (dig.n > 1 && dig.last() == 0) dig.pop();
}

// Digits (no trailing zeros) of the shortest decimal 0.d1d2...dn * 10^decpt that reads back as v;
// RT_REPR_MIN <= v < RT_REPR_MAX.  n is at most 17
RT_HD void rt_shortest_digits(double v, RtDigits &dig, int *decpt) {
    uint64_t bits;
    __builtin_memcpy(&bits, &v, 8);
    const uint64_t f = (bits & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    const int e = (int)((bits >> 52) & 0x7FF) - 1075;                 // v = f * 2^e, -149 <= e <= -23 in the range above
    const bool even = (f & 1ull) == 0;                                // (the interval's ends read back as v iff f is even)
    // (a power of two: the neighbour below is half as far)
    const bool pow2 = f == (1ull << 52);
    rt_u128 r = (rt_u128)f << (pow2 ? 2 : 1), s, mp = pow2 ? 2 : 1, mm = 1;
    int sh = (pow2 ? 2 : 1) - e;                                      // s = 2^sh
    // k with 10^(k-1) <= high < 10^k, high = (r + m+) / s the upper end of the interval: a guess from the binary exponent
    // (floor(log2 v) * log10(2), one too small at most) ...
    int k = (((e + 52) * 78913) >> 18) + 1;
    if (k >= 0) {
        const uint64_t ipow[11] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                   10000000000ull};
        s = ((rt_u128)1 << sh) * ipow[k];
    } else {
        // v / 10^k = v * 5^-k * 2^-k: the powers of five into r and the gaps, the powers of two out of s -- the numbers stay
        // below 2^116 down to 1e-29
        int j = -k;
        const uint64_t p5[14] = {1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull, 1953125ull, 9765625ull, 48828125ull,
                                 244140625ull, 1220703125ull};
        s = (rt_u128)1 << (sh - j);
        while (j > 0) { const int t = j > 13 ? 13 : j; r *= p5[t]; mp *= p5[t]; mm *= p5[t]; j -= t; }
    }
    // ... put right
    if (even ? (r + mp >= s) : (r + mp > s)) { s *= 10u; ++k; }
    else if (even ? ((r + mp) * 10u < s) : ((r + mp) * 10u <= s)) { r *= 5u; mp *= 5u; mm *= 5u; s >>= 1; --k; }
    *decpt = k;
    // every slot mean from a thousandth up: 64-bit integers do
    if (s < ((rt_u128)1 << 60)) rt_generate<uint64_t>((uint64_t)r, (uint64_t)s, (uint64_t)mp, (uint64_t)mm, even, dig, decpt);
    else rt_generate<rt_u128>(r, s, mp, mm, even, dig, decpt);
}

// Where the characters go: counted (the pass that sizes the rows) or stored
struct RtCount {
    int n = 0;
    RT_HD void put(char) { ++n; }
};
struct RtStore {
    char *p;
    RT_HD void put(char c) { *p++ = c; }
};

// A double as its printed digits: what the digit generation leaves and the layout needs -- 16 bytes when it waits in memory between
// the kernel that makes the digits (a lane per NUMBER: every lane in the generation) and the kernels that lay out rows (a lane per row)
struct RtNum {
    RtDigits dig;
    int decpt = 0;
    bool neg = false, zero = false, ok = false;     // ok: a double this code prints
};

RT_HD RtNum rt_num_of(double v) {
    RtNum n;
    if (!(v == v)) return n;
    uint64_t bits;
    __builtin_memcpy(&bits, &v, 8);
    if (bits >> 63) { n.neg = true; bits &= ~(1ull << 63); __builtin_memcpy(&v, &bits, 8); }
    if (v == 0.0) { n.zero = n.ok = true; return n; }
    if (!(v >= RT_REPR_MIN && v < RT_REPR_MAX)) return n;
    rt_shortest_digits(v, n.dig, &n.decpt);
    n.ok = true;
    return n;
}

// meta: bits 0-3 the first of 17 digits, 4-8 the number of digits, 9-16 decpt + 64, 17 negative, 18 zero, 19 ok
RT_HD void rt_num_pack(const RtNum &n, uint64_t *lo, uint32_t *meta) {
    *lo = n.dig.lo;
    *meta = (n.dig.hi & 15u) | ((uint32_t)n.dig.n << 4) | ((uint32_t)(n.decpt + 64) << 9) | (n.neg ? 1u << 17 : 0u) | (n.zero ? 1u << 18 : 0u) |
            (n.ok ? 1u << 19 : 0u);
}
RT_HD RtNum rt_num_unpack(uint64_t lo, uint32_t meta) {
    RtNum n;
    n.dig.lo = lo; n.dig.hi = meta & 15u; n.dig.n = (int)((meta >> 4) & 31u);
    n.decpt = (int)((meta >> 9) & 255u) - 64;
    n.neg = (meta >> 17) & 1u; n.zero = (meta >> 18) & 1u; n.ok = (meta >> 19) & 1u;
    return n;
}

// the characters of an RtNum (ok), Python's layout (float_repr_style 'short': exponent form iff decpt > 16 or decpt < -3, at least two
// exponent digits, ".0" behind an integer)
template <class Sink>
RT_HD void rt_put_num(Sink &o, const RtNum &n) {
    if (n.neg) o.put('-');
    if (n.zero) { o.put('0'); o.put('.'); o.put('0'); return; }
    const RtDigits &dig = n.dig;
    const int nd = dig.n, decpt = n.decpt;
    if (decpt < -3) {                                                 // d[.ddd]e-XX
        o.put((char)('0' + dig.at(0)));
        if (nd > 1) { o.put('.'); for (int i = 1; i < nd; ++i) o.put((char)('0' + dig.at(i))); }
        o.put('e'); o.put('-');
        const int ex = 1 - decpt;                                     // 5 .. 29 here
        o.put((char)('0' + ex / 10)); o.put((char)('0' + ex % 10));
        return;
    }
    if (decpt <= 0) {
        o.put('0'); o.put('.');
        for (int i = 0; i < -decpt; ++i) o.put('0');
        for (int i = 0; i < nd; ++i) o.put((char)('0' + dig.at(i)));
        return;
    }
    if (decpt >= nd) {
        for (int i = 0; i < nd; ++i) o.put((char)('0' + dig.at(i)));
        for (int i = nd; i < decpt; ++i) o.put('0');
        o.put('.'); o.put('0');
        return;
    }
    for (int i = 0; i < decpt; ++i) o.put((char)('0' + dig.at(i)));
    o.put('.');
    for (int i = decpt; i < nd; ++i) o.put((char)('0' + dig.at(i)));
}

// how many characters rt_put_num writes
RT_HD int rt_num_length(const RtNum &n) {
    int len = n.neg ? 1 : 0;
    if (n.zero) return len + 3;
    const int nd = n.dig.n, decpt = n.decpt;
    if (decpt < -3) return len + nd + (nd > 1 ? 1 : 0) + 4;
    if (decpt <= 0) return len + 2 - decpt + nd;
    if (decpt >= nd) return len + decpt + 2;
    return len + nd + 1;
}

// repr(v) -> false: not a double this code prints (nan, inf, outside the range above)
template <class Sink>
RT_HD bool rt_put_repr(Sink &o, double v) {
    const RtNum n = rt_num_of(v);
    if (!n.ok) return false;
    rt_put_num(o, n);
    return true;
}

// a non-negative integer below 2^32, decimal (divisions by constants: multiplications)
template <class Sink>
RT_HD void rt_put_uint(Sink &o, uint32_t a) {
    uint32_t p = 1000000000u;
    bool seen = false;
    for (int i = 0; i < 10; ++i) {
        const uint32_t d = a / p;
        a -= d * p;
        seen = seen || d != 0 || i == 9;
        if (seen) o.put((char)('0' + d));
        p /= 10u;
    }
}

// repr(d / 1e4) from the integer d (mc_format.cpp put_fixed4: at most ten significant digits ARE the shortest digits)
template <class Sink>
RT_HD void rt_put_fixed4(Sink &o, int32_t d) {
    uint32_t a = d < 0 ? (uint32_t)(-(int64_t)d) : (uint32_t)d;
    if (d < 0) o.put('-');
    const uint32_t ip = a / 10000u;
    uint32_t fp = a % 10000u;
    rt_put_uint(o, ip);
    o.put('.');
    if (fp == 0) { o.put('0'); return; }
    const char f4[4] = {(char)('0' + fp / 1000u), (char)('0' + fp / 100u % 10u), (char)('0' + fp / 10u % 10u), (char)('0' + fp % 10u)};
    int nf = 4;
    while (f4[nf - 1] == '0') --nf;
    for (int i = 0; i < nf; ++i) o.put(f4[i]);
}

// str(np.round(p, 2)) of a probability (extract_contexts.py:207): hundredths -> false: not in [0, 1] (the host's general path)
template <class Sink>
RT_HD bool rt_put_prob2(Sink &o, double p1, double hundredths /* nearbyint(p1 * 100) */) {
    (void)p1;
    if (!(hundredths >= 0.0 && hundredths <= 100.0)) return false;
    const int hi = (int)hundredths;
    if (hi == 100) { o.put('1'); o.put('.'); o.put('0'); return true; }
    o.put('0'); o.put('.'); o.put((char)('0' + hi / 10));
    if (hi % 10) o.put((char)('0' + hi % 10));
    return true;
}
