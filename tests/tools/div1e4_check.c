/* Proof by exhaustion for k1_emit's div1e4(): for every int32 x,  x / 10000.0  (IEEE double division) equals
 *   q0 = x * r;  q = fma(fma(-q0, 10000.0, x), r, q0)   with r = RN(1 / 10000.0)
 * bit for bit (signed zero included).  Run by tests/test_numerics.py (2 s on 8 threads). */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <pthread.h>
static const double D = 10000.0;
static volatile long bad[16];
typedef struct { int64_t lo, hi; int id; } job;
static void *run(void *p) {
    job *j = (job *)p;
    const double r = 1.0 / D;            /* correctly rounded reciprocal */
    long nb = 0;
    for (int64_t v = j->lo; v < j->hi; ++v) {
        const double x = (double)(int32_t)v;
        const double q0 = x * r;
        const double rem = fma(-q0, D, x);
        const double q = fma(rem, r, q0);
        const double t = x / D;
        if (q != t || (q == 0.0 && signbit(q) != signbit(t))) { if (nb < 3) printf("x=%lld q=%a t=%a\n", (long long)v, q, t); ++nb; }
    }
    bad[j->id] = nb;
    return 0;
}
int main(void) {
    pthread_t th[8]; job jobs[8];
    const int64_t lo = -2147483648LL, hi = 2147483648LL, step = (hi - lo) / 8;
    for (int i = 0; i < 8; ++i) { jobs[i].lo = lo + i * step; jobs[i].hi = i == 7 ? hi : lo + (i + 1) * step; jobs[i].id = i; pthread_create(&th[i], 0, run, &jobs[i]); }
    long tot = 0;
    for (int i = 0; i < 8; ++i) { pthread_join(th[i], 0); tot += bad[i]; }
    printf("mismatches over all int32: %ld\n", tot);
    return 0;
}
