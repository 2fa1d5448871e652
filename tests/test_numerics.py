"""NumPy semantics the path depends on, restated in the oracle and in the host formatter."""
import os
import random

import pytest

import numpy as np

from oracle import py_oracle as po
from mcaller_amd.extract_contexts import fmt_float, round2


def test_pairwise_mean_matches_numpy():
    rng = random.Random(3)
    for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 130, 136, 137, 255, 256, 257, 300, 1000, 1031]:
        for _ in range(20):
            vals = [round(rng.gauss(-0.17, 2.44), 2) for _ in range(n)]
            assert po.np_mean(vals) == float(np.mean([np.float64(v) for v in vals])), n


def test_round4_is_integer_division():
    """np.round(float(e) - float(m), 4) == fl((E4 - M4) / 1e4) for 2- to 4-decimal prints."""
    rng = random.Random(5)
    for dec in (2, 3, 4):
        for _ in range(20000):
            e = round(rng.uniform(40, 140), dec)
            m = round(rng.uniform(40, 140), dec)
            es, ms = ('%.' + str(dec) + 'f') % e, ('%.' + str(dec) + 'f') % m
            want = float(np.round(float(es) - float(ms), 4))
            e4 = int(round(float(es) * 10000)); m4 = int(round(float(ms) * 10000))
            assert want == (e4 - m4) / 10000.0
            assert po.round_dec(float(es) - float(ms), 4) == want


def test_float_formatting_matches_numpy_str():
    rng = random.Random(7)
    vals = [0.0, -0.0, 1.0, -2.5, 1e-5, 1e-4, 123456789.125, 1e16, 0.1 + 0.2, 7.055265349382997]
    vals += [rng.uniform(-10, 10) for _ in range(5000)] + [rng.uniform(-1e-3, 1e-3) for _ in range(2000)]
    for v in vals:
        assert fmt_float(v) == str(np.float64(v)), v
        assert po.fmt_float(v) == str(np.float64(v))
    for _ in range(5000):
        p = rng.random()
        assert fmt_float(round2(p)) == str(np.round(np.float64(p), 2))
        assert po.round_dec(p, 2) == float(np.round(np.float64(p), 2))


def test_native_repr_matches_python():
    """mc_repr_double (the native .diffs formatter's number printing) == repr(float) == str(np.float64)."""
    import struct
    from mcaller_amd._lib import repr_double
    rng = random.Random(11)
    vals = [0.0, -0.0, 1.0, -2.5, 1e-5, 1e-4, 9.999e-5, 0.0001, 0.001, 123456789.125, 1e15, 1e16, 9999999999999998.0, 1e17,
            1.5e300, 5e-324, 2.2250738585072014e-308, 0.1 + 0.2, 7.055265349382997, float('inf'), float('-inf'), 100.0, 12.0]
    vals += [rng.uniform(-10, 10) for _ in range(20000)] + [rng.uniform(-1e-3, 1e-3) for _ in range(5000)]
    vals += [round(rng.uniform(-30, 30), 4) for _ in range(20000)]                    # what slot means look like
    vals += [rng.randrange(-400000, 400000) / 10000.0 / rng.randrange(1, 9) for _ in range(20000)]
    vals += [struct.unpack('<d', struct.pack('<Q', rng.getrandbits(64)))[0] for _ in range(20000)]   # any bit pattern
    # the formatter's decimal fast path (a mean that is fl(D / 10^8)): its edges, and the neighbours of such doubles (one ulp off: not D / 10^8)
    dec = [rng.randrange(1, 10 ** rng.randrange(1, 16)) / 1e8 for _ in range(20000)] + [1e-4, 0.00010001, 9999999.99999999, 1e7, 10000000.00000001,
                                                                                         99999.99999999, 0.5, 0.25, 0.125, 1 / 3, 2 / 3, 1e7 - 2 ** -29]
    vals += dec + [-x for x in dec[:2000]] + [float(np.nextafter(x, 0.0)) for x in dec[:5000]] + [float(np.nextafter(x, 1e9)) for x in dec[:5000]]
    for v in vals:
        if v != v:
            assert repr_double(v) == 'nan'
        else:
            assert repr_double(v) == repr(v), v
    for _ in range(5000):
        p = rng.random()
        assert repr_double(float(np.round(np.float64(p), 2))) == str(np.round(np.float64(p), 2))


def test_the_device_row_writers_digits_match_python():
    """mc_repr_double_rowtext -- mc_rowtext.h's shortest round-trip digits (free-format generation in 128-bit integer arithmetic,
    what the GPU's row writer prints slot means and read qualities with), built for the host -- == repr(float) over its whole range
    [1e-29, 1e9) and zero; everything else it refuses (the host formatter's case).  Half-way cases go to the even digit like dtoa.c."""
    import math
    import struct
    from mcaller_amd._lib import repr_double_rowtext as rt
    rng = random.Random(17)
    assert [rt(v) for v in (0.0, -0.0, 1.0, -2.5, 0.1 + 0.2, 1e-4, 9.999999999999999e-05, 5e-5, 1e-29, 999999999.9999999)] == \
        ['0.0', '-0.0', '1.0', '-2.5', '0.30000000000000004', '0.0001', '9.999999999999999e-05', '5e-05', '1e-29', '999999999.9999999']
    for v in (float('nan'), float('inf'), float('-inf'), 1e9, -1e9, 9.9e-30, 5e-324, 1e300):
        assert rt(v) is None
    vals = [-647985485.19140625, 2.9802322387695312e-08, 0.5, 0.25, 0.125, 1 / 3, 2 / 3, 7.055265349382997, 123456789.125, 5.551115123125783e-17,
            9.25185853854297e-18, 1.3552527156068805e-20]
    vals += [math.ldexp(1.0 + rng.random(), rng.randrange(-96, 29)) * rng.choice((-1, 1)) for _ in range(60000)]        # any mantissa, any exponent of the range
    vals += [rng.randrange(-2000000, 2000000) / (rng.randrange(1, 13) * 1e4) for _ in range(40000)]                      # slot means
    vals += [(rng.randrange(-2000000, 2000000) / 1e4) / rng.randrange(1, 13) for _ in range(40000)]
    vals += [round(rng.uniform(3, 40), rng.randrange(0, 16)) for _ in range(10000)]                                      # read qualities
    for _ in range(20000):                                                                                               # sums with a rounding residue
        n = rng.randrange(2, 10)
        xs = [rng.randrange(-100000, 100000) / 1e4 for _ in range(n)]
        vals.append((sum(xs) - sum(reversed(xs))) / n)
        vals.append(sum(xs) / n)
    for e in range(-96, 30):                                                                                            # powers of two (the lower gap is half the upper) ...
        p2 = math.ldexp(1.0, e)
        vals += [p2, float(np.nextafter(p2, 0.0)), float(np.nextafter(p2, 1e300))]
    for e in range(-29, 9):                                                                                             # ... and of ten, with their neighbours
        p10 = float('1e%d' % e)
        v = w = p10
        for _ in range(40):
            vals += [v, w]
            v, w = float(np.nextafter(v, 0.0)), float(np.nextafter(w, 1e300))
    vals += [(rng.getrandbits(rng.randrange(30, 54)) | 1) / math.ldexp(1.0, rng.randrange(1, 31)) for _ in range(40000)]    # short binary fractions: the half-way cases
    vals += [struct.unpack('<d', struct.pack('<Q', rng.getrandbits(64)))[0] for _ in range(20000)]                       # any bit pattern (mostly refused)
    refused = 0
    for v in vals:
        got = rt(v)
        if v == v and (v == 0.0 or 1e-29 <= abs(v) < 1e9):
            assert got == repr(v), v
        else:
            assert got is None, v
            refused += 1
    assert refused < len(vals) // 4

def test_division_by_1e4_without_dividing(tmp_path):
    """k1_emit's div1e4() (reciprocal, two fma) is the IEEE quotient for every int32: proof by exhaustion, in C."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'div1e4_check.c')
    exe = str(tmp_path / 'div1e4_check')
    subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-mfma', '-o', exe, src, '-lm', '-lpthread'])
    out = subprocess.check_output([exe], timeout=600).decode()
    assert 'mismatches over all int32: 0' in out, out


def test_fixed_point_repr_matches_python():
    """mc_repr_fixed4(d) (how the formatter prints the slot means that travel as integers: no floating-point conversion) ==
    repr(d / 1e4) == str(np.float64(d / 1e4)): the edges of the 32-bit range, every trailing-zero pattern, 200 000 random d."""
    from mcaller_amd._lib import repr_fixed4
    rng = random.Random(13)
    ds = [0, 1, -1, 9, 10, 99, 100, 999, 1000, 9999, 10000, -10000, 10001, 12300, 120000, 1234567, -7055, 2147483647, -2147483648,
          -2147483647, 2147480000, 100000000, 99999999, 50000, 5]
    ds += [rng.randrange(-400000, 400000) for _ in range(100000)]             # what (event - model) x 1e4 looks like
    ds += [rng.randrange(-2 ** 31, 2 ** 31) for _ in range(100000)]
    ds += [rng.randrange(-3000, 3000) * 100 for _ in range(5000)] + [rng.randrange(-300, 300) * 10000 for _ in range(2000)]
    for d in ds:
        assert repr_fixed4(d) == repr(d / 1e4), d
    for d in ds[:2000]:
        assert repr_fixed4(d) == str(np.float64(d) / 1e4)
