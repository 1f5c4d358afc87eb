#!/usr/bin/env python3
"""tools/ubench/mfma_order_check.py DUMP [TRIALS] -- which arithmetic model reproduces v_mfma_f32_16x16x4_f32 bit for bit?

Reads the dump of tools/ubench/mfma_order.hip and evaluates candidate models in exact rational arithmetic with one
explicit round-to-nearest-even to binary32 wherever the model rounds:
  seq     acc = C; for k = 0..3: acc = round(acc + a_k b_k)        (a chain of fused multiply-adds, ascending k)
  rev     the same, descending k
  exact   round(C + sum_k a_k b_k)                                   (one rounding)
  pairs   round(round(C + a0 b0 + a1 b1) + a2 b2 + a3 b3)
  tree    round(C + round(round(a0 b0 + a1 b1) + round(a2 b2 + a3 b3)))
each with subnormal results kept ("ieee") or flushed to zero ("ftz", also subnormal inputs).  Prints the fraction of
outputs each model reproduces, per trial class.
"""
import struct
import sys
from fractions import Fraction

import numpy as np


def f32_round(x: Fraction, ftz: bool) -> float:
    """x rounded to the nearest binary32 (ties to even); returns a Python float holding that value exactly."""
    if x == 0:
        return 0.0
    s = -1 if x < 0 else 1
    ax = -x if x < 0 else x
    # exponent e with 2^e <= ax < 2^(e+1)
    e = ax.numerator.bit_length() - ax.denominator.bit_length()
    if Fraction(2) ** e > ax:
        e -= 1
    elif Fraction(2) ** (e + 1) <= ax:
        e += 1
    qe = max(e, -126) - 23
    q = ax / (Fraction(2) ** qe)
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    v = Fraction(n) * (Fraction(2) ** qe)
    if v >= Fraction(2) ** 128:
        return s * float("inf")
    if ftz and v < Fraction(2) ** -126:
        return 0.0
    return s * float(v)


def flush(v: float, ftz: bool) -> Fraction:
    if ftz and abs(v) < 2.0 ** -126:
        return Fraction(0)
    return Fraction(v)


def models(c, prods, ftz):
    out = {}
    acc = c
    for p in prods:
        acc = Fraction(f32_round(acc + p, ftz))
    out["seq"] = float(acc)
    acc = c
    for p in reversed(prods):
        acc = Fraction(f32_round(acc + p, ftz))
    out["rev"] = float(acc)
    out["exact"] = f32_round(c + sum(prods), ftz)
    if len(prods) == 4:
        out["pairs"] = f32_round(Fraction(f32_round(c + prods[0] + prods[1], ftz)) + prods[2] + prods[3], ftz)
        t0 = Fraction(f32_round(prods[0] + prods[1], ftz))
        t1 = Fraction(f32_round(prods[2] + prods[3], ftz))
        out["tree"] = f32_round(c + Fraction(f32_round(t0 + t1, ftz)), ftz)
    else:  # two chained instructions: the model applied per instruction
        h = len(prods) // 2
        out["pairs"] = f32_round(Fraction(f32_round(c + sum(prods[:h]), ftz)) + sum(prods[h:]), ftz)  # exact per instruction
    return out


def main():
    path = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    raw = open(path, "rb").read()
    (T,) = struct.unpack_from("<i", raw, 0)
    off = 4
    cls = np.frombuffer(raw, np.int32, T, off); off += 4 * T
    a = np.frombuffer(raw, np.float32, T * 64, off).reshape(T, 64); off += 4 * T * 64
    b = np.frombuffer(raw, np.float32, T * 64, off).reshape(T, 64); off += 4 * T * 64
    a2 = np.frombuffer(raw, np.float32, T * 64, off).reshape(T, 64); off += 4 * T * 64
    b2 = np.frombuffer(raw, np.float32, T * 64, off).reshape(T, 64); off += 4 * T * 64
    c = np.frombuffer(raw, np.float32, T * 256, off).reshape(T, 64, 4); off += 4 * T * 256
    d = np.frombuffer(raw, np.float32, T * 256, off).reshape(T, 64, 4)
    stats = {}
    for t in range(min(T, limit)):
        k_cls = int(cls[t])
        for l in range(0, 64, 3):
            for v in range(4):
                i, j = 4 * (l // 16) + v, l % 16
                got = float(d[t, l, v])
                if not np.isfinite(c[t, l, v]) or not np.isfinite(got):
                    continue
                for ftz in (False, True):
                    prods = [flush(float(a[t, i + 16 * k]), ftz) * flush(float(b[t, j + 16 * k]), ftz) for k in range(4)]
                    if k_cls == 3:
                        prods += [flush(float(a2[t, i + 16 * k]), ftz) * flush(float(b2[t, j + 16 * k]), ftz) for k in range(4)]
                    res = models(flush(float(c[t, l, v]), ftz), prods, ftz)
                    for name, val in res.items():
                        key = (k_cls, name, "ftz" if ftz else "ieee")
                        hit, n = stats.get(key, (0, 0))
                        same = np.float32(val).tobytes() == np.float32(got).tobytes() or (val == 0.0 and got == 0.0)
                        stats[key] = (hit + int(same), n + 1)
    for k_cls in range(4):
        print(f"class {k_cls}:")
        for key in sorted(stats):
            if key[0] == k_cls:
                hit, n = stats[key]
                print(f"   {key[1]:6s} {key[2]:5s} {hit:6d} / {n:6d}  = {hit / n:.4f}")


if __name__ == "__main__":
    main()
