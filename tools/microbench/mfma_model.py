#!/usr/bin/env python3
"""Fit a bit-exact model of v_mfma_f32_32x32x16_{bf16,f16}'s accumulation from raw input/output pairs.

  mfma_model.py gen <bf16|f16> in.bin          seeded case blocks for tools/microbench/mfma_dump.hip
  mfma_model.py fit <bf16|f16> in.bin out.bin  compare the dumped outputs with candidate models (exact integer arithmetic)

Diagnostic only (DESIGN.md "accumulation of the matrix pipe"); nothing in the product imports this."""
import sys
import numpy as np

NB = {"bf16": 3200, "f16": 3200}


def to_fmt(x, fmt):
    """float64 array -> (u16 bits, exact float64 value) by truncation of the mantissa (values chosen to be representable anyway)."""
    x32 = x.astype(np.float32)
    if fmt == "bf16":
        bits = (x32.view(np.uint32) >> 16).astype(np.uint16)
        val = (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    else:
        h = x32.astype(np.float16)
        bits = h.view(np.uint16)
        val = h.astype(np.float64)
    return bits, val


def rand_vals(rng, shape, e_lo, e_hi, fmt, mant_bits=None, p_zero=0.0):
    mb = (7 if fmt == "bf16" else 10) if mant_bits is None else mant_bits
    e = rng.integers(e_lo, e_hi + 1, size=shape)
    m = rng.integers(0, 1 << mb, size=shape) / float(1 << mb)
    s = rng.choice([-1.0, 1.0], size=shape)
    v = s * (1.0 + m) * np.exp2(e.astype(np.float64))
    if p_zero:
        v = np.where(rng.random(shape) < p_zero, 0.0, v)
    return v


def gen(fmt, path):
    rng = np.random.default_rng(20251004 + (fmt == "f16"))
    n = NB[fmt]
    A = np.zeros((n, 32, 16)); B = np.zeros((n, 16, 32)); Cm = np.zeros((n, 32, 32))
    kinds = np.zeros(n, np.int32)
    for b in range(n):
        t = b % 16
        kinds[b] = t
        if t == 0:      # one product against C = +-1.xxx: product exponents -18 .. -30 below C
            A[b, :, 0] = rand_vals(rng, 32, -16, -10, fmt); B[b, 0, :] = rand_vals(rng, 32, -14, -8, fmt)
            Cm[b] = rand_vals(rng, (32, 32), 0, 0, "bf16", mant_bits=23)
        elif t == 1:    # one product against C = +-1.0 exactly
            A[b, :, 0] = rand_vals(rng, 32, -16, -10, fmt); B[b, 0, :] = rand_vals(rng, 32, -14, -8, fmt)
            Cm[b] = rng.choice([-1.0, 1.0], size=(32, 32))
        elif t in (2, 3, 4, 5):   # nz small products (positions: first nz k / random k), C = +-1.xxx
            nz = (2, 4, 8, 16)[t - 2]
            ks = np.arange(nz) if b % 32 < 16 else rng.permutation(16)[:nz]
            A[b][:, ks] = rand_vals(rng, (32, nz), -15, -11, fmt); B[b][ks, :] = rand_vals(rng, (nz, 32), -14, -10, fmt)
            Cm[b] = rand_vals(rng, (32, 32), 0, 0, "bf16", mant_bits=23)
        elif t == 6:    # C = 0, all 16 products, wide exponent spread
            A[b] = rand_vals(rng, (32, 16), -6, 6, fmt); B[b] = rand_vals(rng, (16, 32), -6, 6, fmt)
        elif t == 7:    # C = 0, two products of very different size
            ks = rng.permutation(16)[:2]
            A[b][:, ks[0]] = rand_vals(rng, 32, 0, 2, fmt); B[b][ks[0], :] = rand_vals(rng, 32, 0, 2, fmt)
            A[b][:, ks[1]] = rand_vals(rng, 32, -14, -6, fmt); B[b][ks[1], :] = rand_vals(rng, 32, -14, -6, fmt)
        elif t in (8, 9):   # general: products around 2^0, C from 2^-4 to 2^12
            A[b] = rand_vals(rng, (32, 16), -3, 3, fmt, p_zero=0.2 * (t == 9)); B[b] = rand_vals(rng, (16, 32), -3, 3, fmt, p_zero=0.3 * (t == 9))
            Cm[b] = rand_vals(rng, (32, 32), -4, 12, "bf16", mant_bits=23)
        elif t in (10, 11):  # network-like: weights N(0, .06), activations relu(N(0,1)), C a partial sum
            A[b] = rng.normal(0, 0.06, (32, 16)); B[b] = np.maximum(rng.normal(0, 1.0, (16, 32)), 0.0) if t == 10 else rng.normal(0, 1.0, (16, 32))
            Cm[b] = rng.normal(0, 1.0, (32, 32)) * rng.choice([0.1, 1.0, 4.0])
        elif t == 12:   # pairs of tiny products at chosen k distances (do sub-granule parts combine within a group?)
            d = (1, 2, 4, 8)[(b // 16) % 4]
            A[b][:, [0, d]] = rand_vals(rng, (32, 2), -15, -12, fmt); B[b][[0, d], :] = rand_vals(rng, (2, 32), -14, -11, fmt)
            Cm[b] = rand_vals(rng, (32, 32), 0, 0, "bf16", mant_bits=23)
        elif t == 13:   # one big product + C + tiny ones: which exponent sets the alignment?
            A[b][:, 3] = rand_vals(rng, 32, 2, 4, fmt); B[b][3, :] = rand_vals(rng, 32, 2, 4, fmt)
            ks = [0, 5, 9, 14]
            A[b][:, ks] = rand_vals(rng, (32, 4), -12, -9, fmt); B[b][ks, :] = rand_vals(rng, (4, 32), -12, -9, fmt)
            Cm[b] = rand_vals(rng, (32, 32), -2, 2, "bf16", mant_bits=23)
        elif t == 14:   # cancelling big products, small C
            A[b] = rand_vals(rng, (32, 16), 0, 1, fmt); B[b] = rand_vals(rng, (16, 32), 0, 1, fmt)
            Cm[b] = rand_vals(rng, (32, 32), -20, -8, "bf16", mant_bits=23)
        else:           # f16 subnormal operands / bf16 tiny operands
            lo = -24 if fmt == "f16" else -60
            A[b] = rand_vals(rng, (32, 16), lo, lo + 12, fmt); B[b] = rand_vals(rng, (16, 32), 8, 12, fmt)
            Cm[b] = rand_vals(rng, (32, 32), -14, -2, "bf16", mant_bits=23)
    Ab, _ = to_fmt(A, fmt); Bb, _ = to_fmt(B, fmt)
    with open(path, "wb") as f:
        f.write(np.int32(n).tobytes())
        for b in range(n):
            f.write(Ab[b].tobytes()); f.write(Bb[b].tobytes()); f.write(Cm[b].astype(np.float32).tobytes())
    print(f"{fmt}: {n} blocks -> {path}")


def load(fmt, pin, pout):
    raw = np.fromfile(pin, np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    blk = raw[4:].reshape(n, 6144)
    Ab = blk[:, :1024].copy().view(np.uint16).reshape(n, 32, 16)
    Bb = blk[:, 1024:2048].copy().view(np.uint16).reshape(n, 16, 32)
    C = blk[:, 2048:].copy().view(np.float32).reshape(n, 32, 32)
    D = np.fromfile(pout, np.float32).reshape(n, 32, 32)
    if fmt == "bf16":
        Av = (Ab.astype(np.uint32) << 16).view(np.float32); Bv = (Bb.astype(np.uint32) << 16).view(np.float32)
    else:
        Av = Ab.view(np.float16).astype(np.float32); Bv = Bb.view(np.float16).astype(np.float32)
    return Av, Bv, C, D


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        gen(sys.argv[2], sys.argv[3])
    else:
        print("fit: see tools/microbench/mfma_fit.py")
