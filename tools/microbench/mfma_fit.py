#!/usr/bin/env python3
"""Which arithmetic does v_mfma_f32_32x32x16_{bf16,f16} implement?  Compares the dump of tools/microbench/mfma_dump.hip with a family
of candidate models, bit for bit, per test category (tools/microbench/mfma_model.py gen).

  mfma_fit.py <bf16|f16> in.bin out.bin [max_blocks]

Model family: the 16 products are exact; every addend (products and C) is aligned to the largest exponent among them and cut
to a multiple of 2^(Emax - 23 - G) (mode: toward zero / floor / nearest); the cut addends are summed exactly (optionally in
sequential groups of `gs` k's with an fp32 RNE rounding between groups) and the sum is rounded to fp32 (RNE / toward zero)."""
import sys, itertools
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__file__))
from mfma_model import load

KIND_NAMES = ["1 prod vs C=1.x", "1 prod vs C=+-1", "2 small", "4 small", "8 small", "16 small", "C=0 wide", "C=0 big+tiny", "general",
              "general sparse", "net relu", "net gauss", "pair distance", "big+C+tiny", "cancel", "tiny operands"]


def fexp(x):
    """floor(log2|x|) for nonzero float64, -10000 for 0"""
    m, e = np.frexp(x)
    return np.where(x == 0, -10000, e - 1)


def cut(x, q, mode):
    s = np.ldexp(x, -q)
    if mode == "zero":
        s = np.trunc(s)
    elif mode == "floor":
        s = np.floor(s)
    else:
        s = np.rint(s)
    return np.ldexp(s, q)


def to_f32(x, final):
    if final == "rne":
        return x.astype(np.float32)
    y = x.astype(np.float32)                      # toward zero: step back when the RNE result is larger in magnitude
    over = np.abs(y.astype(np.float64)) > np.abs(x)
    return np.where(over, np.nextafter(y, np.float32(0)), y)


def model(P, C, G, mode, gs, final, expo="norm", PE=None):
    """P [..,16] float64 exact products, C [..] float64; returns float32"""
    acc = C.copy()
    for g0 in range(0, 16, gs):
        Pg = P[..., g0:g0 + gs]
        eg = fexp(Pg) if PE is None else np.where(Pg == 0, -10000, PE[..., g0:g0 + gs])
        emax = np.maximum(eg.max(-1), fexp(acc))
        q = (emax - 23 - G)
        tot = cut(acc, q, mode) + cut(Pg, q[..., None], mode).sum(-1)
        acc = to_f32(tot, final).astype(np.float64) if g0 + gs < 16 or True else tot
    return acc.astype(np.float32)


def main():
    fmt, pin, pout = sys.argv[1:4]
    nmax = int(sys.argv[4]) if len(sys.argv) > 4 else 800
    A, B, C, D = load(fmt, pin, pout)
    n = min(nmax, A.shape[0])
    A, B, C, D = A[:n].astype(np.float64), B[:n].astype(np.float64), C[:n].astype(np.float64), D[:n]
    kinds = np.arange(n) % 16
    P = A[:, :, None, :] * np.transpose(B, (0, 2, 1))[:, None, :, :]          # [n, m, nn, k] exact
    # unnormalised product exponent ea + eb
    PE = fexp(A)[:, :, None, :] + fexp(np.transpose(B, (0, 2, 1)))[:, None, :, :]
    exact = np.array([[[float(np.float32(sum(map(float, P[b, i, j])) + C[b, i, j])) for j in range(32)] for i in range(32)] for b in range(0)])  # unused
    ref_bits = D.view(np.uint32)
    cands = []
    for G in (0, 1, 2, 3, 4, 5, 8, 24):
        for mode in ("zero", "floor", "near"):
            for gs in (16, 8, 4):
                for final in ("rne", "zero"):
                    for ex in ("norm", "unnorm"):
                        cands.append((G, mode, gs, final, ex))
    results = []
    for (G, mode, gs, final, ex) in cands:
        out = model(P, C, G, mode, gs, final, PE=PE if ex == "unnorm" else None)
        ok = (out.view(np.uint32) == ref_bits) | ((out == 0) & (D == 0))
        per = [float(ok[kinds == t].mean()) for t in range(16)]
        results.append((float(ok.mean()), (G, mode, gs, final, ex), per))
    results.sort(key=lambda r: -r[0])
    print(f"{fmt}: {n} blocks x 1024 cases; best models (match fraction overall | per category)")
    for tot, c, per in results[:14]:
        print(f"  G={c[0]:<2d} cut={c[1]:5s} group={c[2]:<2d} final={c[3]:4s} exp={c[4]:6s}  {tot:.5f} | " + " ".join(f"{p:.3f}" for p in per))
    print("categories: " + "; ".join(f"{i}={s}" for i, s in enumerate(KIND_NAMES)))


if __name__ == "__main__":
    main()
