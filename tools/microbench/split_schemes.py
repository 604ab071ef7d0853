#!/usr/bin/env python3
"""Offline comparison of operand-splitting schemes for the fp32 chain kernels, on the bit-level model of the matrix pipe
(tools/microbench/mfma_sim.c, fitted to hardware dumps): the 8x256 (or 4x128) fixture network, forward + backward, every
scheme's activations / parameter gradients against an fp64 evaluation, next to the reference's own CPU fp32 results.

  python tools/microbench/split_schemes.py [8x256|4x128] [scheme ...]

Diagnostic only: reads tests/golden, runs on the CPU; nothing in the product imports it."""
import ctypes as C, os, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = C.CDLL(os.path.join(HERE, "bin", "libmfma_sim.so"))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sim_gemm(Ap, Bp, steps, nacc=1, f16=False):
    """Ap [npa, M, K], Bp [npb, N, K] float32 pieces; steps [(kstep, ia, ib, acc)] -> D [nacc, M, N]"""
    Ap, Bp = f32(Ap), f32(Bp)
    npa, M, K = Ap.shape; npb, N, K2 = Bp.shape
    assert K == K2
    st = np.ascontiguousarray(np.array(steps, np.int32).reshape(-1, 4))
    D = np.empty((nacc, M, N), np.float32)
    lib.mfma_gemm(M, N, K, Ap.ctypes.data_as(C.c_void_p), Bp.ctypes.data_as(C.c_void_p), len(st), st.ctypes.data_as(C.c_void_p), nacc,
                  D.ctypes.data_as(C.c_void_p), int(f16))
    return D


def fma_gemm(A, B):
    A, B = f32(A), f32(B)
    M, K = A.shape; N = B.shape[0]
    D = np.empty((M, N), np.float32)
    lib.fma_gemm(M, N, K, A.ctypes.data_as(C.c_void_p), B.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p))
    return D


# ------------------------------------------------------------------ splitting
def trunc_bf16(x):
    return (f32(x).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def rne_bf16(x):
    u = f32(x).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x, mode):
    x = f32(x)
    cut = trunc_bf16 if mode == "trunc" else rne_bf16
    p1 = cut(x); r = x - p1; p2 = cut(r); p3 = r - p2
    assert np.all(cut(p3) == p3)
    return np.stack([p1, p2, p3])


def split_h(x, n):
    """n fp16 pieces by round-to-nearest (values must be in range)"""
    x = f32(x); out = []; r = x.copy()
    for _ in range(n):
        p = r.astype(np.float16).astype(np.float32); out.append(p); r = r - p
    return np.stack(out)


def pad16(a):
    K = a.shape[-1]; Kp = (K + 15) // 16 * 16
    if Kp == K:
        return a
    out = np.zeros(a.shape[:-1] + (Kp,), a.dtype); out[..., :K] = a
    return out


# ------------------------------------------------------------------ schemes: (A pieces, B pieces, steps, nacc, f16, post-scale)
TERMS_SMALL_FIRST = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]        # (ia, ib): a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1


def steps_for(order, nk, terms):
    s = []
    if order == "kmajor":                       # the kernels' order: per k-step all terms
        for k in range(nk):
            s += [(k, ia, ib, 0) for ia, ib in terms]
    elif order == "corr_first":                 # all corrections of all k-steps, then the leading products
        for k in range(nk):
            s += [(k, ia, ib, 0) for ia, ib in terms[:-1]]
        s += [(k, terms[-1][0], terms[-1][1], 0) for k in range(nk)]
    elif order == "split_acc":                  # corrections in a second accumulator
        for k in range(nk):
            s += [(k, ia, ib, 1) for ia, ib in terms[:-1]] + [(k, terms[-1][0], terms[-1][1], 0)]
    elif order == "lead_first":                 # the leading products of all k-steps, then the corrections (small first) of all k-steps
        s += [(k, terms[-1][0], terms[-1][1], 0) for k in range(nk)]
        for k in range(nk):
            s += [(k, ia, ib, 0) for ia, ib in terms[:-1]]
    elif order == "term_major":                 # one term at a time over all k, smallest term first
        for ia, ib in terms:
            s += [(k, ia, ib, 0) for k in range(nk)]
    else:
        raise ValueError(order)
    return s


class Scheme:
    def __init__(self, name, fmt="bf16", split="trunc", order="kmajor", terms=None, na=3, nb=3):
        self.name, self.fmt, self.split, self.order = name, fmt, split, order
        self.na, self.nb = na, nb
        self.terms = terms or TERMS_SMALL_FIRST

    def gemm(self, A, B):
        """A [M, K] (weights side), B [N, K] -> A . B^T [M, N] as the scheme computes it (fp32)"""
        A, B = pad16(f32(A)), pad16(f32(B))
        nk = A.shape[1] // 16
        if self.fmt == "fma":
            return fma_gemm(A, B)
        if self.fmt == "bf16":
            Ap, Bp = split3(A, self.split)[: self.na], split3(B, self.split)[: self.nb]
            D = sim_gemm(Ap, Bp, steps_for(self.order, nk, self.terms), 2 if self.order == "split_acc" else 1)
            return D[0] + D[1] if self.order == "split_acc" else D[0]
        # fp16 pieces with power-of-two scaling: A by one scale (max -> 2^12), B per row (per sample; max -> 2^12)
        TA, TB = int(os.environ.get("H2_TA", 12)), int(os.environ.get("H2_TB", 12))
        sa = TA - np.floor(np.log2(np.abs(A).max()))
        mb = np.abs(B).max(axis=1, keepdims=True); mb = np.where(mb == 0, 1.0, mb)
        sb = TB - np.floor(np.log2(mb))
        Ap = split_h(np.ldexp(A, int(sa)), self.na); Bp = split_h(f32(B * np.exp2(sb)), self.nb)
        D = sim_gemm(Ap, Bp, steps_for(self.order, nk, self.terms), 2 if self.order == "split_acc" else 1, f16=True)
        D = D[0] + D[1] if self.order == "split_acc" else D[0]
        return f32(D * np.exp2(-(sa + sb.T)))


H2 = [(1, 0), (0, 1), (0, 0)]                    # fp16 two-piece: a2 b1, a1 b2, a1 b1
H2W3 = [(2, 0), (1, 0), (0, 1), (0, 0)]          # weights in three pieces (exact), activations in two
SCHEMES = {
    "fma": Scheme("fp32 fma chain", fmt="fma"),
    "x3": Scheme("bf16x3 trunc split, k-major small-first (round-2 kernels)"),
    "x3_rne": Scheme("bf16x3 RNE split, k-major small-first", split="rne"),
    "x3_corr_first": Scheme("bf16x3 trunc, corrections of all k first", order="corr_first"),
    "x3_rne_corr_first": Scheme("bf16x3 RNE, corrections of all k first", split="rne", order="corr_first"),
    "x3_split_acc": Scheme("bf16x3 trunc, split accumulators", order="split_acc"),
    "x3_rne_split_acc": Scheme("bf16x3 RNE, split accumulators", split="rne", order="split_acc"),
    "x3_rne_term_major": Scheme("bf16x3 RNE, term-major", split="rne", order="term_major"),
    "x3_lead_first": Scheme("bf16x3 trunc, leading products of all k first", order="lead_first"),
    "x3_rne_lead_first": Scheme("bf16x3 RNE, leading products of all k first", split="rne", order="lead_first"),
    "h2w3_split_acc": Scheme("fp16: weights x3, activations x2, split acc", fmt="f16", terms=H2W3, na=3, nb=2, order="split_acc"),
    "h2": Scheme("fp16x2 RNE scaled, 3 products k-major", fmt="f16", terms=H2, na=2, nb=2),
    "h2_corr_first": Scheme("fp16x2 RNE scaled, corrections first", fmt="f16", terms=H2, na=2, nb=2, order="corr_first"),
    "h2_split_acc": Scheme("fp16x2 RNE scaled, split acc", fmt="f16", terms=H2, na=2, nb=2, order="split_acc"),
    "h2w3": Scheme("fp16: weights x3, activations x2, 4 products k-major", fmt="f16", terms=H2W3, na=3, nb=2),
    "h2w3_corr_first": Scheme("fp16: weights x3, activations x2, corrections first", fmt="f16", terms=H2W3, na=3, nb=2, order="corr_first"),
}


# ------------------------------------------------------------------ the network
def load(tag):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{tag}.npz"))
    L, hidden, depth, skip_at = (int(v) for v in g["cfg"])
    params = [g[f"p{i:02d}"] for i in range(2 * depth + 4)]
    m = np.load(os.path.join(ROOT, "tests", "golden", f"mlp_{tag}.npz"))
    return dict(depth=depth, skip_at=skip_at, hidden=hidden), params, m


def forward(cfg, P, x, gemm, dt):
    depth, skip = cfg["depth"], cfg["skip_at"]
    h = x.astype(dt); acts, ins = [], []
    for l in range(depth):
        ins.append(h)
        z = gemm(P[2 * l], h).T.astype(dt) + P[2 * l + 1].astype(dt)          # [M, hidden]
        h = np.maximum(z, 0); acts.append(h)
        if skip > 0 and l == skip - 1:
            h = np.concatenate([h, x.astype(dt)], -1)
    Wh = np.concatenate([P[2 * depth + 2], P[2 * depth]], 0)                 # rows r,g,b,sigma
    bh = np.concatenate([P[2 * depth + 3], P[2 * depth + 1]], 0)
    zo = gemm(Wh, h).T.astype(dt) + bh.astype(dt)
    rgb = 1 / (1 + np.exp(-zo[:, :3])); sig = np.maximum(zo[:, 3:], 0)
    return acts, ins, h, rgb.astype(dt), sig.astype(dt)


def backward(cfg, P, acts, ins, hlast, rgb, sig, g_rgb, g_sig, gemm, wgemm, dt):
    depth, skip, hid = cfg["depth"], cfg["skip_at"], cfg["hidden"]
    dzh = np.concatenate([g_rgb.astype(dt) * (rgb * (1 - rgb)), np.where(sig > 0, g_sig.astype(dt), 0)], -1).astype(dt)   # [M, 4]
    Wh = np.concatenate([P[2 * depth + 2], P[2 * depth]], 0)
    grads = [None] * (2 * depth + 4)
    gh = wgemm(dzh.T, hlast.T)                                              # [4, K]
    grads[2 * depth + 2], grads[2 * depth] = gh[:3], gh[3:]
    bsum = dzh.sum(0, dtype=dt); grads[2 * depth + 3], grads[2 * depth + 1] = bsum[:3], bsum[3:]
    dh = gemm(Wh.T.copy(), dzh).T.astype(dt)                                # [M, hidden(+in)]
    dzs = [None] * depth
    for l in range(depth - 1, -1, -1):
        dz = np.where(acts[l] > 0, dh[:, :hid], 0).astype(dt); dzs[l] = dz
        grads[2 * l] = wgemm(dz.T, ins[l].T); grads[2 * l + 1] = dz.sum(0, dtype=dt)
        if l > 0:
            Wl = P[2 * l][:, :hid]                                          # gradient w.r.t. the hidden part only (the input carries none)
            dh = gemm(Wl.T.copy(), dz).T.astype(dt)
    return grads, dzs


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1][0].isdigit() else "8x256"
    names = [a for a in sys.argv[1:] if a in SCHEMES] or ["fma", "x3"]
    cfg, P32, m = load(tag)
    x = m["x"]; g_rgb, g_sig = m["g_rgb"], m["g_sigma"]
    P64 = [p.astype(np.float64) for p in P32]
    g64mm = lambda A, B: A.astype(np.float64) @ B.astype(np.float64).T
    a64, i64, hl64, rgb64, sig64 = forward(cfg, P64, x, g64mm, np.float64)
    G64, dz64 = backward(cfg, P64, a64, i64, hl64, rgb64, sig64, g_rgb, g_sig, g64mm, g64mm, np.float64)
    nt = len(P32)
    cpu = [m[f"g{i:02d}"].astype(np.float64) for i in range(nt)]
    relmax = lambda g, r: float(np.abs(g - r).max() / np.abs(r).max())
    cpu_err = [relmax(cpu[i], G64[i]) for i in range(nt)]
    print(f"{tag}: reference CPU fp32 per-tensor error vs fp64 (max|d| / max|g|), weights: " + " ".join(f"{cpu_err[i]:.1e}" for i in range(0, nt, 2)))
    wg = SCHEMES["x3"]                                                      # weight-gradient GEMMs: as the kernels (chunks of samples -> slabs)
    for name in names:
        sc = SCHEMES[name]
        t0 = time.time()
        wsc = SCHEMES[os.environ["WGRAD_SCHEME"]] if "WGRAD_SCHEME" in os.environ else (SCHEMES["x3"] if name != "fma" else sc)

        def wgemm(A, B, wsc=wsc):                                           # A [N, M samples], B [K, M]: chunks of 256 samples, slabs summed in fp32
            M = A.shape[1]; acc = None
            for c0 in range(0, M, 256):
                part = wsc.gemm(A[:, c0:c0 + 256], B[:, c0:c0 + 256])
                acc = part if acc is None else f32(acc + part)
            return acc
        acts, ins, hl, rgb, sig = forward(cfg, P32, x, sc.gemm, np.float32)
        G, dzs = backward(cfg, P32, acts, ins, hl, rgb, sig, g_rgb, g_sig, sc.gemm, wgemm, np.float32)
        ea = [float(np.linalg.norm(acts[l] - a64[l]) / np.linalg.norm(a64[l])) for l in range(cfg["depth"])]
        ez = [float(np.linalg.norm(dzs[l] - dz64[l]) / np.linalg.norm(dz64[l])) for l in range(cfg["depth"])]
        bias = [float(np.mean((acts[l] - a64[l])[a64[l] > 0] / a64[l][a64[l] > 0])) for l in range(cfg["depth"])]
        eg = [relmax(G[i].astype(np.float64), G64[i]) for i in range(nt)]
        gall = np.concatenate([g.reshape(-1) for g in G]).astype(np.float64); g64all = np.concatenate([g.reshape(-1) for g in G64])
        print(f"\n[{name}] {sc.name}   ({time.time() - t0:.0f} s)")
        print("  H[l] rel L2 error      : " + " ".join(f"{e:.1e}" for e in ea))
        print("  H[l] mean relative bias: " + " ".join(f"{e:+.1e}" for e in bias))
        print("  dZ[l] rel L2 error     : " + " ".join(f"{e:.1e}" for e in ez))
        print("  weight grads, per tensor max|d|/max|g|: " + " ".join(f"{eg[i]:.1e}" for i in range(0, nt, 2)))
        print("     ... ratio to the CPU fp32 reference's: " + " ".join(f"{eg[i] / cpu_err[i]:.1f}" for i in range(0, nt, 2)))
        print(f"  bias grads worst ratio {max(eg[i] / max(cpu_err[i], 1e-12) for i in range(1, nt, 2)):.1f};  whole-vector rel L2 {np.linalg.norm(gall - g64all) / np.linalg.norm(g64all):.2e}"
              f"  (CPU: {np.linalg.norm(np.concatenate([c.reshape(-1) for c in cpu]) - g64all) / np.linalg.norm(g64all):.2e})", flush=True)


if __name__ == "__main__":
    main()
