"""Diagnostic: the training stash written by the x3 forward against the fp32-MFMA forward's, region by region."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden_params
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
for tag in ("4x128", "8x256"):
    cfg, params = golden_params(tag)
    m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
    st = m._ensure_packed(); x3 = st.repack_x3(1)
    R, S = 64, 32
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev); o = (-4.0 * d + 0.1).contiguous(); u = torch.rand(R, S, generator=g).to(dev)
    plan = st.plan(R * S); ztab = ops.depth_table(2.0, 6.0, S, dev)
    sA = torch.zeros_like(plan.stash); sB = torch.zeros_like(plan.stash); cA = torch.empty(R, 3, device=dev); cB = torch.empty(R, 3, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    common = (o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
    lib.call("tnerf_train_fwd_fused", C.byref(st.desc), st.packed.data_ptr(), *common, cA.data_ptr(), sA.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_train_fwd_fused_x3", C.byref(st.desc), x3.packed.data_ptr(), *common, cB.data_ptr(), sB.data_ptr(), plan.Mp, sp)
    torch.cuda.synchronize()
    H, NE, depth = (128 if cfg["hidden"] <= 128 else 256), (32 if cfg["L"] == 10 else 20), cfg["depth"]
    rows = 2 * NE + depth * H + 4 + depth * H + 4
    M = R * S; Mp = plan.Mp
    A = sA[: (Mp // 32 + 1) * rows * 32].view(-1, rows, 32).cpu(); B = sB[: (Mp // 32 + 1) * rows * 32].view(-1, rows, 32).cpu()
    nb = M // 32
    print(tag, "comp diff", float((cA - cB).abs().max()))
    print("  enc rows   ", float((A[:nb, :2 * NE] - B[:nb, :2 * NE]).abs().max()))
    print("   per row:", [round(float((A[:nb, r] - B[:nb, r]).abs().max()), 3) for r in range(2 * NE)])
    print("   A row0..5 sample0:", A[0, :6, 0].tolist(), " B:", B[0, :6, 0].tolist())
    r0 = 2 * NE
    for l in range(depth):
        print(f"  H[{l}]       ", float((A[:nb, r0:r0 + H] - B[:nb, r0:r0 + H]).abs().max()), "max", float(A[:nb, r0:r0 + H].abs().max()))
        r0 += H
    print("  out rows   ", float((A[:nb, r0:r0 + 4] - B[:nb, r0:r0 + 4]).abs().max()))
    body = rows * (Mp + 32)
    mA = sA[body:].view(torch.int32).cpu(); mB = sB[body:].view(torch.int32).cpu()
    NT = H // 32
    for l in range(depth):
        a = mA[l * (Mp + 32) * NT: l * (Mp + 32) * NT + M * NT]; b = mB[l * (Mp + 32) * NT: l * (Mp + 32) * NT + M * NT]
        x = (a ^ b)
        print(f"  mask[{l}] words differing {int((x != 0).sum())} of {a.numel()}, bits {int(sum(bin(int(v) & 0xffffffff).count('1') for v in x[x != 0][:2000]))}")
    # ---- dgrad: both kernels on the fp32 forward's stash (same inputs), dZ rows compared
    gc = torch.randn(R, 3, generator=g).to(dev) / (3 * R)
    sC = sA.clone()
    lib.call("tnerf_train_dgrad_fused", C.byref(st.desc), st.packed.data_ptr(), *common, gc.data_ptr(), sA.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_train_dgrad_fused_x3", C.byref(st.desc), x3.packed.data_ptr(), *common, gc.data_ptr(), sC.data_ptr(), plan.Mp, sp)
    torch.cuda.synchronize()
    A = sA[: (Mp // 32 + 1) * rows * 32].view(-1, rows, 32).cpu(); B = sC[: (Mp // 32 + 1) * rows * 32].view(-1, rows, 32).cpu()
    r0 = 2 * NE + depth * H + 4
    for l in range(depth):
        da = A[:nb, r0:r0 + H].double(); db = B[:nb, r0:r0 + H].double()
        print(f"  dZ[{l}]      max diff {float((da - db).abs().max()):.3e}  rel L2 {float((da - db).norm() / da.norm()):.3e}  max {float(da.abs().max()):.3e}")
        r0 += H
    print("  dzh rows   ", float((A[:nb, r0:r0 + 4] - B[:nb, r0:r0 + 4]).abs().max()), "max", float(A[:nb, r0:r0 + 4].abs().max()))
