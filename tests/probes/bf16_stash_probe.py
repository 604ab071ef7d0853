"""Decode the bf16 training stash after fwd + dgrad and compare each record with the CPU restatement."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")]
import torch
from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O
import nerf
from tnerf import ops, lib as L

dev = torch.device("cuda:0")
tag = os.environ.get("TAG", "4x128")
cfg, params = golden_params(tag)
g = torch.Generator().manual_seed(3)
params = O.mlp_init(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], g)
params[2 * cfg["depth"] + 1] = params[2 * cfg["depth"] + 1] + 0.5
H, depth, skip_at, Lf = cfg["hidden"], cfg["depth"], cfg["skip_at"], cfg["L"]
m = nerf.TinyNeRF(cfg["in_dim"], H, depth, skip_at).to(dev)
with torch.no_grad():
    for p, v in zip(m.parameters(), params):
        p.copy_(v.to(dev))
st = m._ensure_packed()
gg = load_golden(f"render_{tag}")
ro, rd = O.pinhole_rays(int(gg["H"]), int(gg["W"]), float(gg["focal"]), gg["pose"])
R, S = 16, 64
idx = torch.arange(0, ro.shape[0], ro.shape[0] // R)[:R]
o, d = ro[idx].contiguous(), rd[idx].contiguous()
tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(1))
t = torch.rand(R, S, generator=torch.Generator().manual_seed(2))

# ---- CPU restatement, keeping the intermediates
z, pts = O.stratified(2.0, 6.0, S, o, d, t)
xb = O.bf16_round(O.posenc(pts.reshape(-1, 3), Lf, True))
Wb = [O.bf16_round(params[2 * i]) for i in range(depth)]
w_s, b_s, w_c, b_c = params[2 * depth: 2 * depth + 4]
Wh = O.bf16_round(torch.cat([w_c, w_s], 0)); bh = torch.cat([b_c, b_s], 0)
mm = lambda a, b: (a.double() @ b.double()).float()
ins, hs = [], []
inp = xb
for i in range(depth):
    ins.append(inp)
    h = O.bf16_round(torch.relu(mm(inp, Wb[i].t()) + params[2 * i + 1])); hs.append(h)
    inp = torch.cat([h, xb], -1) if i == skip_at - 1 else h
zh = mm(hs[-1], Wh.t()) + bh
rgb = torch.sigmoid(zh[:, :3]).requires_grad_(True); sigma = torch.relu(zh[:, 3:4]).requires_grad_(True)
comp = O.composite(rgb.reshape(R, S, 3), sigma.reshape(R, S, 1), z, d, True)[0]
loss = ((comp - tgt) ** 2).sum() / (3 * R)
gcomp = (2.0 * (comp - tgt) / (3 * R)).detach()
d_rgb, d_sigma = torch.autograd.grad(loss, [rgb, sigma])
rgb, sigma = rgb.detach(), sigma.detach()
dzh = O.bf16_round(torch.cat([d_rgb * (rgb * (1 - rgb)), d_sigma * (sigma > 0).float()], -1))
dzs = [None] * depth
dz = O.bf16_round(mm(dzh, Wh) * (hs[-1] > 0).float()); dzs[depth - 1] = dz
for i in range(depth - 1, 0, -1):
    dz = O.bf16_round(mm(dz, Wb[i][:, :H]) * (hs[i - 1] > 0).float()); dzs[i - 1] = dz

# ---- GPU
b = st.repack_bf16(); bp = b.train_plan(R, S)
bp.stash.zero_()
ztab = ops.depth_table(2.0, 6.0, S, dev)
compd = torch.empty(R, 3, device=dev)
od, dd, td, gcd = o.to(dev), d.to(dev), t.to(dev), gcomp.to(dev)
s_ = torch.cuda.current_stream(dev).cuda_stream
L.call("tnerf_train_fwd_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), od.data_ptr(), dd.data_ptr(), R, S, ztab.data_ptr(), 1, td.data_ptr(), 0, 0, 1, compd.data_ptr(), bp.stash.data_ptr(), s_)
L.call("tnerf_train_dgrad_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), od.data_ptr(), dd.data_ptr(), R, S, ztab.data_ptr(), 1, td.data_ptr(), 0, 0, 1, gcd.data_ptr(), bp.stash.data_ptr(), s_)
torch.cuda.synchronize()
print("comp err", float((compd.cpu() - comp.detach()).abs().max()))
raw = bp.stash.cpu()
NT = H // 32
n_ft = 2 + 2 * NT * depth + 1
ntiles = bp.n_tiles
frag = raw[:(ntiles + 1) * n_ft * 2048].view(torch.bfloat16).float().reshape(ntiles + 1, n_ft, 2, 64, 8)


def acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def decode(ft0, ntile_f):
    """-> [M = ntiles*32, ntile_f*32] matrix (sample slot, feature)."""
    out = torch.zeros(ntiles * 32, ntile_f * 32)
    for tf in range(ntile_f):
        rec = frag[:ntiles, ft0 + tf]                      # [tile, u, lane, e]
        for u in range(2):
            for lane in range(64):
                c, hh = lane & 31, lane >> 5
                for e in range(8):
                    out[torch.arange(ntiles) * 32 + acc_row(8 * u + e, hh), tf * 32 + c] = rec[:, u, lane, e]
    return out


ft_enc, ft_h, ft_dz, ft_dzh = 0, [2 + NT * l for l in range(depth)], [2 + NT * depth + NT * l for l in range(depth)], 2 + 2 * NT * depth
got = decode(ft_h[0], NT)
print("H_0 nnz got/want", int((got != 0).sum()), int((hs[0] != 0).sum()), "sum", float(got.sum()), float(hs[0].sum()))
print("tile0 sorted match:", torch.equal(torch.sort(got[:32].reshape(-1))[0], torch.sort(hs[0][:32].reshape(-1))[0]))
print("got[0,:8]", got[0, :8].tolist()); print("want[0,:8]", hs[0][0, :8].tolist())
print("got[:8,0]", got[:8, 0].tolist()); print("want[:8,0]", hs[0][:8, 0].tolist())
# where does want[0, 0..3] appear in tile 0's record?
rec0 = frag[0, ft_h[0]]
for f in range(4):
    v = float(hs[0][0, f])
    loc = (rec0 == v).nonzero()
    print(f"want[sample 0, feature {f}] = {v}: found at (u, lane, e) = {loc[:4].tolist()}")
for smp in (1, 5):
    v = float(hs[0][smp, 2])
    print(f"want[sample {smp}, feature 2] = {v}: found at {(rec0 == v).nonzero()[:4].tolist()}")
for l in range(depth):
    got = decode(ft_h[l], NT)
    print(f"H_{l}: max err {float((got - hs[l]).abs().max()):.3e} (max |want| {float(hs[l].abs().max()):.3e})")
got = decode(ft_dzh, 1)[:, :4]
print(f"dZh: max err {float((got - dzh).abs().max()):.3e} (max |want| {float(dzh.abs().max()):.3e}); rest of tile max {float(decode(ft_dzh, 1)[:, 4:].abs().max()):.3e}")
for l in range(depth - 1, -1, -1):
    got = decode(ft_dz[l], NT)
    print(f"dZ_{l}: max err {float((got - dzs[l]).abs().max()):.3e} (max |want| {float(dzs[l].abs().max()):.3e})")
# enc tiles: slot -> column
enc = decode(ft_enc, 2)
cols = {}
for u in range(4):
    for hh in range(2):
        for e in range(8):
            a = 8 * u + e
            col = 3 + 6 * (a // 3) + (a % 3) + 3 * hh if a < 3 * Lf else (hh if a == 3 * Lf else ((2 if hh == 0 else -1) if a == 3 * Lf + 1 else -1))
            if col >= 0:
                cols[col] = 32 * (u >> 1) + acc_row(8 * (u & 1) + e, hh)
err = max(float((enc[:, cols[c]] - xb[:, c]).abs().max()) for c in range(cfg["in_dim"]))
print(f"ENC: max err {err:.3e}")
# wgrad from the decoded stash vs slabs/reduce
L.call("tnerf_wgrad_bf16", C.byref(st.desc), bp.stash.data_ptr(), bp.n_tiles, bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), s_)
L.call("tnerf_wgrad_reduce", bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.n_params, st.grad.data_ptr(), s_)
torch.cuda.synchronize()
flat = st.grad.cpu()
want = [None] * len(params)
for i in range(depth):
    want[2 * i] = mm(dzs[i].t(), ins[i]); want[2 * i + 1] = dzs[i].double().sum(0).float()
gWh = mm(dzh.t(), hs[-1]); gbh = dzh.double().sum(0).float()
want[2 * depth] = gWh[3:4]; want[2 * depth + 1] = gbh[3:4]; want[2 * depth + 2] = gWh[:3]; want[2 * depth + 3] = gbh[:3]
off = 0
for i, x in enumerate(want):
    n = x.numel(); gx = flat[off:off + n].reshape(x.shape); off += n
    print(f"grad p{i:02d} {tuple(x.shape)}: rel {float((gx - x).norm() / (x.norm() + 1e-20)):.2e}")
