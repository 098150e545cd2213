"""Which summation order does torch's GPU `norm(dim=-1)` use for short rows?  Emulates candidates in numpy."""
import numpy as np, torch
torch.manual_seed(0)
F = lambda a, b, c: np.float32(np.float64(a) + np.float64(b) * np.float64(c))      # fused a + b*c

def cand(row, T, fma, combine_fma=False, vt=1):
    n = len(row)
    v = np.zeros(T, np.float32)
    started = np.zeros(T, bool)
    for t in range(T):
        idx = t
        while idx < n:
            x = row[idx]
            if fma: v[t] = F(v[t], x, x)
            else: v[t] = np.float32(v[t] + np.float32(x * x))
            idx += T
    off = T // 2
    while off >= 1:
        nv = v.copy()
        for t in range(T):
            if t + off < T: nv[t] = np.float32(v[t] + v[t + off])
        v = nv; off //= 2
    return np.sqrt(v[0])

def seq(row, fma):
    a = np.float32(0)
    for x in row: a = F(a, x, x) if fma else np.float32(a + np.float32(x * x))
    return np.sqrt(a)

for n in (33, 12, 3, 2):
    x = torch.randn(4000, n)
    g = torch.norm(x.cuda(), dim=1).cpu().numpy()
    xs = x.numpy()
    res = {}
    for T in (1, 2, 4, 8, 16, 32, 64):
        for fma in (0, 1):
            res["T%d%s" % (T, "f" if fma else "")] = sum(cand(xs[i], T, fma) != g[i] for i in range(600))
    res["seq"] = sum(seq(xs[i], 0) != g[i] for i in range(600)); res["seqf"] = sum(seq(xs[i], 1) != g[i] for i in range(600))
    print(n, {k: v for k, v in sorted(res.items(), key=lambda kv: kv[1])[:5]})
# transcendental agreement GPU torch vs (nothing to compare on host) -- just report a checksum for later comparison
a = torch.linspace(-3, 3, 100001)
for name in ("exp", "sin", "cos", "asin", "atan"):
    f = getattr(torch, name)
    arg = a.clamp(-1, 1) if name == "asin" else a
    d = (f(arg.cuda()).cpu() != f(arg)).float().mean().item()
    print("gpu-vs-cpu torch mismatch rate", name, "%.4f" % d)
# division by python scalar on GPU
b = torch.rand(100000)
print("gpu div scalar == x*(1/s):", torch.equal((b.cuda() / 0.0005).cpu(), (b * torch.tensor(1.0 / np.float32(0.0005))).float()), " == true div:", torch.equal((b.cuda() / 0.0005).cpu(), b / 0.0005))
