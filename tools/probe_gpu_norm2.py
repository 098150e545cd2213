import numpy as np, torch, itertools
torch.manual_seed(0)
F = lambda a, b, c: np.float32(np.float64(a) + np.float64(b) * np.float64(c))
def tree(v):
    v = list(v); T = len(v); off = T // 2
    while off >= 1:
        v = [np.float32(v[t] + v[t + off]) if t + off < len(v) else v[t] for t in range(len(v))]
        off //= 2
    return v[0]
def scheme(row, T, vt, contiguous, fma, acc4):
    n = len(row); vals = []
    for t in range(T):
        idxs = [t * vt + j for j in range(vt)] if contiguous else []
        if not contiguous:
            idxs = list(range(t, n, T))
        else:
            # grid-stride over chunks of T*vt
            idxs = []
            base = 0
            while base < n:
                idxs += [base + t * vt + j for j in range(vt)]
                base += T * vt
        idxs = [i for i in idxs if i < n]
        if acc4:
            accs = [np.float32(0)] * 4
            for j, i in enumerate(idxs):
                x = row[i]; a = accs[j % 4]
                accs[j % 4] = F(a, x, x) if fma else np.float32(a + np.float32(x * x))
            v = np.float32(np.float32(accs[0] + accs[1]) + np.float32(accs[2] + accs[3]))
        else:
            v = np.float32(0)
            for i in idxs:
                x = row[i]; v = F(v, x, x) if fma else np.float32(v + np.float32(x * x))
        vals.append(v)
    return np.sqrt(tree(vals))
for n in (33, 12):
    x = torch.randn(2000, n); g = torch.norm(x.cuda(), dim=1).cpu().numpy(); xs = x.numpy()
    res = {}
    for T, vt, cont, fma, a4 in itertools.product((1, 2, 4, 8, 16, 32, 64), (1, 2, 4), (0, 1), (0, 1), (0, 1)):
        if not cont and vt > 1: continue
        res[(T, vt, cont, fma, a4)] = sum(scheme(xs[i], T, vt, cont, fma, a4) != g[i] for i in range(300))
    print(n, sorted(res.items(), key=lambda kv: kv[1])[:6])
b = torch.rand(200000); g = (b.cuda() / 0.0005).cpu()
inv = np.float32(1.0) / np.float32(0.0005)
print("inv", repr(inv), "mismatch x*inv %.4f  true-div %.4f  x*2000 %.4f" % ((g != b * torch.tensor(inv)).float().mean(), (g != b / 0.0005).float().mean(), (g != b * 2000.0).float().mean()))
c = torch.rand(200000) + 0.5
gg = (b.cuda() / c.cuda()).cpu()
print("tensor/tensor gpu vs cpu mismatch %.5f" % (gg != b / c).float().mean())
print("sqrt gpu vs cpu mismatch %.5f" % (torch.sqrt(c.cuda()).cpu() != torch.sqrt(c)).float().mean())
print("fmod", (torch.remainder(b.cuda() * 100, 1.7995).cpu() != torch.remainder(b * 100, 1.7995)).float().mean().item())
