"""Which order does torch-ROCm's reduce kernel sum a contiguous fp32 row of 33 / 12 / 3 / 2 squares in (torch.norm(x, dim=1))?
Hypothesis (the structure of ATen's Reduce.cuh for a reduction over the fastest dimension with fewer than 128 inputs per
output): block.x = T = the largest power of two <= n (at most 32 here) threads share a row; thread t takes x[t], x[t + T], ...
into up to four accumulators (acc_k = fma(x, x, 0) for the k-th element of the thread), adds the accumulators 0 + 1 + 2 + 3 in
that order, and the row's threads combine by shuffle-down with offsets 1, 2, 4, ... (ASCENDING); the result is sqrt().
Prints the fraction of rows each variant reproduces bit for bit.  GPU box: python tools/probe_gpu_norm3.py [rows]"""
import sys
import numpy as np
import torch


def last_pow2(n):
    p = 1
    while p * 2 <= n:
        p *= 2
    return p


def emulate(x, fma, ascending, T=None, zero_first=True):
    """x [R, n] float32 numpy -> sum of squares [R] float32 in the hypothesised order."""
    R, n = x.shape
    T = T or min(last_pow2(n), 32)
    x64 = x.astype(np.float64)
    sq = (x64 * x64)                                            # exact in f64
    part = np.zeros((R, T), np.float32)
    for t in range(T):
        idx = list(range(t, n, T))
        accs = []
        for k, i in enumerate(idx):
            if fma:
                accs.append(sq[:, i].astype(np.float32))           # fma(x, x, 0) = round(x^2)
            else:
                accs.append((x[:, i] * x[:, i]).astype(np.float32))
        assert len(accs) <= 4
        v = accs[0] if accs else np.zeros(R, np.float32)
        for a in accs[1:]:
            v = (v + a).astype(np.float32)
        part[:, t] = v
    if ascending:
        off = 1
        while off < T:
            nxt = part.copy()
            for t in range(T):
                if t + off < T:
                    nxt[:, t] = (part[:, t] + part[:, t + off]).astype(np.float32)
            part = nxt
            off *= 2
    else:
        off = T // 2
        while off >= 1:
            nxt = part.copy()
            for t in range(T):
                if t + off < T:
                    nxt[:, t] = (part[:, t] + part[:, t + off]).astype(np.float32)
            part = nxt
            off //= 2
    return part[:, 0]


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    torch.manual_seed(0)
    for n in (33, 12, 3, 2, 6, 13):
        for shift in (0, 1, 3):                                  # base pointer alignment: the tensor starts `shift` floats into a buffer
            big = torch.randn(R * n + 8)
            x = big[shift:shift + R * n].view(R, n)
            xg = big.cuda()[shift:shift + R * n].view(R, n)
            assert xg.is_contiguous()
            g = torch.norm(xg, dim=1).cpu().numpy()
            g2 = torch.sqrt((xg * xg).sum(dim=1)).cpu().numpy()     # the sum kernel: same reduce structure, no fma
            xs = x.numpy()
            out = []
            for fma in (1, 0):
                for asc in (1, 0):
                    s = emulate(xs, fma, asc)
                    r_np = np.sqrt(s)
                    r_gpu = torch.sqrt(torch.from_numpy(s).cuda()).cpu().numpy()
                    out.append(("fma=%d asc=%d" % (fma, asc), float((r_np == g).mean()), float((r_gpu == g).mean()), float((r_gpu == g2).mean())))
            print("n=%d shift=%d rows=%d" % (n, shift, R))
            for name, a, b, c in out:
                print("   %s: norm == np.sqrt(emul) %.6f   norm == gpu_sqrt(emul) %.6f   sqrt(sum(x*x)) == gpu_sqrt(emul) %.6f" % (name, a, b, c))
    # sqrt itself: torch-ROCm against correctly rounded
    c = torch.rand(1 << 20) * 100 + 1e-3
    print("gpu sqrt == correctly rounded sqrt: %.6f" % float((torch.sqrt(c.cuda()).cpu().numpy() == np.sqrt(c.numpy())).mean()))
    # num_outputs dependence: the same rows inside a smaller tensor
    big = torch.randn(1 << 16, 33)
    ref = torch.norm(big.cuda(), dim=1).cpu()
    for m in (1, 3, 8, 100, 4096, 16384):
        sub = torch.norm(big[:m].cuda(), dim=1).cpu()
        print("rows %d of the same data equal the 65536-row result: %s" % (m, bool(torch.equal(sub, ref[:m]))))


if __name__ == "__main__":
    main()
