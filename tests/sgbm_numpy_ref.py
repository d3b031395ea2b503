"""Second, independent restatement of the 3-way SGBM semantics in numpy (vectorised over d, Python loops over
x and y) -- used only to cross-check oracle/sgbm3way.c on tiny images, so that a slip in either restatement shows.
Written from SURVEY.md Appendix A + the QUIRK list in oracle/sgbm3way.c, not from the C code's structure:
costs are built as whole [H,W1,D] volumes and the stripes re-slice them."""
import numpy as np

SHRT_MAX = 32767


def _bt_interval(a):
    """a: int [W] row (already with border handling).  returns (a, lo, hi) half-pixel interval."""
    W = a.shape[0]
    l = a.copy(); r = a.copy()
    l[1:] = (a[1:] + a[:-1]) // 2
    r[:-1] = (a[:-1] + a[1:]) // 2
    return a, np.minimum(np.minimum(l, r), a), np.maximum(np.maximum(l, r), a)


def pixel_cost_volume(L, R, minD, D, cap):
    H, W = L.shape
    ft = max(cap, 15) | 1
    Li = L.astype(np.int64); Ri = R.astype(np.int64)
    maxD = minD + D
    minX1, maxX1 = max(maxD, 0), W + min(minD, 0)
    W1 = maxX1 - minX1
    pix = np.zeros((H, W1, D), np.int64)
    for y in range(H):
        ya, yb = max(y - 1, 0), min(y + 1, H - 1)
        chans = []
        for img in (Li, Ri):
            g = np.full(W, ft, np.int64)
            s = (img[y, 2:] - img[y, :-2]) * 2 + (img[ya, 2:] - img[ya, :-2]) + (img[yb, 2:] - img[yb, :-2])
            g[1:-1] = np.clip(s, -ft, ft) + ft
            raw = img[y].copy(); raw[0] = ft; raw[-1] = ft
            chans.append((g, raw))
        for c, shift in ((0, 0), (1, 2)):
            u, u0, u1 = _bt_interval(chans[0][c])
            v, v0, v1 = _bt_interval(chans[1][c])
            for xc in range(W1):
                x = xc + minX1
                xr = x - (minD + np.arange(D))
                c0 = np.maximum(0, np.maximum(u[x] - v1[xr], v0[xr] - u[x]))
                c1 = np.maximum(0, np.maximum(v[xr] - u1[x], u0[x] - v[xr]))
                pix[y, xc] += np.minimum(c0, c1) >> shift
    return pix


def hsum_volume(pix, bs):
    H, W1, D = pix.shape
    r = bs // 2
    idx = np.clip(np.arange(W1)[:, None] + np.arange(-r, r + 1)[None, :], 0, W1 - 1)
    return pix[:, idx, :].sum(axis=2)


def step(C, Lp, minp, P1, P2):
    pad = np.concatenate([[SHRT_MAX], Lp, [SHRT_MAX]])
    m = np.minimum(np.minimum(pad[:-2], pad[2:]) + P1, np.minimum(Lp, minp + P2))
    out = np.clip(C + m - (minp + P2), -32768, 32767)
    return out, out.min()


def compute(L, R, minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
            uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, return_raw=False):
    H, W = L.shape
    minD, D, bs = minDisparity, numDisparities, blockSize
    P1 = P1 if P1 > 0 else 2
    P2 = max(P2 if P2 > 0 else 5, P1 + 1)
    uniq = uniquenessRatio if uniquenessRatio >= 0 else 10
    d12 = disp12MaxDiff if disp12MaxDiff > 0 else 1
    maxD = minD + D
    minX1, maxX1 = max(maxD, 0), W + min(minD, 0)
    W1 = maxX1 - minX1
    INV = (minD - 1) * 16
    hs = hsum_volume(pixel_cost_volume(L, R, minD, D, preFilterCap), bs)
    r = bs // 2
    stripe = -(-H // 4)
    overlap = (bs // 2 + 1) + -(-stripe // 10)
    disp = np.full((H, W), INV, np.int64)
    for n in range(4):
        s0 = max(min(n * stripe - overlap, H), 0); s1 = min((n + 1) * stripe, H); o0 = min(n * stripe, H)
        top = np.zeros((W1, D), np.int64); topmin = np.zeros(W1, np.int64)
        for y in range(s0, s1):
            rows = np.clip(np.arange(y - r, y + r + 1), s0, H - 1)
            C = hs[rows].sum(axis=0)
            Ll = np.zeros((W1, D), np.int64)
            prev, pm = np.zeros(D, np.int64), 0
            for x in range(W1):
                prev, pm = step(C[x], prev, pm, P1, P2)
                Ll[x] = prev
                top[x], topmin[x] = step(C[x], top[x], topmin[x], P1, P2)
            d2 = np.full(W, INV, np.int64); d2c = np.full(W, SHRT_MAX, np.int64)
            prev, pm = np.zeros(D, np.int64), 0
            for x in range(W1 - 1, -1, -1):
                prev, pm = step(C[x], prev, pm, P1, P2)
                if y < o0:
                    continue
                S = np.clip(Ll[x] + prev + top[x], -32768, 32767)
                best = int(np.argmin(S)); mS = int(S[best])
                if uniq > 0:
                    bad = (S * (100 - uniq) < mS * 100) & (np.abs(np.arange(D) - best) > 1)
                    if bad.any():
                        continue
                x2 = x + minX1 - best - minD
                if d2c[x2] > mS:
                    d2c[x2] = mS; d2[x2] = best + minD
                if 0 < best < D - 1:
                    den = max(int(S[best - 1] + S[best + 1] - 2 * S[best]), 1)
                    num = int(S[best - 1] - S[best + 1]) * 16 + den
                    q = abs(num) // (2 * den)
                    dsp = best * 16 + (q if num >= 0 else -q)       # C division truncates toward zero
                else:
                    dsp = best * 16
                disp[y, x + minX1] = dsp + minD * 16
            if y < o0:
                continue
            for x in range(minX1, maxX1):
                d1 = int(disp[y, x])
                if d1 == INV:
                    continue
                _d = d1 >> 4; d_ = (d1 + 15) >> 4
                _x = x - _d; x_ = x - d_
                if (0 <= _x < W and d2[_x] >= minD and abs(d2[_x] - _d) > d12 and
                        0 <= x_ < W and d2[x_] >= minD and abs(d2[x_] - d_) > d12):
                    disp[y, x] = INV
    raw = disp.astype(np.int16)
    p = np.pad(raw, 1, mode="edge")
    stack = np.stack([p[i:i + H, j:j + W] for i in range(3) for j in range(3)], axis=0)
    med = np.sort(stack, axis=0)[4]
    if speckleWindowSize > 0:
        med = speckles(med, INV, speckleWindowSize, 16 * speckleRange)
    return (med, raw) if return_raw else med


def speckles(img, new_val, max_size, max_diff):
    """Connected components (4-neighbourhood, |diff| <= max_diff, neither == new_val) via union-find."""
    H, W = img.shape
    a = img.astype(np.int64)
    parent = np.arange(H * W)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    for y in range(H):
        for x in range(W):
            if a[y, x] == new_val:
                continue
            for yy, xx in ((y + 1, x), (y, x + 1)):
                if yy < H and xx < W and a[yy, xx] != new_val and abs(a[y, x] - a[yy, xx]) <= max_diff:
                    ra, rb = find(y * W + x), find(yy * W + xx)
                    if ra != rb:
                        parent[rb] = ra
    roots = np.array([find(i) for i in range(H * W)])
    counts = np.bincount(roots, minlength=H * W)
    out = img.copy().reshape(-1)
    kill = (counts[roots] <= max_size) & (a.reshape(-1) != new_val)
    out[kill] = new_val
    return out.reshape(H, W)
