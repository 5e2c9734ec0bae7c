"""Word-level CPU emulation of `tw_lds_kernel` (csrc/kernels_twlds.hip): merge_mask_list (reference utils/textmask.py:73-132)
for ONE window on bit planes, exactly as the kernel computes it -- 32-pixel words, run ids from a prefix count of run starts,
word-local 8-neighbourhoods with one halo bit either side, union-find over run ids, the reference's `w * h < 3` bounding-box
rule as a bit pattern (components of one pixel, or two pixels side by side / on top of each other), per-root sums.

Every formula here is the kernel's (same names); tests/test_twlds_emul.py checks it against the oracle's pixel-level
`merge_mask_list`.  Plain Python over words: small windows only."""
import numpy as np

M32 = 0xFFFFFFFF


def popc(v):
    return bin(v & M32).count("1")


def ctz(v):
    return (v & -v).bit_length() - 1


def to_plane(mask_bool):
    """(H, W) bool -> (H, wp) list of ints, bit x & 31 of word x >> 5; bits beyond W are zero (the kernel's invariant)."""
    H, W = mask_bool.shape
    wp = (W + 31) >> 5
    out = [[0] * wp for _ in range(H)]
    ys, xs = np.nonzero(mask_bool)
    for y, x in zip(ys.tolist(), xs.tolist()):
        out[y][x >> 5] |= 1 << (x & 31)
    return out


def from_plane(p, W):
    H = len(p)
    out = np.zeros((H, W), bool)
    for y in range(H):
        for x in range(W):
            out[y, x] = (p[y][x >> 5] >> (x & 31)) & 1
    return out


class Win:
    def __init__(self, H, W):
        self.H, self.W, self.wp = H, W, (W + 31) >> 5
        self.last = M32 if W % 32 == 0 else (1 << (W % 32)) - 1         # valid bits of a row's last word

    def valid(self, wi):
        return self.last if wi == self.wp - 1 else M32

    def get(self, p, y, wi, fill=0):
        """word wi of row y; `fill` outside the window"""
        if y < 0 or y >= self.H or wi < 0 or wi >= self.wp:
            return fill
        return p[y][wi]

    # L(v)[x] = v[x - 1], R(v)[x] = v[x + 1] inside one row (zeros outside the window unless `ones`)
    def L(self, p, y, wi, ones=False):
        cur, prev = self.get(p, y, wi, M32 if ones else 0), self.get(p, y, wi - 1, M32 if ones else 0)
        return ((cur << 1) | (prev >> 31)) & M32

    def R(self, p, y, wi, ones=False):
        cur, nxt = self.get(p, y, wi, M32 if ones else 0), self.get(p, y, wi + 1, M32 if ones else 0)
        return ((cur >> 1) | ((nxt & 1) << 31)) & M32

    def sh(self, p, y, wi, k):
        """bits shifted by k in {-2,-1,0,1,2}: result[x] = p[y][x + k], zeros outside the window"""
        cur, prev, nxt = self.get(p, y, wi), self.get(p, y, wi - 1), self.get(p, y, wi + 1)
        if k == 0:
            return cur
        if k > 0:
            return ((cur >> k) | (nxt << (32 - k))) & M32
        return ((cur << -k) | (prev >> (32 + k))) & M32


def pred_plane(win, mask_u8):
    """pred_bin of merge_mask_list (:85-89): 3x3 CROSS erosion (pixels outside the window ignored), > 60."""
    b = to_plane(mask_u8 > 60)
    out = [[0] * win.wp for _ in range(win.H)]
    for y in range(win.H):
        for wi in range(win.wp):
            # out-of-window neighbours are ones: columns beyond W inside the last word too
            def ext(yy, ww):
                if yy < 0 or yy >= win.H or ww < 0 or ww >= win.wp:
                    return M32
                return b[yy][ww] | (~win.valid(ww) & M32)
            cur, prev, nxt = ext(y, wi), ext(y, wi - 1), ext(y, wi + 1)
            l = ((cur << 1) | (prev >> 31)) & M32
            r = ((cur >> 1) | ((nxt & 1) << 31)) & M32
            out[y][wi] = cur & l & r & ext(y - 1, wi) & ext(y + 1, wi) & win.valid(wi)
    return out


def tiny_filter(win, c):
    """cand minus its components with bounding box w * h < 3: 1x1, 2x1, 1x2 (reference :97 `if w * h < 3: continue`)."""
    out = [[0] * win.wp for _ in range(win.H)]
    for y in range(win.H):
        for wi in range(win.wp):
            S = lambda dy, k: win.sh(c, y + dy, wi, k)          # noqa: E731
            cur = S(0, 0)
            ring = lambda dy: S(dy, -1) | S(dy, 0) | S(dy, 1)   # noqa: E731  the three pixels of row y+dy around x
            up, dn = ring(-1), ring(1)
            single = cur & ~(S(0, -1) | S(0, 1) | up | dn)
            # horizontal pair, this pixel the LEFT one: x+1 set, x-1 / x+2 clear, rows above / below clear over x-1 .. x+2
            hl = cur & S(0, 1) & ~(S(0, -1) | S(0, 2) | up | dn | S(-1, 2) | S(1, 2))
            # ... the RIGHT one: x-1 set, x-2 / x+1 clear, rows above / below clear over x-2 .. x+1
            hr = cur & S(0, -1) & ~(S(0, -2) | S(0, 1) | up | dn | S(-1, -2) | S(1, -2))
            # vertical pair, this pixel the TOP one: below set alone, row y-1 and row y+2 clear around x
            vt = cur & S(1, 0) & ~(S(0, -1) | S(0, 1) | S(1, -1) | S(1, 1) | up | ring(2))
            vb = cur & S(-1, 0) & ~(S(0, -1) | S(0, 1) | S(-1, -1) | S(-1, 1) | dn | ring(-2))
            out[y][wi] = cur & ~(single | hl | hr | vt | vb) & M32
    return out


class Labels:
    """The kernel's labelling of one plane: run starts, prefix counts, union-find over run ids."""

    def __init__(self, win, c):
        self.win, self.c = win, c
        H, wp = win.H, win.wp
        self.s = [[c[y][wi] & ~win.L(c, y, wi) & M32 for wi in range(wp)] for y in range(H)]
        self.base, n = [[0] * wp for _ in range(H)], 0
        for y in range(H):
            for wi in range(wp):
                self.base[y][wi] = n
                n += popc(self.s[y][wi])
        self.n = n
        self.parent = list(range(n))

    def rid(self, y, wi, p):
        """id of the run covering the set pixel (bit p of word wi, row y): starts at or before it, minus one"""
        low = M32 if p == 31 else ((2 << p) - 1)
        return self.base[y][wi] + popc(self.s[y][wi] & low) - 1

    def find(self, a):
        while self.parent[a] != a:
            self.parent[a] = self.parent[self.parent[a]]
            a = self.parent[a]
        return a

    def union(self, a, b):
        a, b = self.find(a), self.find(b)
        if a != b:
            self.parent[max(a, b)] = min(a, b)

    @staticmethod
    def groups(v):
        """maximal groups of set bits of a word, lowest first: (lowest bit, mask)"""
        out = []
        while v:
            lb = v & -v
            grp = v & ~(v + lb) & M32
            out.append((ctz(v), grp))
            v &= ~grp
        return out

    def link_rows(self):
        """8-connectivity: every word-local group against the row above, one halo bit either side"""
        win, c = self.win, self.c
        for y in range(1, win.H):
            for wi in range(win.wp):
                a, al, ar = win.get(c, y - 1, wi), win.get(c, y - 1, wi - 1), win.get(c, y - 1, wi + 1)
                for p, grp in self.groups(c[y][wi]):
                    me = self.rid(y, wi, p)
                    m = (grp | (grp << 1) | (grp >> 1)) & M32
                    an = a & m
                    reps = an & ~(an << 1) & M32
                    while reps:
                        q = ctz(reps)
                        reps &= reps - 1
                        self.union(me, self.rid(y - 1, wi, q))
                    if (grp & 1) and (al >> 31):
                        self.union(me, self.base[y - 1][wi] - 1)
                    if (grp >> 31) and (ar & 1):
                        self.union(me, self.rid(y - 1, wi + 1, 0))

    def roots(self):
        return [self.find(i) for i in range(self.n)]


def accept_round(win, cand_raw, pred, merged):
    """one candidate (reference :92-108): components that lower the xor distance to pred are OR-ed into merged"""
    c = tiny_filter(win, cand_raw)
    lab = Labels(win, c)
    acc = [0] * lab.n
    lab.link_rows()
    for y in range(win.H):
        for wi in range(win.wp):
            for p, grp in lab.groups(c[y][wi]):
                nm = grp & ~merged[y][wi]
                acc[lab.rid(y, wi, p)] += popc(nm & pred[y][wi]) - popc(nm & ~pred[y][wi])
    root = lab.roots()
    tot = [0] * lab.n
    for i in range(lab.n):
        tot[root[i]] += acc[i]
    for y in range(win.H):
        for wi in range(win.wp):
            add = 0
            for p, grp in lab.groups(c[y][wi]):
                if tot[root[lab.rid(y, wi, p)]] > 0:
                    add |= grp
            merged[y][wi] |= add
    return lab.n


def dilate(win, m):
    out = [[0] * win.wp for _ in range(win.H)]
    for y in range(win.H):
        for wi in range(win.wp):
            v = 0
            for dy in (-1, 0, 1):
                v |= win.get(m, y + dy, wi) | win.L(m, y + dy, wi) | win.R(m, y + dy, wi)
            out[y][wi] = v & win.valid(wi)
    return out


def fill_holes(win, pred, merged):
    """reference :113-131: components of the complement smaller than the second largest entry of
    sorted({set pixels} U {component areas}) may be OR-ed in"""
    comp = [[~merged[y][wi] & win.valid(wi) & M32 for wi in range(win.wp)] for y in range(win.H)]
    a0 = sum(popc(merged[y][wi]) for y in range(win.H) for wi in range(win.wp))
    lab = Labels(win, comp)
    lab.link_rows()
    area = [0] * lab.n
    for y in range(win.H):
        for wi in range(win.wp):
            for p, grp in lab.groups(comp[y][wi]):
                area[lab.rid(y, wi, p)] += popc(grp)
    root = lab.roots()
    tot = [0] * lab.n
    for i in range(lab.n):
        tot[root[i]] += area[i]
    # tp = [maximum, multiplicity - 1, runner-up] over {a0} U {areas of the roots}
    entries = [a0] + [tot[i] for i in range(lab.n) if root[i] == i]
    mx = max(entries)
    mult = sum(1 for e in entries if e == mx)
    rest = [e for e in entries if e != mx]
    thr = mx if mult >= 2 else (max(rest) if rest else -1)
    if thr < 0:
        return lab.n
    acc = [0 if tot[i] < thr else -(1 << 30) for i in range(lab.n)]
    for y in range(win.H):
        for wi in range(win.wp):
            for p, grp in lab.groups(comp[y][wi]):
                acc[root[lab.rid(y, wi, p)]] += popc(grp & pred[y][wi]) - popc(grp & ~pred[y][wi])
    for y in range(win.H):
        for wi in range(win.wp):
            add = 0
            for p, grp in lab.groups(comp[y][wi]):
                if acc[root[lab.rid(y, wi, p)]] > 0:
                    add |= grp
            merged[y][wi] |= add
    return lab.n


def merge_mask_list(cands, pred_mask_u8, refine_mode=0):
    """cands: list of (H, W) uint8 {0, 255} candidate masks IN MERGE ORDER; returns the merged mask as (H, W) uint8."""
    H, W = pred_mask_u8.shape
    win = Win(H, W)
    pred = pred_plane(win, pred_mask_u8)
    merged = [[0] * win.wp for _ in range(H)]
    runs = 0
    for cm in cands:
        runs = max(runs, accept_round(win, to_plane(cm > 0), pred, merged))
    if refine_mode == 0:
        merged = dilate(win, merged)
    runs = max(runs, fill_holes(win, pred, merged))
    return (from_plane(merged, W) * 255).astype(np.uint8), runs
