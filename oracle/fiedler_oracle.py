"""fiedler_oracle.py -- TEST INFRASTRUCTURE ONLY (tests/, never imported by the product).

CPU restatement, in numpy, of the deterministic Fiedler-ordering specification this repository publishes for the slot
`MatGetOrdering_Fiedler` (/root/reference/src/petsc_mat_fiedler.c:11-58).  The reference delegates that ordering to
HSL_MC73 (`mc73_order`, /root/reference/src/hslmc73f.F90:15-31), which is proprietary, not vendored and absent here, and
the reference holds no expected output for it (the only example, the 8 x 8 pattern of petsc_mat_fiedler.c:34-36, is dead
code without a result): against the reference itself this path is PARITY UNPINNED.  What this oracle pins is the
product's own published spec (header of spike-petsc_amd/csrc/host/fiedler.c, items 1-6), written a second time in a
different language and a different style (whole-vector numpy statements instead of C loops), so that
`spike_fiedler_order` can be compared BIT FOR BIT -- vector and permutation -- with something that is not itself.

Spec, restated:
 1. graph: edge {i, j}, i != j, if a_ij or a_ji is stored with |value| >= 1e-12 (the drop tolerance of the reference's
    Laplacian builder, /root/reference/src/spectralPartition.c:63-139); weighted mode iff a[0] > 0 (hslmc73f.F90:19):
    weight = sum of the |values| stored for the pair (in storage order), else 1; degree = row sum in ascending column order.
 2. connected components in order of their smallest vertex, each ordered on its own.
 3. multilevel: heavy-edge matching in index order (ties: smaller index) until <= 64 vertices (or the matching stalls:
    coarse > 9/10 fine), dense cyclic-Jacobi eigen-solve on the coarsest graph, piecewise-constant prolongation and a
    single-vector LOBPCG (Jacobi preconditioner, constant deflated) of at most 300 (1000 after a stall) iterations per
    level, stop at ||L x - rho x|| <= 1e-9 max deg.
 4. sign: the entry of largest magnitude (first among ties) positive.   5. order: descending value, ties by index.
 6. arithmetic: IEEE fp64, one operation at a time, every sum in ONE reduction order: chunks of 1024 indices, 256 slots
    ((v[t] + v[t+256]) + v[t+512]) + v[t+768], binary tree s[t] += s[t+o] (o = 128 .. 1), chunk sums added in order.

numpy never fuses a multiply with an add across two ufunc calls, and + - * / sqrt are correctly rounded, so a statement
written once per vector here and once per element in C produces the same bits.
"""
import math

import numpy as np

TOL = 1e-12


# ---- item 6: the one reduction order -----------------------------------------------------------------------------------
def rsum(v):
    """sum of the vector v in the spec's reduction order"""
    n = v.shape[0]
    nch = (n + 1023) // 1024
    pad = np.zeros(nch * 1024)
    pad[:n] = v
    q = pad.reshape(nch, 4, 256)
    s = ((q[:, 0, :] + q[:, 1, :]) + q[:, 2, :]) + q[:, 3, :]
    o = 128
    while o > 0:
        s = s[:, :o] + s[:, o:2 * o]
        o >>= 1
    total = 0.0
    for c in range(nch):
        total = total + float(s[c, 0])
    return total


def rdot(a, b=None):
    return rsum(a * 1.0 if b is None else a * b)


# ---- graphs --------------------------------------------------------------------------------------------------------------
class Graph:
    def __init__(self, n, xadj, adj, w, deg):
        self.n, self.xadj, self.adj, self.w, self.deg = n, xadj, adj, w, deg
        self._slots = None

    def lap(self, x):
        """y_i = deg_i x_i - w_k x_adj(k) ..., the neighbour terms subtracted one after the other in storage order"""
        if self._slots is None:
            cnt = np.diff(self.xadj)
            self._slots = []
            for j in range(int(cnt.max()) if self.n else 0):
                rows = np.nonzero(cnt > j)[0]
                self._slots.append((rows, self.xadj[rows] + j))
        y = self.deg * x
        for rows, pos in self._slots:
            y[rows] = y[rows] - self.w[pos] * x[self.adj[pos]]
        return y


def build_graph(n, ia, ja, a):
    weighted = ia[n] > 0 and a[0] > 0.0
    nb = [dict() for _ in range(n)]          # nb[i][j] = accumulated weight, contributions in storage order
    for i in range(n):
        for k in range(ia[i], ia[i + 1]):
            j = int(ja[k])
            if j < 0 or j >= n:
                raise ValueError("column out of range")
            v = abs(float(a[k]))
            if j == i or v < TOL:
                continue
            nb[i][j] = nb[i].get(j, 0.0) + v
            nb[j][i] = nb[j].get(i, 0.0) + v
    xadj = np.zeros(n + 1, dtype=np.int64)
    adj, w = [], []
    deg = np.zeros(n)
    for i in range(n):
        d = 0.0
        for j in sorted(nb[i]):
            wij = nb[i][j] if weighted else 1.0
            adj.append(j)
            w.append(wij)
            d = d + wij
        deg[i] = d
        xadj[i + 1] = len(adj)
    return Graph(n, xadj, np.array(adj, dtype=np.int64), np.array(w, dtype=np.float64), deg)


def coarsen(g):
    n = g.n
    cmap = [-1] * n
    nc = 0
    for i in range(n):
        if cmap[i] >= 0:
            continue
        best, bw = -1, -1.0
        for k in range(g.xadj[i], g.xadj[i + 1]):
            j = int(g.adj[k])
            if cmap[j] >= 0 or j == i:
                continue
            wk = float(g.w[k])
            if wk > bw or (wk == bw and j < best):
                bw, best = wk, j
        cmap[i] = nc
        if best >= 0:
            cmap[best] = nc
        nc += 1
    members = [[] for _ in range(nc)]
    for i in range(n):
        members[cmap[i]].append(i)
    xadj = np.zeros(nc + 1, dtype=np.int64)
    adj, w = [], []
    deg = np.zeros(nc)
    for ca in range(nc):
        where = {}                       # coarse neighbour -> slot, slots in order of first encounter
        start = len(adj)
        for i in members[ca]:
            for k in range(g.xadj[i], g.xadj[i + 1]):
                cb = cmap[int(g.adj[k])]
                if cb == ca:
                    continue
                if cb in where:
                    w[where[cb]] = w[where[cb]] + float(g.w[k])
                else:
                    where[cb] = len(adj)
                    adj.append(cb)
                    w.append(float(g.w[k]))
        d = 0.0
        for k in range(start, len(adj)):
            d = d + w[k]
        deg[ca] = d
        xadj[ca + 1] = len(adj)
    return Graph(nc, xadj, np.array(adj, dtype=np.int64), np.array(w, dtype=np.float64), deg), np.array(cmap, dtype=np.int64)


# ---- dense symmetric eigen-solver: cyclic Jacobi ----------------------------------------------------------------------------
def jacobi_eig(A):
    """A: (n, n) symmetric, destroyed.  Returns (eigenvalues, V with the eigenvectors in its columns)."""
    n = A.shape[0]
    V = np.eye(n)
    iu = np.triu_indices(n, 1)
    for _sweep in range(100):
        sq = A[iu] * A[iu]
        off = float(np.cumsum(sq)[-1]) if sq.size else 0.0      # strictly left to right, row-major upper triangle
        if off < 1e-30:
            break
        for p in range(n):
            for q in range(p + 1, n):
                apq = float(A[p, q])
                if abs(apq) < 1e-300:
                    continue
                theta = (float(A[q, q]) - float(A[p, p])) / (2.0 * apq)
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + math.sqrt(theta * theta + 1.0))
                c = 1.0 / math.sqrt(t * t + 1.0)
                s = t * c
                colp, colq = A[:, p].copy(), A[:, q].copy()
                A[:, p] = c * colp - s * colq
                A[:, q] = s * colp + c * colq
                rowp, rowq = A[p, :].copy(), A[q, :].copy()
                A[p, :] = c * rowp - s * rowq
                A[q, :] = s * rowp + c * rowq
                vp, vq = V[:, p].copy(), V[:, q].copy()
                V[:, p] = c * vp - s * vq
                V[:, q] = s * vp + c * vq
    return np.diag(A).copy(), V


def first_argmin(ev, skip=-1):
    b = -1
    for i in range(len(ev)):
        if i == skip:
            continue
        if b < 0 or ev[i] < ev[b]:
            b = i
    return b


# ---- LOBPCG refinement of one level -------------------------------------------------------------------------------------------
def refine(g, x, maxit):
    n = g.n
    nd = float(n)
    dmax = float(g.deg.max()) if n else 0.0
    if not (dmax > 0.0):
        dmax = 0.0

    def deflate(v):
        return v - rdot(v) / nd

    def normalize(v):
        s = math.sqrt(rdot(v, v))
        return (v / s if s > 0 else v), s

    x = deflate(x)
    x, s = normalize(x)
    if s == 0.0:
        x = np.where(np.arange(n) % 2 == 1, 1.0, -1.0)
        x = deflate(x)
        x, s = normalize(x)
    Lx = g.lap(x)
    rho = rdot(x, Lx)
    xn = 1.0
    havep = scale = False
    p = Lp = None
    degp = np.where(g.deg > 0, g.deg, 1.0)
    its = 0
    for _it in range(maxit):
        if scale:
            x = x / xn
            Lx = Lx / xn
        w = Lx - rho * x
        r2 = rdot(w, w)
        w = w / degp
        sw = rdot(w)
        its += 1
        scale = False
        if math.sqrt(r2) <= 1e-9 * dmax:
            break
        w = w - sw / nd
        a = rdot(w, x)
        b = rdot(p, x) if havep else 0.0
        w = w - a * x
        pn, a2 = 1.0, 0.0
        if havep:
            p = p - b * x
            Lp = Lp - b * Lx
            pp, wp = rdot(p, p), rdot(w, p)
            pn = math.sqrt(pp)
            if pn > 1e-300:
                a2 = wp / pn
            else:
                havep = False
        if havep:
            p = p / pn
            Lp = Lp / pn
            w = w - a2 * p
        wn = math.sqrt(rdot(w, w))
        if wn < 1e-300:
            break
        w = w / wn
        Lw = g.lap(w)
        m = 3 if havep else 2
        G = np.zeros((m, m))
        G[0, 0] = rdot(x, Lx)
        G[0, 1] = G[1, 0] = rdot(x, Lw)
        G[1, 1] = rdot(w, Lw)
        if havep:
            G[0, 2] = G[2, 0] = rdot(x, Lp)
            G[1, 2] = G[2, 1] = rdot(w, Lp)
            G[2, 2] = rdot(p, Lp)
        ev, V = jacobi_eig(G)
        bi = first_argmin(ev)
        c = [float(V[i, bi]) for i in range(m)]
        if c[0] < 0:
            c = [-q for q in c]
        rho = float(ev[bi])
        if havep:
            pnew = c[1] * w + c[2] * p
            Lpnew = c[1] * Lw + c[2] * Lp
        else:
            pnew = c[1] * w + 0.0
            Lpnew = c[1] * Lw + 0.0
        x = c[0] * x + pnew
        Lx = c[0] * Lx + Lpnew
        p, Lp = pnew, Lpnew
        xn = math.sqrt(rdot(x, x))
        havep = scale = True
    if scale:
        x = x / xn
    return x, its


def ramp(n):
    return np.arange(n, dtype=np.float64) - 0.5 * float(n - 1)


def fiedler_vector(g, level=0):
    n = g.n
    if n <= 64:
        A = np.zeros((n, n))
        for i in range(n):
            A[i, i] = g.deg[i]
            for k in range(g.xadj[i], g.xadj[i + 1]):
                A[i, g.adj[k]] = A[i, g.adj[k]] - g.w[k]
        ev, V = jacobi_eig(A)
        i0 = first_argmin(ev)
        i1 = first_argmin(ev, skip=i0)
        x0 = V[:, i1].copy() if i1 >= 0 else np.zeros(n)
        return refine(g, x0, 300)[0]
    if level >= 40:
        return refine(g, ramp(n), 300)[0]
    c, cmap = coarsen(g)
    if c.n > (9 * n) // 10:
        return refine(g, ramp(n), 1000)[0]
    xc = fiedler_vector(c, level + 1)
    return refine(g, xc[cmap], 300)[0]


def fiedler_order(n, ia, ja, a):
    """(order, vec): order[k] = old index placed at position k; vec = the per-component Fiedler vectors"""
    ia = np.asarray(ia, dtype=np.int64)
    ja = np.asarray(ja, dtype=np.int64)
    a = np.asarray(a, dtype=np.float64)
    g = build_graph(n, ia, ja, a)
    comp = [-1] * n
    order = []
    vec = np.zeros(n)
    ncomp = 0
    for s in range(n):
        if comp[s] >= 0:
            continue
        queue = [s]
        comp[s] = ncomp
        head = 0
        while head < len(queue):
            v = queue[head]
            head += 1
            for k in range(g.xadj[v], g.xadj[v + 1]):
                u = int(g.adj[k])
                if comp[u] < 0:
                    comp[u] = ncomp
                    queue.append(u)
        verts = sorted(queue)
        nc = len(verts)
        ncomp += 1
        if nc <= 2:
            for t, v in enumerate(verts):
                order.append(v)
                vec[v] = (0.7071067811865476 if t == 0 else -0.7071067811865476) if nc == 2 else 0.0
            continue
        loc = {v: t for t, v in enumerate(verts)}
        xadj = np.zeros(nc + 1, dtype=np.int64)
        adj, w = [], []
        for t, v in enumerate(verts):
            for k in range(g.xadj[v], g.xadj[v + 1]):
                adj.append(loc[int(g.adj[k])])
                w.append(float(g.w[k]))
            xadj[t + 1] = len(adj)
        sg = Graph(nc, xadj, np.array(adj, dtype=np.int64), np.array(w, dtype=np.float64), g.deg[verts].copy())
        x = fiedler_vector(sg, 0)
        im = 0
        for t in range(1, nc):
            if abs(x[t]) > abs(x[im]):
                im = t
        if x[im] < 0:
            x = -x
        for t, v in enumerate(verts):
            vec[v] = x[t]
        for t in sorted(range(nc), key=lambda t: (-float(x[t]) + 0.0, verts[t])):
            order.append(verts[t])
    return np.array(order, dtype=np.int64), vec
