"""Multi-rank restatement of the truncated-SPIKE apply on top of the oracle's single-partition primitives
(orc_band_lu / orc_band_lusolve) with a pluggable all-gather.  Test infrastructure: it mirrors, step for step,
what spike_engine.hip does for nranks > 1 (local passes, all-gather of [g_top(first) | g_bottom(last)], redundant
solve of the rank-boundary interface on both neighbours, corrected second pass)."""
import ctypes as C

import numpy as np

import oracle as O

dptr = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dptr)


def row_split(N, world):
    nblk = (N + 63) // 64
    return [(nblk * r // world) * 64 for r in range(world)] + [N]


class DistSpike:
    def __init__(self, N, row0, band_local, P_local, rank, world, allgather, boost_rel=1e-10):
        L = O.lib()
        self.L, self.N, self.row0, self.rank, self.world, self.allgather = L, N, row0, rank, world, allgather
        nd, n = band_local.shape
        self.K = K = (nd - 1) // 2
        self.n = n
        self.A = np.ascontiguousarray(band_local)
        self.LU = self.A.copy()
        self.starts = O.partition(n, P_local)
        self.P = P_local
        dmax = np.abs(self.A[K]).max()
        dmax = max(x[0] for x in allgather(np.array([dmax])))  # all-reduce(max), as the engine does
        for p in range(P_local):
            L.orc_band_lu(K, _p(self.LU), n, int(self.starts[p]), int(self.starts[p + 1]), boost_rel * dmax)
        # tips of every chain (zero where there is no neighbour)
        self.Wt = np.zeros((P_local, K, K))
        self.Vb = np.zeros((P_local, K, K))
        for p in range(P_local):
            s, e = int(self.starts[p]), int(self.starts[p + 1])
            for b in range(K):
                if row0 + s > 0:
                    col = np.zeros(n)
                    for a in range(b + 1):
                        col[s + a] = self.A[b - a, s + a]
                    self._solve(p, col)
                    self.Wt[p, :, b] = col[s:s + K]
                if row0 + e < N:
                    col = np.zeros(n)
                    for a in range(b, K):
                        col[e - K + a] = self.A[2 * K + b - a, e - K + a]
                    self._solve(p, col)
                    self.Vb[p, :, b] = col[e - K:e]
        ex = allgather(np.concatenate([self.Wt[0].ravel(), self.Vb[-1].ravel()]))
        self.V_prev = ex[rank - 1][K * K:].reshape(K, K) if rank > 0 else None
        self.W_next = ex[rank + 1][:K * K].reshape(K, K) if rank < world - 1 else None

    def _solve(self, p, v):
        self.L.orc_band_lusolve(self.K, _p(self.LU), self.n, int(self.starts[p]), int(self.starts[p + 1]), _p(v), _p(v))

    def _pass(self, f):
        x = f.copy()
        for p in range(self.P):
            self._solve(p, x)
        return x

    def _C(self, p):   # C_p(a,b) = A[s+a, s-K+b]
        K, s = self.K, int(self.starts[p])
        M = np.zeros((K, K))
        for a in range(K):
            for b in range(a, K):
                M[a, b] = self.A[b - a, s + a]
        return M

    def _B(self, p):   # B_p(a,b) = A[e-K+a, e+b]
        K, e = self.K, int(self.starts[p + 1])
        M = np.zeros((K, K))
        for a in range(K):
            for b in range(a + 1):
                M[a, b] = self.A[2 * K + b - a, e - K + a]
        return M

    def apply(self, f, variant=1):
        K = self.K
        g = self._pass(f)
        if variant == 0 or K == 0:
            return g
        st = self.starts
        ex = self.allgather(np.concatenate([g[:K], g[self.n - K:]]))
        f2 = f.copy()

        def iface(W, V, gb, gt):
            xt = np.linalg.solve(np.eye(K) - W @ V, gt - W @ gb)
            return xt, gb - V @ xt

        for i in range(self.P - 1):
            e = int(st[i + 1])
            xt, xb = iface(self.Wt[i + 1], self.Vb[i], g[e - K:e], g[e:e + K])
            f2[e:e + K] -= self._C(i + 1) @ xb
            f2[e - K:e] -= self._B(i) @ xt
        if self.rank > 0:
            xt, xb = iface(self.Wt[0], self.V_prev, ex[self.rank - 1][K:], g[:K])
            f2[:K] -= self._C(0) @ xb
        if self.rank < self.world - 1:
            xt, xb = iface(self.W_next, self.Vb[-1], g[self.n - K:], ex[self.rank + 1][:K])
            f2[self.n - K:] -= self._B(self.P - 1) @ xt
        return self._pass(f2)
