"""mc64_oracle.py -- CPU ORACLE for the weighted-matching ordering (MC64 "job 5").  TEST INFRASTRUCTURE ONLY.

Only tests/ may import this file.  The product's matching is spike-petsc_amd/csrc/host/mc64.c (C); this is a second,
separately written restatement of the same reference routine, in plain Python loops (small cases: n up to a few
thousand), so that the product's permutation / scalings can be compared BIT FOR BIT on inputs whose optimum is not
unique -- the only place where an assignment solver's answer depends on the traversal order.

What is restated (paths relative to /root/reference, read as text; nothing is imported, compiled or copied from it):

  job-5 cost construction     src/hslmc64.c:703-743  (c = log(colmax) - log|a|; exact zero -> RINF/n, RINF/n set at :407-408)
  matching                    src/hslmc64.c:1917-2380  HSLmc64WD
    dual initialisation       :1973-1986   row minimum u, LAST minimum in column-scan order wins ("a > u : skip")
    cheap assignment          :1987-2009   rows in order; dense-column rule n/10 when n > 50
    second pass               :2014-2099   reduced-cost minimum with the unmatched-row preference, one-step augment
    main loop                 :2109-2329   Dijkstra from every unmatched column, two-part queue in one array
    duals / completion        :2331-2352, HSLmc64XD :2555-2607
  heap (iway = 2, min-heap)   mc64DD :3962, mc64ED :4044, mc64FD :4140
  post-scaling                src/hslmc64.c:822-832  (v_j -= log(colmax_j) when the matching is perfect)
  caller convention           src/petsc_mat_wbm.c:29-33,52-58 (CSR arrays of A handed to the CSC interface; 1-based)

Pinning: the reference holds ONE known answer for this path, the 3x3 matrix of src/wbm.c:485-497 with
perm = [3,1,2] and the scalings recorded in SURVEY.md section 8c (tests/golden/mc64_wbm_3x3.json); this oracle reproduces
it bit for bit (tests/test_mc64_oracle.py).  The reference's own C file cannot be compiled here without writing stand-ins
for PETSc headers, so beyond that vector tie-breaking parity with the reference is "parity unpinned"; what the tests pin
is oracle == product (bit-exact) and oracle == optimal (scipy.optimize.linear_sum_assignment objective).
"""
import math
import sys

RINF = sys.float_info.max


class _Queue:
    """The reference's q/l/d triple: q[1..qlen] is a binary min-heap on d, q[low..up-1] the rows at distance dmin,
    q[up..n] the rows already scanned; l[i] = position of row i in q (0 = not queued)."""

    def __init__(self, n):
        self.q = [0] * (n + 2)
        self.l = [0] * (n + 2)
        self.d = [0.0] * (n + 2)
        self.qlen = 0

    def sift_up(self, i):                                   # mc64DD, iway = 2  (hslmc64.c:3962-4020)
        q, l, d = self.q, self.l, self.d
        pos = l[i]
        if pos > 1:
            di = d[i]
            while True:
                parent = pos // 2
                qk = q[parent]
                if di >= d[qk]:
                    break
                q[pos] = qk
                l[qk] = pos
                pos = parent
                if pos <= 1:
                    break
        q[pos] = i
        l[i] = pos

    def _sift_down_from(self, pos, i, di):
        q, l, d = self.q, self.l, self.d
        while True:
            child = 2 * pos
            if child > self.qlen:
                break
            dk = d[q[child]]
            if child < self.qlen:
                dr = d[q[child + 1]]
                if dk > dr:
                    child += 1
                    dk = dr
            if di <= dk:
                break
            q[pos] = q[child]
            l[q[pos]] = pos
            pos = child
        q[pos] = i
        l[i] = pos

    def pop_root(self):                                     # mc64ED, iway = 2  (hslmc64.c:4044-4120)
        i = self.q[self.qlen]
        di = self.d[i]
        self.qlen -= 1
        self._sift_down_from(1, i, di)

    def delete_at(self, pos0):                              # mc64FD, iway = 2  (hslmc64.c:4140-4260)
        q, l, d = self.q, self.l, self.d
        if self.qlen == pos0:
            self.qlen -= 1
            return
        i = q[self.qlen]
        di = d[i]
        self.qlen -= 1
        pos = pos0
        if pos > 1:
            while True:
                parent = pos // 2
                qk = q[parent]
                if di >= d[qk]:
                    break
                q[pos] = qk
                l[qk] = pos
                pos = parent
                if pos <= 1:
                    break
        q[pos] = i
        l[i] = pos
        if pos != pos0:
            return
        self._sift_down_from(pos, i, di)


def _match(n, ip, irn, a):
    """HSLmc64WD on a square n x n pattern; ip/irn/a are 1-based lists (index 0 unused)."""
    Q = _Queue(n)
    q, l, d = Q.q, Q.l, Q.d
    u = [RINF] * (n + 1)
    iperm = [0] * (n + 1)
    jperm = [0] * (n + 1)
    out = [0] * (n + 1)
    pr = [0] * (n + 1)
    for k in range(1, n + 1):
        d[k] = 0.0
        pr[k] = ip[k]
    num = 0
    # ---- :1973-1986  u_i = min over the row; the test is "a > u: skip", so equal values overwrite (last one wins)
    for j in range(1, n + 1):
        for k in range(ip[j], ip[j + 1]):
            i = irn[k]
            if not (a[k] > u[i]):
                u[i] = a[k]
                iperm[i] = j
                l[i] = k
    # ---- :1987-2009  cheap assignment
    for i in range(1, n + 1):
        j = iperm[i]
        if j == 0:
            continue
        iperm[i] = 0
        if jperm[j] != 0:
            continue
        if ip[j + 1] - ip[j] > n // 10 and n > 50:
            continue
        num += 1
        iperm[i] = j
        jperm[j] = l[i]
    if num != n:
        # ---- :2014-2099  second pass over the columns still unassigned
        for j in range(1, n + 1):
            if jperm[j] != 0:
                continue
            k1, k2 = ip[j], ip[j + 1] - 1
            if k1 > k2:
                continue
            i0 = irn[k1]
            vj = a[k1] - u[i0]
            k0 = k1
            for k in range(k1 + 1, k2 + 1):
                i = irn[k]
                di = a[k] - u[i]
                if di > vj:
                    continue
                if di < vj or di == RINF:
                    take = True
                else:                                       # tie: only an unmatched row may displace a matched one
                    take = not (iperm[i] != 0 or iperm[i0] == 0)
                if take:
                    vj, i0, k0 = di, i, k
            d[j] = vj
            k, i = k0, i0
            if iperm[i] != 0:
                hit = None
                k = k0
                while k <= k2:
                    i = irn[k]
                    if not (a[k] - u[i] > vj):
                        jj = iperm[i]
                        # jj == 0 needs an unmatched row at the minimum while i0 is matched: only reachable through
                        # the "di == RINF" branch above, where the reference indexes pr[0] (undefined); skipped
                        kk1, kk2 = (pr[jj], ip[jj + 1] - 1) if jj > 0 else (1, 0)
                        if kk1 <= kk2:
                            for kk in range(kk1, kk2 + 1):
                                ii = irn[kk]
                                if iperm[ii] > 0:
                                    continue
                                if a[kk] - u[ii] <= d[jj]:
                                    hit = (jj, kk, ii)
                                    break
                            if hit is not None:
                                break
                            pr[jj] = kk2 + 1
                    k += 1
                if hit is None:
                    continue
                jj, kk, ii = hit
                jperm[jj] = kk
                iperm[ii] = jj
                pr[jj] = kk + 1
            num += 1
            jperm[j] = k
            iperm[i] = j
            pr[j] = k + 1
    if num != n:
        for i in range(1, n + 1):
            d[i] = RINF
            l[i] = 0
        isp = jsp = 0
        # ---- :2109-2329  main loop
        for jord in range(1, n + 1):
            if jperm[jord] != 0:
                continue
            dmin = RINF
            Q.qlen = 0
            low = up = n + 1
            csp = RINF
            j = jord
            pr[j] = -1
            cnt = 0
            for k in range(ip[j], ip[j + 1]):
                i = irn[k]
                dnew = a[k] - u[i]
                if dnew >= csp:
                    continue
                if iperm[i] == 0:
                    csp, isp, jsp = dnew, k, j
                else:
                    if dnew < dmin:
                        dmin = dnew
                    d[i] = dnew
                    cnt += 1
                    q[cnt] = k
            for kk in range(1, cnt + 1):
                k = q[kk]
                i = irn[k]
                if csp <= d[i]:
                    d[i] = RINF
                    continue
                if d[i] <= dmin:
                    low -= 1
                    q[low] = i
                    l[i] = low
                else:
                    Q.qlen += 1
                    l[i] = Q.qlen
                    Q.sift_up(i)
                jj = iperm[i]
                out[jj] = k
                pr[jj] = j
            for _ in range(num):
                if low == up:
                    if Q.qlen == 0:
                        break
                    i = q[1]
                    if d[i] >= csp:
                        break
                    dmin = d[i]
                    while True:
                        Q.pop_root()
                        low -= 1
                        q[low] = i
                        l[i] = low
                        if Q.qlen == 0:
                            break
                        i = q[1]
                        if d[i] > dmin:
                            break
                q0 = q[up - 1]
                dq0 = d[q0]
                if dq0 >= csp:
                    break
                up -= 1
                j = iperm[q0]
                vj = dq0 - a[jperm[j]] + u[q0]
                for k in range(ip[j], ip[j + 1]):
                    i = irn[k]
                    if l[i] >= up:
                        continue
                    dnew = vj + a[k] - u[i]
                    if dnew >= csp:
                        continue
                    if iperm[i] == 0:
                        csp, isp, jsp = dnew, k, j
                        continue
                    if d[i] <= dnew:
                        continue
                    if l[i] >= low:
                        continue
                    d[i] = dnew
                    if dnew <= dmin:
                        lpos = l[i]
                        if lpos != 0:
                            Q.delete_at(lpos)
                        low -= 1
                        q[low] = i
                        l[i] = low
                    else:
                        if l[i] == 0:
                            Q.qlen += 1
                            l[i] = Q.qlen
                        Q.sift_up(i)
                    jj = iperm[i]
                    out[jj] = k
                    pr[jj] = j
            if csp != RINF:
                num += 1
                i = irn[isp]
                iperm[i] = jsp
                jperm[jsp] = isp
                j = jsp
                for _ in range(num):
                    jj = pr[j]
                    if jj == -1:
                        break
                    k = out[j]
                    i = irn[k]
                    iperm[i] = jj
                    jperm[jj] = k
                    j = jj
                for kk in range(up, n + 1):
                    i = q[kk]
                    u[i] = u[i] + d[i] - csp
            for kk in range(low, n + 1):
                i = q[kk]
                d[i] = RINF
                l[i] = 0
            for kk in range(1, Q.qlen + 1):
                i = q[kk]
                d[i] = RINF
                l[i] = 0
    # ---- :2331-2352  column duals; unmatched rows get u = 0; completion of a deficient matching (HSLmc64XD)
    v = [0.0] * (n + 1)
    for j in range(1, n + 1):
        k = jperm[j]
        v[j] = a[k] - u[irn[k]] if k != 0 else 0.0
    for i in range(1, n + 1):
        if iperm[i] == 0:
            u[i] = 0.0
    if num != n:
        taken = [0] * (n + 1)
        free_rows = []
        for i in range(1, n + 1):
            if iperm[i] == 0:
                free_rows.append(i)
            else:
                taken[iperm[i]] = i
        t = 0
        for j in range(1, n + 1):
            if taken[j] == 0:
                iperm[free_rows[t]] = -j
                t += 1
    return iperm, num, u, v


def mc64_job5(n, colptr, rowind, val):
    """0-based CSC in (what the reference's wrapper passes are the CSR arrays of A, i.e. this sees A^T).
    Returns (perm, u, v, num): perm[i] = 0-based column matched to row i, or -(j+1) for a completed row."""
    ne = int(colptr[n])
    ip = [0] + [int(colptr[j]) + 1 for j in range(n + 1)]
    irn = [0] + [int(rowind[k]) + 1 for k in range(ne)]
    a = [0.0] * (ne + 1)
    colmax = [0.0] * (n + 1)
    rinf_n = RINF / n                                       # hslmc64.c:407-408
    for j in range(1, n + 1):                               # hslmc64.c:706-736
        fact = 0.0
        for k in range(ip[j], ip[j + 1]):
            a[k] = abs(float(val[k - 1]))
            if a[k] > fact:
                fact = a[k]
        colmax[j] = fact
        fact = math.log(fact) if fact != 0.0 else rinf_n
        for k in range(ip[j], ip[j + 1]):
            a[k] = fact - math.log(a[k]) if a[k] != 0.0 else rinf_n
    iperm, num, u, v = _match(n, ip, irn, a)
    if num == n:                                            # hslmc64.c:822-832
        for j in range(1, n + 1):
            v[j] = v[j] - math.log(colmax[j]) if colmax[j] != 0.0 else 0.0
    perm = [iperm[i] - 1 if iperm[i] > 0 else iperm[i] for i in range(1, n + 1)]
    return perm, u[1:], v[1:], num


def wbm_ordering(n, ia, ja, a):
    """MatGetOrdering_WBM (src/petsc_mat_wbm.c:13-61): hands the CSR arrays of A to the CSC interface, makes the
    permutation 0-based (:55), row IS = identity, column IS = perm (:57-58); the scalings are dropped (:56,59)."""
    perm, _, _, _ = mc64_job5(n, ia, ja, a)
    return list(range(n)), perm
