"""Second, independent restatement of the reference loops in pure Python (small cases
only).  Written straight from /root/reference, not from oracle/*.c, so that the C
oracle is checked by something other than itself.  Python floats are IEEE f64 and
CPython never contracts a*b+c, so results are comparable bit for bit."""
import math


def fora_setting(n, m, epsilon, alpha=0.2, rmax_scale=1.0, opt=False):
    # graph.h:177-178, algo.h:455-463
    delta = 1.0 / n
    pfail = 1.0 / n
    rmax = epsilon * math.sqrt(delta / 3 / m / math.log(2 / pfail))
    if opt:
        rmax *= rmax_scale / (1 - alpha)
    else:
        rmax *= rmax_scale
    omega = (2 + epsilon) * math.log(2 / pfail) / delta / epsilon / epsilon
    return rmax, omega


def push_fifo(adj, s, rmax, alpha=0.2):
    # algo.h:954-1018
    reserve, residue = {}, {}          # dict insertion order == iMap occur order
    rsum = 1.0
    if len(adj[s]) == 0:
        reserve[s] = 1
        return reserve, residue, 0.0
    inq = set([s])
    q = [s]
    residue[s] = 1.0
    left = 0
    while left < len(q):
        v = q[left]
        inq.discard(v)
        left += 1
        v_residue = residue[v]
        residue[v] = 0
        if v not in reserve:
            reserve[v] = v_residue * alpha
        else:
            reserve[v] += v_residue * alpha
        out = len(adj[v])
        rsum -= v_residue * alpha
        if out == 0:
            residue[s] += v_residue * (1 - alpha)
            if len(adj[s]) > 0 and residue[s] / len(adj[s]) >= rmax and s not in inq:
                inq.add(s)
                q.append(s)
            continue
        avg = ((1.0 - alpha) * v_residue) / out
        for nx in adj[v]:
            if nx not in residue:
                residue[nx] = avg
            else:
                residue[nx] += avg
            d = len(adj[nx])
            ratio = residue[nx] / d if d else math.inf  # x/0 = +inf for x > 0
            if ratio >= rmax and nx not in inq:
                inq.add(nx)
                q.append(nx)
    return reserve, residue, rsum


def walk_counts(residue, rsum, omega, alpha=0.2, opt=False):
    # query.h:270,282 / query.h:349,364
    check = rsum
    if opt:
        check *= (1 - alpha)
    N = int(omega * check)
    out = []
    for _, r in residue.items():
        if opt:
            r = r * (1 - alpha)
        out.append(int(math.ceil(r / check * N)))
    return N, out


M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return [c0, c1, c2, c3]


def walk(adj, seed, stream, rnd, start, j, alpha=0.2, no_zero_hop=False):
    # algo.h:124-142 / 144-166 under the Philox contract of oracle/fora_oracle.c
    if len(adj[start]) == 0:
        return start
    alpha32 = int(alpha * 4294967296.0)
    key = (seed & M32, (seed >> 32) & M32)
    cur = start
    t = 0
    w = None
    while True:
        if t % 2 == 0:
            w = philox4x32_10((start & M32, j & M32, ((j >> 32) & 0xFFFF) | ((rnd & 0xFF) << 16) | (((t >> 1) & 0xFF) << 24), (stream ^ ((t >> 9) * 0x9E3779B9)) & M32), key)
        ws, wm = w[(t & 1) * 2], w[(t & 1) * 2 + 1]
        if not (no_zero_hop and t == 0) and ws < alpha32:
            return cur
        d = len(adj[cur])
        if d > 0:
            cur = adj[cur][(wm * d) >> 32]
        else:
            cur = start
        t += 1
