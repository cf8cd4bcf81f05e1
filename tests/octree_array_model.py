"""Array ("data-parallel") formulation of the reference's DistributeOctTree (src/mdBRIEFextractorOct.cpp:631-861).

The reference walks a std::list and splices children to the front; the HIP kernel (csrc/mcs_octree.hip) instead
rebuilds the whole node list once per pass from prefix sums, never moving keys (a key only carries the list position
of its node).  This file is the executable specification of that formulation, written with the same passes the
kernel runs; tests/test_octree_model.py checks it against the literal list-based oracle on random inputs.
All coordinates are integers (FAST emits integer pixel positions), so the tree is pure integer arithmetic.
"""
import math


def _round_half_even(v):
    return int(round(v))  # python round() is half-to-even


def _split_geometry(nd):
    x0, x1, y0, y1 = nd
    hx = math.ceil((x1 - x0) / 2.0)
    hy = math.ceil((y1 - y0) / 2.0)
    mx, my = x0 + hx, y0 + hy
    # n1 (UL), n2 (UR), n3 (BL), n4 (BR)
    return mx, my, [(x0, mx, y0, my), (mx, x1, y0, my), (x0, mx, my, y1), (mx, x1, my, y1)]


def _quadrant(x, y, mx, my):
    if x < mx:
        return 0 if y < my else 2
    return 1 if y < my else 3


def distribute(xs, ys, resp, minX, maxX, minY, maxY, N):
    """returns the indices of the selected keys in output (= final list) order"""
    n = len(xs)
    W, H = maxX - minX, maxY - minY
    nIni = _round_half_even(W / H)
    if nIni < 1:
        return []
    hX = W / nIni
    # ---- roots: list order = i ascending, empty roots erased
    rootb = [int(hX * i) for i in range(nIni + 1)]
    rcnt = [0] * nIni
    kroot = [0] * n
    for k in range(n):
        r = int(xs[k] / hX)
        r = min(r, nIni - 1)
        kroot[k] = r
        rcnt[r] += 1
    pos_of_root, nodes = {}, []          # nodes: list-ordered dicts
    for i in range(nIni):
        if rcnt[i] > 0:
            pos_of_root[i] = len(nodes)
            nodes.append({"box": (rootb[i], rootb[i + 1], 0, H), "cnt": rcnt[i], "cre": -1})
    knode = [pos_of_root[kroot[k]] for k in range(n)]

    def child_counts(split_set):
        """for every node in split_set: geometry + per-quadrant key counts (parallel over keys on the GPU)"""
        geo = {i: _split_geometry(nodes[i]["box"]) for i in split_set}
        cc = {i: [0, 0, 0, 0] for i in split_set}
        kq = [0] * n
        for k in range(n):
            i = knode[k]
            if i in geo:
                q = _quadrant(xs[k], ys[k], geo[i][0], geo[i][1])
                kq[k] = q
                cc[i][q] += 1
        return geo, cc, kq

    def rebuild(proc_order, geo, cc, kq):
        """proc_order: nodes split in this pass, in PROCESSING order.  Children of later-processed nodes end up in
        front (push_front); inside one node the order is n4,n3,n2,n1; untouched nodes keep their relative order."""
        nonlocal nodes, knode
        proc = set(proc_order)
        nchild = {i: sum(1 for c in cc[i] if c > 0) for i in proc_order}
        total_children = sum(nchild.values())
        # suffix sums over processing order
        start, acc = {}, 0
        for i in reversed(proc_order):
            start[i] = acc
            acc += nchild[i]
        new_nodes = [None] * (total_children + len(nodes) - len(proc_order))
        childpos = {}
        cre = 0
        for i in proc_order:                      # creation order: processing order, n1..n4
            off = start[i]
            for q in (3, 2, 1, 0):                # list order inside the node: n4 first
                if cc[i][q] > 0:
                    childpos[(i, q)] = off
                    off += 1
            for q in (0, 1, 2, 3):
                if cc[i][q] > 0:
                    c = cc[i][q]
                    new_nodes[childpos[(i, q)]] = {"box": geo[i][2][q], "cnt": c, "cre": cre if c > 1 else -1}
                    if c > 1:
                        cre += 1
        keep_pos, r = {}, total_children
        for i in range(len(nodes)):
            if i not in proc:
                keep_pos[i] = r
                nd = dict(nodes[i]); nd["cre"] = -1
                new_nodes[r] = nd
                r += 1
        for k in range(n):
            i = knode[k]
            knode[k] = childpos[(i, kq[k])] if i in proc else keep_pos[i]
        nodes = new_nodes

    finish = False
    guard = 0
    while not finish:
        guard += 1
        assert guard < 200
        prev = len(nodes)
        # ---- phase A: split every node with more than one key, in list order
        proc_order = [i for i in range(len(nodes)) if nodes[i]["cnt"] > 1]
        geo, cc, kq = child_counts(set(proc_order))
        rebuild(proc_order, geo, cc, kq)
        nToExpand = sum(1 for nd in nodes if nd["cre"] >= 0)
        if len(nodes) >= N or len(nodes) == prev:
            finish = True
        elif len(nodes) + 3 * nToExpand > N:
            # ---- phase B: largest first (ties: later created first), stop as soon as the list reaches N
            while not finish:
                prev = len(nodes)
                cand = [i for i in range(len(nodes)) if nodes[i]["cre"] >= 0]
                cand.sort(key=lambda i: (nodes[i]["cnt"], nodes[i]["cre"]), reverse=True)
                geo, cc, kq = child_counts(set(cand))
                size, proc_order = len(nodes), []
                for i in cand:
                    proc_order.append(i)
                    size += sum(1 for c in cc[i] if c > 0) - 1
                    if size >= N:
                        break
                rebuild(proc_order, geo, cc, kq)
                if len(nodes) >= N or len(nodes) == prev:
                    finish = True
    # ---- best key per node: max response, first (lowest index) wins ties
    best = [None] * len(nodes)
    for k in range(n):
        i = knode[k]
        if best[i] is None or resp[k] > resp[best[i]]:
            best[i] = k
    return best
