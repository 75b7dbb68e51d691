"""Offline simulation of wave scheduling policies for the closest-hit traversal, from the per-ray phase traces that
tools/raytrace_dump.py writes (node steps of every node phase between two leaf visits, in queue order).

A wave runs 16 rays in lockstep with the while-while loop of traverse_quad: an outer iteration = a node phase that lasts
until every active ray reached a leaf (or finished), then one leaf step for those at a leaf.  Cost model: node step
NODE instructions, leaf step LEAF instructions (wave-level)."""
import sys
import numpy as np

NODE, LEAF = 34.0, 48.0

def load(path):
    z = np.load(path)
    st, tr = z["steps"], z["trace"].astype(np.int32)
    nn, nl = (st & 0xfff).astype(np.int32), ((st >> 12) & 0xff).astype(np.int32)
    ok = nl < 16
    # phases: nl+1 node phases (the last one ends the traversal)
    return nn, nl, tr, ok

def static_rounds(tr, nl, group=16, order=None):
    """cost of static rounds of `group` consecutive rays"""
    n = tr.shape[0] // group * group
    idx = np.arange(n) if order is None else order[:n]
    t = tr[idx].reshape(-1, group, 16)
    l = nl[idx].reshape(-1, group)
    # phase k is active for ray r iff k <= nl[r]; node phase cost = max over active rays of tr[.,k]
    k = np.arange(16)[None, None, :]
    active = k <= l[:, :, None]
    node_steps = np.where(active, t, 0).max(axis=1).sum(axis=1)            # per wave
    leaf_steps = l.max(axis=1)
    useful_node = np.where(active, t, 0).sum(axis=(1, 2))
    useful_leaf = l.sum(axis=1)
    cost = NODE * node_steps + LEAF * leaf_steps
    useful = NODE * useful_node + LEAF * useful_leaf
    return cost.sum(), useful.sum() / group, node_steps.sum(), leaf_steps.sum()

def simulate_pool(tr, nl, per_wave=256, refill_at=6, refill_cost=60.0, cap_iters=None, group=16):
    """per-wave pool: a wave owns `per_wave` consecutive rays; at the top of an outer iteration, if >= refill_at quads
    are idle (or all), idle quads take the next rays of the pool (cost refill_cost instructions per refill event)."""
    n = tr.shape[0]
    total = 0.0
    events = 0
    for w0 in range(0, n, per_wave):
        t = tr[w0:w0 + per_wave]; l = nl[w0:w0 + per_wave]
        m = t.shape[0]
        nxt = 0
        ray = -np.ones(group, np.int64); phase = np.zeros(group, np.int64)
        while True:
            idle = ray < 0
            ni = idle.sum()
            if ni == group and nxt >= m:
                break
            if ni == group or (ni >= refill_at and nxt < m):
                take = min(ni, m - nxt)
                slots = np.nonzero(idle)[0][:take]
                ray[slots] = np.arange(nxt, nxt + take); phase[slots] = 0
                nxt += take
                total += refill_cost; events += 1
            act = ray >= 0
            if not act.any():
                continue
            r = ray[act]; p = phase[act]
            steps = t[r, np.minimum(p, 15)]
            total += NODE * steps.max()
            # after the node phase: rays with p == l[r] are finished, others take a leaf step
            fin = p >= l[r]
            if (~fin).any():
                total += LEAF
            a = np.nonzero(act)[0]
            ray[a[fin]] = -1
            phase[a[~fin]] += 1
    return total, events

def simulate_capped(tr, nl, cap_iters=3, group=16, spill_cost=40.0, restore_cost=40.0):
    """static rounds of 16 capped at cap_iters outer iterations; unfinished rays are suspended and re-grouped among
    themselves (per wave stream: approximated globally in queue order), repeated until none are left."""
    cur_tr, cur_l = tr.copy(), nl.copy()
    total = 0.0
    level = 0
    while cur_tr.shape[0] > 0:
        n = cur_tr.shape[0]
        pad = (-n) % group
        if pad:
            cur_tr = np.concatenate([cur_tr, np.zeros((pad, 16), cur_tr.dtype)]); cur_l = np.concatenate([cur_l, -np.ones(pad, cur_l.dtype)])
        t = cur_tr.reshape(-1, group, 16); l = cur_l.reshape(-1, group)
        k = np.arange(16)[None, None, :]
        lim = np.minimum(l, cap_iters - 1) if cap_iters else l
        active = (k <= lim[:, :, None]) & (l[:, :, None] >= 0)
        node_steps = np.where(active, t, 0).max(axis=1).sum(axis=1)
        leaf_steps = np.maximum(np.minimum(l, cap_iters if cap_iters else l), 0).max(axis=1)
        total += (NODE * node_steps + LEAF * leaf_steps).sum()
        if level > 0:
            total += restore_cost * t.shape[0]
        if not cap_iters:
            break
        susp = (cur_l >= cap_iters)
        any_susp_wave = susp.reshape(-1, group).any(axis=1)
        total += spill_cost * any_susp_wave.sum()
        nt = cur_tr[susp][:, cap_iters:]
        nt = np.concatenate([nt, np.zeros((nt.shape[0], cap_iters), nt.dtype)], axis=1)
        cur_tr, cur_l = nt, cur_l[susp] - cap_iters
        level += 1
    return total

if __name__ == "__main__":
    for path in sys.argv[1:]:
        nn, nl, tr, ok = load(path)
        print(path, "rays", nn.size, "node/ray %.2f leaf/ray %.2f" % (nn.mean(), nl.mean()), "traces complete: %.4f" % ok.mean())
        nl = np.minimum(nl, 15)
        cost, useful, ns, ls = static_rounds(tr, nl)
        print("  static 16: wave instr %.3e  lane use %.3f  (wave node steps %d, leaf steps %d)" % (cost, useful / cost, ns, ls))
        ideal = useful
        for cap in (2, 3, 4, 6):
            c = simulate_capped(tr, nl, cap)
            print("  capped at %d outer iterations + regroup: %.3e  (%.3f of static, lane use %.3f)" % (cap, c, c / cost, ideal / c))
        sub = slice(0, min(nn.size, 200000))
        cs, _, _, _ = static_rounds(tr[sub], nl[sub])
        for thr in (4, 6, 8, 10):
            for rc in (40.0, 80.0):
                c, ev = simulate_pool(tr[sub], nl[sub], per_wave=256, refill_at=thr, refill_cost=rc)
                print("  pool refill>=%d cost %d: %.3f of static (%d refill events)" % (thr, rc, c / cs, ev))


def simulate_threshold(tr, nl, K=4, group=16, nmax=None, leaf_first=False):
    """lockstep wave of `group` rays; policy: do node steps while at least K rays are at a node (or nobody waits at a
    leaf), otherwise one leaf step for every ray waiting at a leaf.  Returns (cost, node_steps, leaf_steps)."""
    n = tr.shape[0] // group * group
    if nmax:
        n = min(n, nmax // group * group)
    t = tr[:n].reshape(-1, group, 16).astype(np.int64)
    l = nl[:n].reshape(-1, group).astype(np.int64)
    W = t.shape[0]
    phase = np.zeros((W, group), np.int64)
    left = t[:, :, 0].copy()                 # node steps left in the current phase
    done = np.zeros((W, group), bool)
    # a ray with left == 0 and phase < l is at a leaf; with left == 0 and phase == l it is done
    def settle():
        nonlocal done
        fin = (~done) & (left == 0) & (phase >= l)
        done |= fin
    settle()
    node_steps = np.zeros(W, np.int64); leaf_steps = np.zeros(W, np.int64)
    wi = np.arange(W)
    while True:
        at_node = (~done) & (left > 0)
        at_leaf = (~done) & (left == 0)
        a = at_node.sum(1); b = at_leaf.sum(1)
        alive = (a + b) > 0
        if not alive.any():
            break
        do_node = alive & ((a >= K) | (b == 0)) & (a > 0)
        do_leaf = alive & ~do_node
        node_steps += do_node
        leaf_steps += do_leaf
        left -= (at_node & do_node[:, None])
        adv = at_leaf & do_leaf[:, None]
        phase += adv
        nxt = np.take_along_axis(t, np.minimum(phase, 15)[:, :, None], axis=2)[:, :, 0]
        left = np.where(adv, nxt, left)
        settle()
    return NODE * node_steps.sum() + LEAF * leaf_steps.sum(), node_steps.sum(), leaf_steps.sum()
