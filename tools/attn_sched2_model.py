"""Discrete model of the synchronisation protocol of attn_fwd_kernel<D, 2 / 3> (BAGEL_ATTN_SCHED=2 / 3, csrc/attention.hip): the per-role
step programs, the 3-slot K/V^T ring and the DMA issue / wait rules, checked for every tile count T:
  * every role executes the same number of barriers (no deadlock);
  * every LDS read of tile t happens while slot t % 3 holds tile t, completely landed, and no refill of that slot is in flight;
  * no DMA is issued into a slot that some role still has to read in the current or a later step.
No GPU needed: python tools/attn_sched2_model.py"""


def programs(T):
    """role -> {global step: [(kind, tile)]}: reads of K(tile) / V(tile) per step, as the kernel's linear programs issue them."""
    last = 2 * T + 2
    A, B = {g: [] for g in range(last + 1)}, {g: [] for g in range(last + 1)}
    A[0].append(("K", 0))                                  # block_qk(0)
    for t in range(1, T):
        A[2 * t] += [("V", t - 1), ("K", t)]               # block_pv_qk(t)
    A[2 * T].append(("V", T - 1))                          # block_pv(T)
    B[1].append(("K", 0))
    for t in range(1, T):
        B[2 * t + 1] += [("V", t - 1), ("K", t)]
    B[2 * T + 1].append(("V", T - 1))
    for t in range(T):                                      # SCHED = 3: first V^T fragments of tile t at the end of its vector block
        A[2 * t + 1].append(("V", t))
        B[2 * t + 2].append(("V", t))
    return {"lead (waves 0-3)": A, "follow (waves 4-7)": B, "idle": {g: [] for g in range(last + 1)}}, last


def check(T):
    progs, last = programs(T)
    nbar = {r: len(p) for r, p in progs.items()}
    assert len(set(nbar.values())) == 1 and nbar["idle"] == 2 * T + 3, nbar
    # DMA timeline: tile u issued at the start of step issue[u], complete (waited + barrier) after the end of step ready[u]
    issue, ready = {0: -1, 1: -1}, {0: -1, 1: 1}           # prologue: tiles 0 and 1 requested, tile 0 waited before step 0
    for g in range(last + 1):
        if g % 2 == 0 and g >= 2:
            u = g // 2 + 1
            if u < T:
                issue[u] = g
                ready[u] = g + 1                            # vmcnt(0) at the end of the next (odd) step
    for u in range(T):
        assert u in issue, (T, u, "tile never requested")
    reads = {}                                              # tile -> (first, last) step it is read in, over all roles
    for role, prog in progs.items():
        for g, ops in prog.items():
            for kind, t in ops:
                assert ready[t] < g, (T, role, g, kind, t, "read before the tile is resident")
                f, l_ = reads.get(t, (g, g))
                reads[t] = (min(f, g), max(l_, g))
    for t in range(T):
        assert t in reads
    for u in range(3, T):                                   # slot u % 3 held tile u - 3
        assert issue[u] > reads[u - 3][1], (T, u, issue[u], reads[u - 3], "refill issued while the old tile is still read")
    for u in range(T):                                      # and nothing overwrites tile u before its last read
        if u + 3 < T:
            assert issue[u + 3] > reads[u][1]
    return nbar["idle"]


def check4(T):
    """SCHED = 4: 4-slot ring; waves 0-3 request their share of tile u at the start of step 2u-3 (one of their vector blocks),
    waves 4-7 at the start of step 2u-4; at the end of every odd step 2u-1 a wave waits until only the pieces of tile u+1 are
    still in flight (vmcnt(NL)), i.e. its share of tile u has landed."""
    progs, last = programs(T)
    NS = 4
    issue = {"lead": {0: -1, 1: -1}, "follow": {0: -1, 1: -1}}
    for g in range(last + 1):
        if g % 2 == 1:
            u = (g + 3) // 2
            if 2 <= u < T:
                issue["lead"][u] = g
        else:
            u = (g + 4) // 2
            if 2 <= u < T:
                issue["follow"][u] = g
    for half in issue:
        assert sorted(issue[half]) == list(range(T)) or T == 1 and sorted(issue[half]) == [0, 1], (T, half, issue[half])
    # the wait rule: at the end of odd step g, tile u = (g + 1) / 2 must be the OLDEST incomplete tile of the wave and exactly
    # the pieces of tile u + 1 (if it exists) may be newer
    ready = {}
    for half, iss in issue.items():
        for g in range(1, last + 1, 2):
            u = (g + 1) // 2
            if u >= T:
                continue
            newer = [v for v, gi in iss.items() if v > u and gi <= g and v < T]
            assert newer == ([u + 1] if u + 1 < T else []), (T, half, g, u, newer)
            ready[(half, u)] = g
    ready_all = {u: max(ready.get(("lead", u), -1), ready.get(("follow", u), -1)) for u in range(1, T)}
    ready_all[0] = -1
    reads = {}
    for role, prog in progs.items():
        for g, ops in prog.items():
            for kind, t in ops:
                assert ready_all[t] < g, (T, role, g, kind, t)
                f, l_ = reads.get(t, (g, g))
                reads[t] = (min(f, g), max(l_, g))
    for u in range(NS, T):
        for half in issue:
            assert issue[half][u] > reads[u - NS][1], (T, half, u, issue[half][u], reads[u - NS])
    return True


if __name__ == "__main__":
    for T in range(1, 70):
        check(T)
        check4(T)
    print("attn SCHED=2/3/4 protocol models: barrier counts equal, ring reads / refills / waits consistent for T = 1..69")
