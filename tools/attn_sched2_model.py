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


if __name__ == "__main__":
    for T in range(1, 70):
        check(T)
    print("attn SCHED=2 protocol model: barrier counts equal, ring reads/refills consistent for T = 1..69")
