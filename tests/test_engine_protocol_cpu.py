"""CPU: a discrete-event model of the decode engine's intra-workgroup protocol (csrc/engine.hip) -- loaders, consumers, ring slots, the `full` / `freed` words, the
retire-oldest rule and the phase hand-offs -- run on random geometries and random latencies.

What it pins without a GPU: with the clamps the host applies (slots a multiple of the loader count; depth < slots per loader) every unit is issued, published,
consumed and freed exactly once, in an order in which no wave waits on something that can only happen after its own next action (no deadlock), whatever the DMA
and consumer latencies are; and the clamp is NECESSARY (depth == slots per loader dead-locks).  The kernel's bit-identity with the gemv chain is the GPU tests'
business (tests/test_engine_gpu.py); this model is about progress."""
import random

import pytest


def simulate(units_per_phase, nl, nc, nslot, depth, rng, max_events=200000):
    """Event loop.  Returns ("done", steps) or ("deadlock", state)."""
    T = sum(units_per_phase)
    phase_end = []
    acc = 0
    for u in units_per_phase:
        acc += u
        phase_end.append(acc)
    full = [0] * nslot
    freed = [0] * nslot
    landed = set()                                  # units whose DMA has completed (latency: a random number of scheduler turns)
    inflight = {}                                   # unit -> remaining turns
    L = [dict(q=j, issued=[], published=0) for j in range(nl)]       # loader j: next unit, its issued-but-unpublished units
    C = [dict(q=c, phase=0, state="stage") for c in range(nc)]
    staged = [0] * len(units_per_phase)             # consumers that passed the hand-off of a phase
    done_cnt = [0] * len(units_per_phase)           # consumers that finished a phase
    consumed = []
    for step in range(max_events):
        progressed = False
        # DMA flight
        for u in list(inflight):
            inflight[u] -= 1
            if inflight[u] <= 0:
                landed.add(u)
                del inflight[u]
                progressed = True
        order = [("L", j) for j in range(nl)] + [("C", c) for c in range(nc)]
        rng.shuffle(order)
        for kind, i in order:
            if kind == "L":
                ld = L[i]
                # (3) retire when more than `depth` of its units are unpublished -- blocking: nothing else happens for this loader until the oldest landed
                if len(ld["issued"]) > depth or (ld["q"] >= T and ld["issued"]):
                    u = ld["issued"][0]
                    if u in landed:
                        full[u % nslot] = u + 1
                        ld["issued"].pop(0)
                        progressed = True
                    continue
                if ld["q"] >= T:
                    continue
                q = ld["q"]
                # (1) the slot
                if q >= nslot and freed[q % nslot] < q - nslot + 1:
                    continue
                # (2) issue
                inflight[q] = rng.randint(1, 6)
                ld["issued"].append(q)
                ld["q"] = q + nl
                progressed = True
            else:
                cs = C[i]
                ph = cs["phase"]
                if ph >= len(units_per_phase):
                    continue
                if cs["state"] == "stage":
                    # hand-off: every consumer of the workgroup (and, in the kernel, of every other workgroup) has finished the previous phase
                    if ph == 0 or done_cnt[ph - 1] == nc:
                        staged[ph] += 1
                        cs["state"] = "run"
                        progressed = True
                    continue
                q = cs["q"]
                if q >= phase_end[ph]:
                    done_cnt[ph] += 1
                    cs["phase"] = ph + 1
                    cs["state"] = "stage"
                    progressed = True
                    continue
                if full[q % nslot] >= q + 1:
                    assert q in landed, "a unit was published before its DMA landed"
                    consumed.append(q)
                    freed[q % nslot] = q + 1
                    cs["q"] = q + nc
                    progressed = True
        if all(c["phase"] >= len(units_per_phase) for c in C) and all(ld["q"] >= T and not ld["issued"] for ld in L):
            assert sorted(consumed) == list(range(T)), "every unit exactly once"
            return "done", step
        if not progressed and not inflight:
            return "deadlock", dict(L=L, C=C, full=full, freed=freed)
    return "timeout", None


@pytest.mark.parametrize("seed", range(40))
def test_engine_protocol_makes_progress_on_random_geometries(seed):
    rng = random.Random(seed)
    nl = rng.choice([1, 2])
    nc = rng.randint(1, 7)
    nslot = rng.randint(max(3, 2 * nl), 8)
    nslot -= nslot % nl                                            # the host's clamp: every loader owns the slots of its units
    depth = rng.randint(1, nslot // nl - 1)                        # the host's clamp: depth < slots per loader
    phases = [rng.randint(0, 40) for _ in range(rng.randint(1, 4))]
    res, info = simulate(phases, nl, nc, nslot, depth, rng)
    assert res == "done", (res, nl, nc, nslot, depth, phases, info)


def test_engine_protocol_needs_the_depth_clamp():
    """depth == slots per loader: a loader may hold every one of its slots unpublished while it waits for one of them to be handed back -- by a consumer that is
    waiting for exactly those units to be published."""
    dead = 0
    for seed in range(20):
        res, _ = simulate([30, 30], 1, 3, 4, 4, random.Random(seed))
        dead += res == "deadlock"
    assert dead > 0, "the model should exhibit the deadlock the host clamp prevents"
    for seed in range(20):
        res, _ = simulate([30, 30], 1, 3, 4, 3, random.Random(seed))
        assert res == "done"
