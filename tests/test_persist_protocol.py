"""Host model of the hand-over protocol of tile_step_persist_kernel (fem_amd/csrc/kernels_step.hpp): a fixed set of
workgroups, each bound to one of eight contiguous tile ranges, draws (group, tile) items from its range's ticket in
group-major order (workgroups start in launch order, so each range has a live workgroup as soon as eight are
resident, and keeps it until the range is exhausted); item (g, T) may run once every tile in dep[T] (the owners of
T's ring rows, and T) has published g groups; when done it publishes flag[T] = g + 1.  The model runs the protocol under random interleavings, with fewer
workgroups resident than launched, and checks what the kernel relies on:
  * it ends (no workgroup waits for an item nobody will draw), and every item runs exactly once;
  * two neighbouring tiles never run different groups at the same time -- tile T in group g reads its neighbours' rows
    in buffer g % 2 and writes its own in the other, so a neighbour one group ahead would overwrite what T reads, and
    one group behind would read what T overwrites (the ping-pong buffers' read-after-write and write-after-read hazards).
No GPU, no library: the product's kernel is exercised by tests/test_gpu_persist.py."""
import random

import pytest


def ranges(ntiles):
    """the kernel's eight tile ranges: [lo, lo + n) for workgroups with blockIdx % 8 == x"""
    q8, r8 = ntiles >> 3, ntiles & 7
    return [((x * (q8 + 1)) if x < r8 else (r8 * (q8 + 1) + (x - r8) * q8), q8 + (1 if x < r8 else 0)) for x in range(8)]


def ring_deps(ntiles, reach):
    """tiles on a circle, each reading `reach` neighbours either side (symmetric, itself included)"""
    return [sorted({(t + d) % ntiles for d in range(-reach, reach + 1)}) for t in range(ntiles)]


def simulate(ntiles, ngroups, deps, nworkers, resident, seed, max_ticks=2_000_000):
    rng = random.Random(seed)
    rg = ranges(ntiles)
    ticket = [0] * 8
    flag = [0] * ntiles
    running = {}                                   # tile -> group, while its body runs
    ran = set()
    # worker: [state, item, ticks left]; state: 0 draw, 1 wait, 2 run, 3 gone
    workers = [[0, None, 0] for _ in range(nworkers)]
    active = list(range(min(resident, nworkers)))  # dispatch order: the first `resident` workgroups start
    waiting = list(range(len(active), nworkers))
    for _ in range(max_ticks):
        if not active:
            break
        w = rng.choice(active)
        st = workers[w]
        x = w & 7
        lo, n = rg[x]
        if st[0] == 0:
            item = ticket[x]
            ticket[x] += 1
            if item >= n * ngroups:
                st[0] = 3
                active.remove(w)
                if waiting:
                    active.append(waiting.pop(0))  # its slot goes to the next workgroup of the launch
                continue
            st[1] = (item // n, lo + item % n)
            st[0] = 1
        elif st[0] == 1:
            g, t = st[1]
            if all(flag[d] >= g for d in deps[t]):
                for d in deps[t]:
                    assert d not in running or running[d] == g, "tile %d in group %d beside neighbour %d in group %d" % (t, g, d, running[d])
                assert (g, t) not in ran
                ran.add((g, t))
                running[t] = g
                st[0], st[2] = 2, rng.randint(1, 6)
        elif st[0] == 2:
            st[2] -= 1
            if st[2] == 0:
                g, t = st[1]
                del running[t]
                assert flag[t] == g
                flag[t] = g + 1
                st[0] = 0
    else:
        pytest.fail("the protocol did not end: %d workgroups still active" % len(active))
    assert len(ran) == ntiles * ngroups and all(f == ngroups for f in flag)


@pytest.mark.parametrize("ntiles,ngroups,reach,nworkers,resident", [
    (2, 3, 1, 8, 8), (13, 4, 2, 16, 16), (64, 5, 3, 64, 64), (263, 3, 4, 512, 512), (100, 6, 2, 128, 24),
    (57, 4, 9, 64, 8), (9, 7, 1, 24, 8), (40, 3, 40, 32, 32)])
def test_flag_protocol_ends_and_keeps_neighbours_in_step(ntiles, ngroups, reach, nworkers, resident):
    deps = ring_deps(ntiles, min(reach, (ntiles - 1) // 2) if ntiles > 2 else 1)
    for seed in range(3):
        simulate(ntiles, ngroups, deps, nworkers, resident, seed)


def test_ranges_cover_every_tile_once():
    for ntiles in (1, 2, 7, 8, 9, 263, 1056, 4096):
        rg = ranges(ntiles)
        tiles = [t for lo, n in rg for t in range(lo, lo + n)]
        assert tiles == list(range(ntiles))
