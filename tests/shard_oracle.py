"""Test-only shard stepper backed by the CPU oracle (TEST INFRASTRUCTURE: lives under tests/,
never imported by fem_amd/).  Every rank steps the GLOBAL mesh with NaN in every vertex it does
not hold, so a missing, misplaced or stale ghost poisons the owned rows and is caught."""
import numpy as np
import torch

import oracle_lib as ol


class OracleShardStepper:
    def __init__(self, ordered, shard, precision, model, n_loc, c_loc):
        self.shard, self.V = shard, ordered.V
        self.stats = ol.Stats(ol.Topology(ordered.V, ordered.tri), ordered.points, ol.F64)
        self.state = np.ascontiguousarray(np.stack([n_loc, c_loc], axis=1))

    def step(self, nsteps, dt, model, stream=None):
        l2g = self.shard.loc2glob
        for _ in range(nsteps):
            gn, gc = np.full(self.V, np.nan), np.full(self.V, np.nan)
            gn[l2g], gc[l2g] = self.state[:, 0], self.state[:, 1]
            n, c = self.stats.step(gn, gc, dt, 1, model.as_array())
            self.state[:, 0], self.state[:, 1] = n[l2g], c[l2g]

    def pack(self, stream=None):
        return torch.from_numpy(self.state[self.shard.send_idx.astype(np.int64)])

    def ghost_view(self):
        return torch.from_numpy(self.state)[self.shard.n_own:]

    def get_owned(self):
        return self.state[:self.shard.n_own, 0].copy(), self.state[:self.shard.n_own, 1].copy()


class OracleTileStepper(OracleShardStepper):
    """The same oracle stepping behind the interface the overlapped schedule of ShardedRun drives
    (tile_classes / launch / flip / pack(which) / ghost_view(which)): "tiles" = chunks of 48 local
    rows, classified [interior | boundary | ghost] by the rule of fem_plan_create_tiled, two state
    buffers.  A launch advances the chosen tiles' rows by s steps from the CURRENT buffer into the
    NEXT one; rows of tiles that are not launched keep whatever the next buffer held (stale), so a
    schedule that reads a ghost before its exchange, or skips a tile it needs, gives wrong numbers
    or NaN in the owned rows."""

    def __init__(self, ordered, shard, precision, model, n_loc, c_loc, ksteps=2, chunk=48):
        super().__init__(ordered, shard, precision, model, n_loc, c_loc)
        import scipy.sparse as sp
        self.ksteps = ksteps
        topo = ol.Topology(ordered.V, ordered.tri)
        A = sp.csr_matrix((np.ones(len(topo.col), np.int8), topo.col.astype(np.int64), topo.rowptr.astype(np.int64)),
                          shape=(ordered.V, ordered.V))
        l2g = shard.loc2glob.astype(np.int64)
        n_own, n_loc_rows = shard.n_own, len(l2g)
        is_ghost = np.zeros(ordered.V, bool)
        is_ghost[l2g[n_own:]] = True
        is_send = np.zeros(ordered.V, bool)
        is_send[l2g[shard.send_idx.astype(np.int64)]] = True
        tiles, cls = [], []
        for b in list(range(0, n_own, chunk)) + list(range(n_own, n_loc_rows, chunk)):
            e = min(b + chunk, n_own if b < n_own else n_loc_rows)
            rows = np.arange(b, e)
            if b >= n_own:
                c = 2
            else:
                reach = np.zeros(ordered.V, bool)
                reach[l2g[rows]] = True
                for _ in range(ksteps):                       # rows the tile stages: its k rings
                    reach = reach | (A @ reach.astype(np.int8) > 0)
                c = 1 if (reach & is_ghost).any() or is_send[l2g[rows]].any() else 0
            tiles.append(rows)
            cls.append(c)
        order = np.argsort(np.array(cls), kind="stable")
        self.tiles = [tiles[i] for i in order]
        self.classes = tuple(int((np.array(cls) == c).sum()) for c in (0, 1, 2))
        self.bufs = [self.state, np.full_like(self.state, np.nan)]     # next buffer starts as garbage
        self.cur = 0
        self.launch_log = []

    def tile_classes(self):
        return self.classes

    def launch(self, first, count, nsteps, dt, model, stream=None):
        assert 1 <= nsteps <= self.ksteps and first + count <= len(self.tiles)
        self.launch_log.append((first, count, nsteps))
        if count == 0:
            return
        l2g = self.shard.loc2glob
        gn, gc = np.full(self.V, np.nan), np.full(self.V, np.nan)
        src = self.bufs[self.cur]
        gn[l2g], gc[l2g] = src[:, 0], src[:, 1]
        n, c = self.stats.step(gn, gc, dt, nsteps, model.as_array())
        rows = np.concatenate(self.tiles[first:first + count])
        dst = self.bufs[self.cur ^ 1]
        dst[rows, 0], dst[rows, 1] = n[l2g[rows]], c[l2g[rows]]

    def flip(self):
        self.cur ^= 1
        self.state = self.bufs[self.cur]

    def step(self, nsteps, dt, model, stream=None):           # the plain (non-overlapped) path
        done = 0
        while done < nsteps:
            s = min(self.ksteps, nsteps - done)
            self.launch(0, len(self.tiles), s, dt, model)
            self.flip()
            done += s

    def pack(self, stream=None, which=0):
        return torch.from_numpy(self.bufs[self.cur ^ which][self.shard.send_idx.astype(np.int64)].copy())

    def ghost_view(self, which=0):
        return torch.from_numpy(self.bufs[self.cur ^ which])[self.shard.n_own:]
