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
