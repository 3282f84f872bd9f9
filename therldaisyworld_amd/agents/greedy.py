"""Scripted grazing policy (ref: daisy/agents/greedy.py:5-36), host mirror.

The policy maps a (B,N,7,3,3) observation to a (B,N,1) action code; it is O(B*N*4) integer work and
its *RNG draws* define same-seed parity (one ``np.random.rand()`` per call for the whole batch, then
``np.random.randint(9)`` per agent on the random branch), so this mirror keeps both on the host
NumPy stream.  The device-resident twin of the deterministic branch is ``Engine.policy_greedy``
(csrc/dw_agents.hpp ``policy_greedy``), used by ``therldaisyworld_amd.harness`` to run whole
episodes without shipping observations to the host.
"""
import numpy as np


class Greedy:

    def __init__(self, **kwargs):
        self.epsilon = kwargs.get("epsilon", 0.0)
        self.greedy = kwargs.get("greedy", True)
        # flat 3x3 indices of (row, col-1), (row-1, col), (row+1, col), (row, col+1)
        self.move_mask = np.array([[[3, 1, 7, 5]]])

    def draw_branch(self):
        """The per-call coin (ref :23): True -> deterministic branch."""
        return np.random.rand() > self.epsilon

    @staticmethod
    def draw_random_actions(batch, n_agents):
        """The epsilon branch (ref :32)."""
        return np.random.randint(9, size=(batch, n_agents, 1, 1)).reshape(batch, n_agents, -1)

    def __call__(self, obs):
        batch, n_agents = obs.shape[0:2]
        if not self.draw_branch():
            return self.draw_random_actions(batch, n_agents)
        # ref :17-30: food = light + dark of the patch, candidates = food at flat cells 3, 1, 7, 5, np.argmax / np.argmin
        # (first extremum).  Only the four candidate sums are formed (the same additions), and the first extremum is a
        # chain of three strict comparisons: np.argmax along an axis of length 4 costs 60 us for 1000 x 4 agents.
        obs = np.asarray(obs)
        c = [obs[:, :, 1, r, k] + obs[:, :, 2, r, k] for r, k in ((1, 0), (0, 1), (2, 1), (1, 2))]
        if any(np.isnan(x).any() for x in c):                # (np.argmax's NaN rule; covers are never NaN)
            cand = np.stack(c, axis=-1)
            pick = np.argmax(cand, axis=-1) if self.greedy else np.argmin(cand, axis=-1)
            return (4 + pick).reshape(batch, n_agents, -1)
        best = c[0]
        pick = np.full(best.shape, 4, dtype=np.int64)
        for i in (1, 2, 3):
            better = c[i] > best if self.greedy else c[i] < best
            if i < 3:
                best = np.where(better, c[i], best)
            pick[better] = 4 + i
        return pick.reshape(batch, n_agents, -1)
