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
        food = (obs[..., 1, :, :] + obs[..., 2, :, :]).reshape(batch, n_agents, 9)
        candidates = food[:, :, self.move_mask[0, 0]]
        if self.draw_branch():
            pick = np.argmax(candidates, axis=-1) if self.greedy else np.argmin(candidates, axis=-1)
            return (4 + pick).reshape(batch, n_agents, -1)
        return self.draw_random_actions(batch, n_agents)
