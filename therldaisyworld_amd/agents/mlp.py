"""Learned grazing policy (ref: daisy/agents/mlp.py:12-146) — SURVEY.md §8(f) row N3.

A 63 -> 16 -> 32 -> 9 ReLU network on the flattened (7,3,3) observation; the action is the argmax of the
logits.  ``MLP`` keeps the reference's parameter interface (``get_parameters`` / ``set_parameters``: one flat
float64 vector, the weight matrices raveled in layer order; ``make_config`` / ``_apply_config``) so that
populations of an evolution strategy plug in unchanged.  ``__call__(obs)`` is the host mirror for callers
that hold NumPy observations; ``act_on_device(env, agent_begin, agent_end)`` evaluates the same network in a
HIP kernel (``policy_mlp``) straight from the device-resident state into the device action buffer, which is
what ``therldaisyworld_amd.harness.get_fitness`` uses.
"""
import json
import os

import numpy as np

DEFAULT_CONFIG_PATH = os.path.join("results", "default_mlp_config.json")     # ref mlp.py:58,69,79


def glorot(dims):
    """ref: daisy/nn/functional.py:4-10 (one ``np.random.randn`` draw per matrix)."""
    assert len(dims) == 2
    return np.random.randn(*dims) * np.sqrt(2 / (dims[0] + dims[1]))


class MLP:

    def __init__(self, **kwargs):
        self.in_dim = 63
        self.out_dim = 9
        self.h_dim = [16, 32]
        self.act_name = "relu"
        self.initialize_parameters()

    def _shapes(self):
        dims = [self.in_dim, *self.h_dim, self.out_dim]
        return list(zip(dims[:-1], dims[1:]))

    def initialize_parameters(self):
        self.layers = [glorot(shape) for shape in self._shapes()]

    def get_parameters(self):
        return np.concatenate([layer.ravel() for layer in self.layers])

    def set_parameters(self, parameters):
        parameters = np.asarray(parameters, dtype=np.float64)
        start = 0
        for ii, (a, b) in enumerate(self._shapes()):
            self.layers[ii] = parameters[start:start + a * b].reshape(a, b)
            start += a * b

    def make_config(self, include_parameters=True):
        config = {"in_dim": self.in_dim, "out_dim": self.out_dim, "h_dim": self.h_dim, "act_name": self.act_name}
        if include_parameters:
            config["parameters"] = list(self.get_parameters())
        return config

    def _apply_config(self, config):
        self.in_dim, self.out_dim = config["in_dim"], config["out_dim"]
        self.h_dim, self.act_name = config["h_dim"], config["act_name"]
        self.initialize_parameters()
        if "parameters" in config:
            self.set_parameters(np.array(config["parameters"]))

    def save_config(self, filepath=None):
        """ref mlp.py:55-63: the config (with parameters) as one JSON object."""
        with open(DEFAULT_CONFIG_PATH if filepath is None else filepath, "w") as f:
            json.dump(self.make_config(), f)

    def load_config(self, filepath=None):
        """ref mlp.py:66-74"""
        with open(DEFAULT_CONFIG_PATH if filepath is None else filepath, "r") as f:
            return json.load(f)

    def restore_config(self, filepath=None):
        """ref mlp.py:76-83 (which raises in `_apply_config`, mlp.py:41; here it loads the file, so the
        result files the reference ships, e.g. results/cmaes_exp_002/*_best_agent_gen127.json, restore)."""
        self._apply_config(self.load_config(filepath))

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = np.matmul(x, layer)
            x = x * (x > 0.0)
        return np.matmul(x, self.layers[-1])

    def get_action(self, obs):
        x = obs.reshape(*obs.shape[:-3], self.in_dim)
        return np.argmax(self.forward(x), axis=-1, keepdims=True)

    def __call__(self, obs):
        return self.get_action(obs)

    def act_on_device(self, env, agent_begin=0, agent_end=None):
        """Fill the environment's DEVICE action buffer for agents [agent_begin, agent_end) from the current
        device-resident observations (no observation download)."""
        env._sync_to_device()
        env._engine.policy_mlp(self.get_parameters(), agent_begin, agent_end, env._L_pass)

    def reset(self):
        pass
