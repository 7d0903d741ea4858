"""Pair-data front end (bayes_sim_ig_amd/pairs.py): the reference's recorded
.npz layout and the restated Pendulum generator, checked against the data file
the reference's regression test holds (tests/golden/pendulum_ref.npz slice)."""
import os

import numpy as np
import torch

from conftest import GOLDEN, golden
from bayes_sim_ig_amd import pairs


def test_load_reference_npz_layout(tmp_path):
    g = golden('pendulum_ref.npz')
    path = os.path.join(GOLDEN, 'pendulum_ref.npz')
    th, st, ac = pairs.load_pairs_npz(path, state_dim=3)
    assert th.shape == (1000, 2) and st.shape == (1000, 10, 3) and ac.shape == (1000, 10, 1)
    np.testing.assert_array_equal(st[5, 2].numpy(), g['data'][5].reshape(10, 4)[2, :3])
    out = str(tmp_path / 'rt.npz')
    pairs.save_pairs_npz(out, th, st, ac)
    th2, st2, ac2 = pairs.load_pairs_npz(out, state_dim=3)
    assert torch.equal(th, th2) and torch.equal(st, st2) and torch.equal(ac, ac2)


def test_pendulum_dynamics_reproduce_reference_recordings():
    """One step of the restated dynamics from every recorded state reproduces
    the reference's recorded next state (its 'ones' policy: torque 2.0)."""
    g = golden('pendulum_ref.npz')
    rows = g['data'][:200].reshape(200, 10, 4).astype(np.float64)
    length, mass = g['params'][:200, 0].astype(np.float64), g['params'][:200, 1].astype(np.float64)
    for t in range(9):
        th = np.arctan2(rows[:, t, 1], rows[:, t, 0])
        thdot = rows[:, t, 2]
        u = np.clip(rows[:, t, 3], -2.0, 2.0)
        newthdot = thdot + (-3 * 10.0 / (2 * length) * np.sin(th + np.pi) +
                            3.0 / (mass * length ** 2) * u) * 0.05
        newth = th + newthdot * 0.05
        newthdot = np.clip(newthdot, -8.0, 8.0)
        pred = np.column_stack([np.cos(newth), np.sin(newth), newthdot])
        np.testing.assert_allclose(pred, rows[:, t + 1, :3], rtol=2e-5, atol=2e-5)


def test_pendulum_generator_contract():
    th, st, ac = pairs.pendulum_pairs(64, 20, policy='random', seed=3)
    assert th.shape == (64, 2) and st.shape == (64, 21, 3) and ac.shape == (64, 21, 1)
    assert th.dtype == st.dtype == ac.dtype == torch.float32
    assert float(th.min()) >= 0.01 and float(th.max()) <= 2.0
    np.testing.assert_allclose((st[..., 0] ** 2 + st[..., 1] ** 2).numpy(), 1.0, atol=1e-5)
    assert float(st[..., 2].abs().max()) <= 8.0 + 1e-6
    assert float(ac.min()) >= 0.0 and float(ac.max()) < 1.0
    assert torch.equal(ac[:, -1], ac[:, -2])                     # padded last action
    th2, st2, ac2 = pairs.pendulum_pairs(64, 20, policy='random', seed=3)
    assert torch.equal(st, st2) and torch.equal(th, th2)
    # the generator itself follows the recorded reference transitions
    _, so, ao = pairs.pendulum_pairs(8, 9, policy='ones', params=(1.0, 0.5), seed=1)
    assert float(ao.min()) == 1.0
