"""(theta, trajectory) pair front end — what replaces Isaac Gym on this path.

The reference obtains training pairs from ``collect_trajectories`` over a
closed simulator (utils/collect_trajectories.py:15-93); its output contract is
``params [N, D]``, ``states [N, T+1, sd]``, ``actions [N, T+1, ad]`` (fp32).
This module produces tensors with that contract from

  * recorded pairs in the reference's regression-test ``.npz`` layout
    (``params`` [N, D], ``data`` [N, (T)*(sd+ad)] with [s_t | a_t] per step:
    bayes_sim_ig/tests/regression_tests.py:31-43), and
  * a vectorised restatement of the reference's numpy Pendulum
    (sim/openai_env_wrappers.py:77-93 reset, :153-177 dynamics; parameters
    (length, mass) ~ U[0.01, 2]^2 as in cfg/pendulum.yaml:36-50) driven by the
    reference's collection policies ``policy_random`` / ``policy_ones``
    (collect_trajectories.py:96-101).

Host-side data preparation (numpy); not part of the accelerated path.
"""
import numpy as np
import torch


def load_pairs_npz(path, state_dim, act_dim=1, device='cpu'):
    """-> (params [N,D], states [N,T,sd], actions [N,T,ad]) float32 tensors."""
    loaded = np.load(path)
    params = torch.from_numpy(np.asarray(loaded['params'])).float()
    data = torch.from_numpy(np.asarray(loaded['data'])).float()
    if params.dim() == 1:                      # a single recorded trajectory
        params, data = params.reshape(1, -1), data.reshape(1, -1)
    step = state_dim + act_dim
    assert data.shape[1] % step == 0, 'row width is not a multiple of state_dim + act_dim'
    sa = data.reshape(params.shape[0], -1, step)
    return (params.to(device), sa[:, :, :state_dim].contiguous().to(device),
            sa[:, :, state_dim:].contiguous().to(device))


def save_pairs_npz(path, params, states, actions):
    """Inverse of load_pairs_npz (same layout as the reference's test data)."""
    sa = torch.cat([states, actions], dim=-1).reshape(states.shape[0], -1)
    np.savez_compressed(path, params=params.detach().cpu().numpy().astype(np.float64),
                        data=sa.detach().cpu().numpy().astype(np.float64))


def pendulum_pairs(n, traj_len, policy='random', lows=(0.01, 0.01), highs=(2.0, 2.0),
                   params=None, seed=None, device='cpu'):
    """n Pendulum episodes of ``traj_len`` actions -> the collect_trajectories
    contract: params [n,2] = (length, mass), states [n, traj_len+1, 3] =
    (cos th, sin th, thdot), actions [n, traj_len+1, 1] in [-1,1] (the last
    action repeated, as pad_states_actions does).

    Dynamics (openai_env_wrappers.py:162-172): dt = 0.05, g = 10,
    torque u = clip(2*a, -2, 2);
      thdot' = thdot + (-3g/(2l) sin(th+pi) + 3/(m l^2) u) dt;  th' = th + thdot' dt;
      thdot' clipped to [-8, 8] after th' is formed.
    Initial state th ~ U[-pi, pi], thdot ~ U[-1, 1] (:85-90)."""
    rs = np.random.RandomState(seed) if seed is not None else np.random
    lows, highs = np.asarray(lows, dtype=np.float64), np.asarray(highs, dtype=np.float64)
    if params is None:
        theta = rs.uniform(lows, highs, size=(n, 2))
    else:
        theta = np.broadcast_to(np.asarray(params, dtype=np.float64), (n, 2)).copy()
    length, mass = theta[:, 0], theta[:, 1]
    th = rs.uniform(-np.pi, np.pi, size=n)
    thdot = rs.uniform(-1.0, 1.0, size=n)
    max_speed, max_torque, dt, g = 8.0, 2.0, 0.05, 10.0
    states = np.empty((n, traj_len + 1, 3))
    actions = np.empty((n, traj_len + 1, 1))
    states[:, 0] = np.column_stack([np.cos(th), np.sin(th), thdot])
    for t in range(traj_len):
        if policy in ('random', 'policy_random'):
            act = rs.rand(n)                                    # torch.rand_like: U[0,1)
        elif policy in ('ones', 'policy_ones'):
            act = np.ones(n)
        else:
            raise ValueError('unknown collection policy %r' % (policy,))
        u = np.clip(act * max_torque, -max_torque, max_torque)
        newthdot = thdot + (-3 * g / (2 * length) * np.sin(th + np.pi) +
                            3.0 / (mass * length ** 2) * u) * dt
        th = th + newthdot * dt
        thdot = np.clip(newthdot, -max_speed, max_speed)
        actions[:, t, 0] = act
        states[:, t + 1] = np.column_stack([np.cos(th), np.sin(th), thdot])
    actions[:, traj_len] = actions[:, traj_len - 1]             # pad (summarizers.py:52-58)
    to = lambda a: torch.from_numpy(a).float().to(device)      # noqa: E731
    return to(theta), to(states), to(actions)
