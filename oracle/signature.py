"""ORACLE (test infrastructure only) — truncated path signature.

The reference calls ``signatory.signature(paths, depth)``
(bayes_sim_ig/utils/summarizers.py:158,164).  signatory is a third-party
C++/CUDA package that is NOT under /root/reference and is unpinned
(README.md:345-350 "git clone master") — PARITY UNPINNED against signatory.
This file restates the published definition:

  S^k(X)_{i1..ik} = integral over 0<t1<..<tk<1 of dX^{i1}_{t1} ... dX^{ik}_{tk}

for the piecewise-linear path through the given points (basepoint=False),
levels 1..depth concatenated, each level flattened in C order
(index (i1..ik) -> sum_j i_j d^(k-j)), which is signatory's output layout.

Two independent evaluations are given so they can check each other:
  * ``signature``       — Chen's identity, S <- S (x) exp(delta) per segment
  * ``signature_brute`` — numpy fp64 iterated sums straight from the
                          definition for piecewise-linear paths (depth <= 3)
Known-answer vectors are in tests/test_oracle_summaries.py.
"""
import numpy as np
import torch


def signature(paths, depth):
    """paths [N, L, d] -> [N, d + d^2 + ... + d^depth] (same dtype)."""
    assert paths.dim() == 3 and depth >= 1
    n, length, d = paths.shape
    assert length >= 2
    inc = paths[:, 1:, :] - paths[:, :-1, :]
    levels = [paths.new_zeros((n,) + (d,) * k) for k in range(1, depth + 1)]
    for step in range(length - 1):
        dx = inc[:, step, :]
        # powers dx^{(x)k}/k!
        expo = [dx]
        for k in range(2, depth + 1):
            expo.append(expo[-1].unsqueeze(-1) *
                        dx.reshape((n,) + (1,) * (k - 1) + (d,)) / k)
        new_levels = []
        for k in range(1, depth + 1):
            acc = levels[k - 1] + expo[k - 1]
            for j in range(1, k):
                left = levels[j - 1]                      # level j
                right = expo[k - j - 1]                   # level k-j
                acc = acc + (left.reshape((n,) + (d,) * j + (1,) * (k - j)) *
                             right.reshape((n,) + (1,) * j + (d,) * (k - j)))
            new_levels.append(acc)
        levels = new_levels
    return torch.cat([lv.reshape(n, -1) for lv in levels], dim=1)


def signature_brute(path, depth):
    """One path [L, d] (array-like) -> fp64 signature, depth <= 3, by
    composing per-segment iterated integrals explicitly (no Chen recursion
    on tensors: closed-form sums over ordered segment tuples)."""
    x = np.asarray(path, dtype=np.float64)
    dl = x[1:] - x[:-1]                    # [L-1, d]
    m, d = dl.shape
    out = [dl.sum(axis=0)]
    if depth >= 2:
        s2 = np.zeros((d, d))
        for a in range(m):
            s2 += np.outer(dl[a], dl[a]) / 2.0
            for b in range(a + 1, m):
                s2 += np.outer(dl[a], dl[b])
        out.append(s2.reshape(-1))
    if depth >= 3:
        s3 = np.zeros((d, d, d))
        for a in range(m):
            s3 += np.einsum('i,j,k->ijk', dl[a], dl[a], dl[a]) / 6.0
            for b in range(a + 1, m):
                s3 += np.einsum('i,j,k->ijk', dl[a], dl[a], dl[b]) / 2.0
                s3 += np.einsum('i,j,k->ijk', dl[a], dl[b], dl[b]) / 2.0
                for c in range(b + 1, m):
                    s3 += np.einsum('i,j,k->ijk', dl[a], dl[b], dl[c])
        out.append(s3.reshape(-1))
    assert depth <= 3
    return np.concatenate(out)
