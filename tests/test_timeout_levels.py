"""Host logic of what a persistent launch's time-out costs a model (MDNN._give_up_a_level): a rank that
stayed resident across the gradient exchange in the calls since the snapshot loses its residency first;
anything else -- no data-parallel group, no resident call since, a second time-out -- the persistent
kernels.  Pure Python, no GPU."""
import warnings

import pytest

import bayes_sim_ig_amd as B


class _FakeDp:
    def __init__(self, calls):
        self.calls, self.mode = calls, None

    def resident_calls(self):
        return self.calls

    def set_resident(self, mode):
        self.mode = mode


class _FakeModel:
    _give_up_a_level = B.MDNN._give_up_a_level

    def __init__(self, dp):
        self._dp, self.disabled = dp, 0

    def _disable_persistent(self):
        self.disabled += 1


def test_a_resident_rank_loses_its_residency_first():
    m = _FakeModel(_FakeDp(calls=7))
    with pytest.warns(RuntimeWarning, match='resident across the gradient exchange timed out'):
        m._give_up_a_level(5)                  # two resident calls since the snapshot
    assert m._dp.mode is False and m.disabled == 0
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m._give_up_a_level(7)                  # no resident call since: the persistent kernels go
    assert m.disabled == 1


def test_without_a_resident_call_the_persistent_kernels_go():
    for dp in (None, _FakeDp(calls=0)):
        m = _FakeModel(dp)
        m._give_up_a_level(0)
        assert m.disabled == 1 and (dp is None or dp.mode is None)
