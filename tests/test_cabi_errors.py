"""C-ABI argument validation and host-side queries (no GPU needed: every
check below returns before the first HIP call).  BSIG_EINVAL is what the
reference would raise as an AssertionError."""
import ctypes as C

import pytest

from bayes_sim_ig_amd import _lib

FAKE = C.c_void_p(4096)      # never dereferenced: validation fails first


@pytest.fixture(scope='module')
def lib():
    return _lib.load()


def err(lib):
    return lib.bsig_last_error().decode()


def test_summary_dim_table(lib):
    assert lib.bsig_summary_dim(0, 21, 3, 1, 0) == 40                  # Pendulum summary_start
    assert lib.bsig_summary_dim(2, 21, 4, 1, 0) == 302                 # Cartpole corrdiff
    assert lib.bsig_summary_dim(2, 51, 60, 8, 0) == 11802              # Ant corrdiff (5 waypoints)
    assert lib.bsig_summary_dim(1, 6, 5, 2, 0) == 6 * 4 * 6 * 2 + 2    # T < 10: all steps
    assert lib.bsig_summary_dim(3, 11, 211, 20, 0) == 232              # ShadowHand: depth 1
    assert lib.bsig_summary_dim(3, 11, 17, 4, 0) == 22 + 22 ** 2 + 22 ** 3
    assert lib.bsig_summary_dim(3, 21, 4, 1, 2) == 6 + 36
    assert lib.bsig_summary_dim(9, 1, 1, 1, 0) == -1


def test_summarizer_argument_errors(lib):
    rc = lib.bsig_crosscorr(FAKE, FAKE, FAKE, 4, 1, 1, 4, 1, 1, 400, None, None)
    assert rc == _lib.BSIG_EINVAL and 'traj_len' in err(lib)            # summarizers.py:94
    rc = lib.bsig_crosscorr(FAKE, FAKE, FAKE, 4, 21, 21, 4, 1, 1, 10, None, None)
    assert rc == _lib.BSIG_EINVAL and 'ld_out' in err(lib)
    rc = lib.bsig_summary_start(None, FAKE, FAKE, 4, 21, 21, 4, 1, 10, 64, None)
    assert rc == _lib.BSIG_EINVAL and 'null' in err(lib)
    rc = lib.bsig_signature(FAKE, FAKE, FAKE, 4, 11, 4, 1, 4, 1 << 20, None)
    assert rc == _lib.BSIG_EINVAL and 'depth' in err(lib)
    rc = lib.bsig_signature(FAKE, FAKE, FAKE, 4, 11, 40, 8, 3, 1 << 20, None)
    assert rc == _lib.BSIG_EUNSUPPORTED and 'depth 3' in err(lib)
    rc = lib.bsig_signature(FAKE, FAKE, FAKE, 4, 1, 4, 1, 2, 1 << 20, None)
    assert rc == _lib.BSIG_EINVAL
    # empty batches are fine and touch nothing
    assert lib.bsig_summary_start(None, None, None, 0, 21, 21, 4, 1, 10, 64, None) == 0
    assert lib.bsig_crosscorr(None, None, None, 0, 21, 21, 4, 1, 1, 304, None, None) == 0


def test_gemm_argument_errors(lib):
    args = dict(lda=8, ldb=8, ldc=8, m=4, n=4, k=8)
    rc = lib.bsig_gemm_f32(None, 8, 0, None, FAKE, 8, 0, None, FAKE, 8, 4, 4, 8, 0, 0, None, None,
                           0, 1.0, None, 0, None)
    assert rc == _lib.BSIG_EINVAL and 'null' in err(lib)
    rc = lib.bsig_gemm_f32(FAKE, 8, 0, None, FAKE, 8, 0, None, FAKE, 8, 4, 4, 8, _lib.EPI_BIAS, 0,
                           None, None, 0, 1.0, None, 0, None)
    assert rc == _lib.BSIG_EINVAL and 'bias' in err(lib)
    rc = lib.bsig_gemm_f32(FAKE, 8, 0, None, FAKE, 8, 0, None, FAKE, 4, 4, 4, 8,
                           _lib.EPI_COS_SIN, 0, None, None, 0, 1.0, None, 0, None)
    assert rc == _lib.BSIG_EINVAL and 'ldc' in err(lib)                 # needs 2N columns
    rc = lib.bsig_gemm_f32(FAKE, 8, 0, None, FAKE, 8, 0, None, FAKE, 8, 4, 4, 8, 77, 0, None, None,
                           0, 1.0, None, 0, None)
    assert rc == _lib.BSIG_EINVAL and 'epilogue' in err(lib)
    assert lib.bsig_gemm_workspace_bytes(100, 260, 4096) >= 100 * 260 * 4
    assert args


def _cfg(input_dim=40, hidden=(24, 24), d=2, k=10, full=False, rff=0):
    cfg = _lib.MdnCfg()
    cfg.input_dim, cfg.n_hidden = input_dim, len(hidden)
    for i, h in enumerate(hidden):
        cfg.hidden[i] = h
    cfg.rff_feats, cfg.rff_scale = rff, 0.1
    cfg.head.out_dim, cfg.head.n_comp, cfg.head.full_cov = d, k, 1 if full else 0
    cfg.head.eps_noise, cfg.head.min_weight, cfg.head.ll_limit = 1e-5, 1e-5, 1e5
    cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps = 1e-3, 0.9, 0.999, 1e-8
    return cfg


def test_parameter_layout(lib):
    cfg = _cfg()
    nh = lib.bsig_head_width(C.byref(cfg.head))
    assert nh == 10 + 2 * 2 * 10
    offs = (C.c_int64 * 12)()
    assert lib.bsig_mdn_param_offsets(C.byref(cfg), offs, 12) == 0
    offs = list(offs)
    # trunk W0 [24,40], b0, W1 [24,24], b1, then the four heads stacked: pi|mu|Diag|Lower
    assert offs[0] == 0 and offs[1] == 24 * 40 and offs[2] == offs[1] + 24
    assert all(o % 4 == 0 for o in offs[:6])                       # 16-byte aligned tensors
    head_w = offs[4]
    assert offs[6] == head_w + 10 * 24 and offs[8] == head_w + 30 * 24   # mu, Diag rows
    assert offs[5] == head_w + nh * 24                              # head biases follow
    assert offs[7] == offs[5] + 10 and offs[9] == offs[5] + 30
    total = lib.bsig_mdn_param_count(C.byref(cfg))
    assert total >= offs[5] + nh and total % 4 == 0
    full = _cfg(d=5, k=3, full=True, hidden=(16,))
    assert lib.bsig_head_width(C.byref(full.head)) == 3 + 2 * 15 + 10 * 3
    assert lib.bsig_mdn_param_offsets(C.byref(cfg), offs if False else (C.c_int64 * 4)(), 4) == _lib.BSIG_EINVAL
    bad = _cfg(rff=64)                                              # MDRFF has no trunk
    assert lib.bsig_mdn_param_count(C.byref(bad)) == -1 and 'trunk' in err(lib)
    odd = _cfg(hidden=(), rff=63)
    assert lib.bsig_mdn_param_count(C.byref(odd)) == -1 and 'even' in err(lib)


def test_fit_plan_argument_errors(lib):
    cfg = _cfg()
    plan = C.c_void_p()
    assert lib.bsig_fit_create(C.byref(cfg), 0, 10, 5, C.byref(plan)) == _lib.BSIG_EINVAL
    assert lib.bsig_fit_create(C.byref(cfg), 100, 200, 100, C.byref(plan)) == 0
    assert lib.bsig_fit_workspace_bytes(plan) > 0
    assert lib.bsig_fit_run(plan, 5, None) == _lib.BSIG_EINVAL and 'not bound' in err(lib)
    fb = _lib.FitBuffers()
    assert lib.bsig_fit_bind(plan, C.byref(fb), 1) == _lib.BSIG_EINVAL and 'null buffer' in err(lib)
    lib.bsig_fit_destroy(plan)
    lib.bsig_fit_destroy(None)                                      # no-op
    # hoisted-RFF plans size their feature block from n_updates
    rff = _cfg(input_dim=2310, hidden=(), d=32, k=4, rff=4096)
    assert lib.bsig_fit_create(C.byref(rff), 100, 200, 100, C.byref(plan)) == 0
    need = lib.bsig_fit_workspace_bytes(plan)
    assert need > (100 * 100 + 6 * 200) * 4096 * 4
    lib.bsig_fit_destroy(plan)


def test_comm_argument_errors_and_external_exchange(lib):
    """bsig_comm_*: validation before any RCCL / HIP call, and a communicator over a
    caller-supplied exchange moving HOST buffers (the entry points only pass pointers on)."""
    handle = C.c_void_p()
    raw = (C.c_ubyte * _lib.COMM_ID_BYTES)()
    assert lib.bsig_comm_unique_id(None) == _lib.BSIG_EINVAL
    rc = lib.bsig_comm_init(None, 2, 0, 0, C.byref(handle))
    assert rc == _lib.BSIG_EINVAL and 'unique id' in err(lib)
    rc = lib.bsig_comm_init(raw, 2, 2, 0, C.byref(handle))
    assert rc == _lib.BSIG_EINVAL and 'rank 2 of world 2' in err(lib)
    rc = lib.bsig_comm_init(raw, 0, 0, 0, C.byref(handle))
    assert rc == _lib.BSIG_EINVAL
    assert lib.bsig_comm_init(raw, 1, 0, 0, None) == _lib.BSIG_EINVAL
    rc = lib.bsig_comm_init_external(2, 0, None, None, C.byref(handle))
    assert rc == _lib.BSIG_EINVAL and 'exchange' in err(lib)
    assert lib.bsig_comm_allreduce(None, FAKE, 4, None) == _lib.BSIG_EINVAL
    assert lib.bsig_comm_world(None) == 0 and lib.bsig_comm_rank(None) == -1
    lib.bsig_comm_destroy(None)                              # no-op
    assert lib.bsig_fit_run_dp(None, None, 1, None, None) == _lib.BSIG_EINVAL

    calls = []

    def exchange(ctx, op, buf, n, root, stream):
        arr = (C.c_float * n).from_address(buf)
        calls.append((op, n, root))
        if op == _lib.EXCHANGE_SUM:
            for i in range(n):
                arr[i] *= 3.0                               # "three identical ranks"
        return 0 if n != 7 else 5

    fn = _lib.EXCHANGE_FN(exchange)
    assert lib.bsig_comm_init_external(3, 1, C.cast(fn, C.c_void_p), None, C.byref(handle)) == 0
    assert lib.bsig_comm_world(handle) == 3 and lib.bsig_comm_rank(handle) == 1
    data = (C.c_float * 4)(1.0, 2.0, 3.0, 4.0)
    assert lib.bsig_comm_allreduce(handle, data, 4, None) == 0
    assert list(data) == [3.0, 6.0, 9.0, 12.0]
    assert lib.bsig_comm_broadcast(handle, data, 4, 2, None) == 0
    assert lib.bsig_comm_broadcast(handle, data, 4, 3, None) == _lib.BSIG_EINVAL   # root outside the group
    assert lib.bsig_comm_allreduce(handle, data, 0, None) == 0                    # empty: no call
    assert calls == [(_lib.EXCHANGE_SUM, 4, 0), (_lib.EXCHANGE_BROADCAST, 4, 2)]
    seven = (C.c_float * 7)()
    rc = lib.bsig_comm_allreduce(handle, seven, 7, None)                          # exchange reports failure
    assert rc == _lib.BSIG_ELAUNCH and 'external exchange failed (5)' in err(lib)
    lib.bsig_comm_destroy(handle)


def test_comm_unique_ids_are_fresh(lib):
    a, b = (C.c_ubyte * _lib.COMM_ID_BYTES)(), (C.c_ubyte * _lib.COMM_ID_BYTES)()
    rc = lib.bsig_comm_unique_id(a)
    if rc == _lib.BSIG_EUNSUPPORTED:
        pytest.skip('no RCCL on this host: ' + err(lib))
    assert rc == 0 and lib.bsig_comm_unique_id(b) == 0
    assert bytes(a) != bytes(b) and any(bytes(a))
