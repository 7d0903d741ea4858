"""Mixture Density NN estimator on MI355X — mirror of the reference's
bayes_sim_ig/models/mdnn.py (class MDNN: same constructor keywords, methods,
attributes, ``state_dict`` keys and error behaviour).

All arithmetic runs in libbsig_hip (csrc/): the parameters live in ONE flat
fp32 device buffer (the named ``nn.Parameter``s are views into it, so
``state_dict`` / ``load_state_dict`` keep working), the trunk / head products
are fp32-MFMA GEMMs, the mixture head + NLL + backward is one fused kernel,
Adam runs over the flat buffer, and ``run_training`` replays the whole update
from a HIP graph (csrc/estimator.hip).  There is no CPU fallback.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import pdf
from . import dp as _dp
from .summarizers import CrossCorrFactors

_ACT_CODES = {nn.Tanh: _lib.ACT_TANH, nn.ReLU: _lib.ACT_RELU,
              nn.LeakyReLU: _lib.ACT_LEAKY_RELU, nn.Sigmoid: _lib.ACT_SIGMOID,
              nn.Identity: _lib.ACT_IDENTITY}


def _on_model_device(fn):
    """Run a method with the model's GPU as the current HIP device and its torch stream
    as the launch stream (a model built with device='cuda:1' must not launch on cuda:0)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        flat = getattr(self, '_flat', None)
        if flat is None or not flat.is_cuda or flat.device.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with _lib.on_device(flat.device):
            return fn(self, *args, **kwargs)
    return wrapped


class PersistentTimeout(RuntimeError):
    """A bounded cross-workgroup poll of a persistent update kernel gave up: its workgroups (one
    per CU) were not all resident -- the GPU is shared or partitioned.  The call's parameters and
    Adam moments are partially updated; MDNN.run_training / BayesSim.fit catch this, restore the
    state they saved at the start and repeat the work on the per-phase kernels."""


class PendingLogs:
    """The 6+6 losses and the non-finite flag of one run_training call, still on
    the device; ``result()`` is the call's single host read-back.  BayesSim.fit
    defers it to the end of the chunk loop so the GPU never waits for the host."""

    def __init__(self, packed, n_e, n_test, verbose, dp=None):
        self.packed, self.n_e, self.n_test, self.verbose = packed, n_e, n_test, verbose
        self.dp = dp      # (eval_its, n_updates, world): `packed` is bsig_fit_run_dp's reduced_logs

    def result(self):
        host = self.packed.cpu().tolist()
        n_e = self.n_e
        if self.dp is not None:
            eval_its, n_updates, world = self.dp
            train_list = [host[it] / world for it in eval_its]
            n_test_all = host[n_updates + n_e]
            test_list = [v / max(n_test_all, 1.0) for v in host[n_updates:n_updates + n_e]]
            bad = (1 if host[n_updates + n_e + 1] > 0 else 0) | (2 if host[n_updates + n_e + 2] > 0 else 0)
            self.n_test = int(n_test_all)
        else:
            train_list, test_list, bad = host[:n_e], host[n_e:2 * n_e], host[2 * n_e]
        if int(bad) & 2:      # a bounded cross-workgroup poll of the persistent kernel gave up
            raise PersistentTimeout('persistent update kernel timed out waiting for another workgroup: its '
                                    'workgroups (one per CU) were not all resident -- is the GPU shared with '
                                    'another process or partitioned?  The parameters and Adam moments of this '
                                    'call are partially updated; set BSIG_NO_PERSISTENT=1 to use the '
                                    'per-phase kernels')
        assert bad == 0, 'non-finite value in forward / loss (mdnn.py:120-124,162-174)'
        if self.n_test == 0:
            test_list = [float('nan')] * n_e   # mean over an empty test split
        if self.verbose:
            for a, b in zip(train_list, test_list):
                print(f'loss: train {a:0.4f} test {b:0.4f}')
        return {'train_loss': train_list, 'test_loss': test_list}


class _FusedNLL(torch.autograd.Function):
    """Autograd node behind ``mdn_loss_fn(*model(x), y)``: the value is the
    loss already computed; backward runs the fused HIP forward+NLL+backward
    (bsig_mdn_loss_grad) into a scratch flat buffer and hands autograd one
    gradient per parameter, so ``loss.backward(); optimizer.step()`` behaves as
    in the reference (mdnn.py:231-234)."""

    @staticmethod
    def forward(ctx, loss_value, model, x, y, noise, seed, *params):
        ctx.model, ctx.x, ctx.y, ctx.noise, ctx.seed = model, x, y, noise, seed
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        tmp = torch.empty_like(m._flat_grad)
        m.loss_and_grad(ctx.x, ctx.y, noise=ctx.noise, seed=ctx.seed, grads_out=tmp)
        grads = [tmp[o:o + int(np.prod(full))].view(full)[tuple(slice(0, d) for d in shape)] * grad_out
                 for o, full, shape in m._param_slices]
        return (None, None, None, None, None, None, *grads)


class MDNN(nn.Module):
    LL_LIMIT = 1.0e5     # limit log likelihood to avoid large gradients
    MIN_WEIGHT = 1.0e-5  # minimum component weights to enable updates
    EPS_NOISE = 1.e-5    # small noise e.g. for numerical stability
    VERBOSE = True       # print the 6 train/test losses per call like the reference
    USE_GRAPH = True     # replay the update from a HIP graph
    PAD_TRUNK_TO = 128   # hidden width the persistent update kernel is built for

    def __init__(self, input_dim, output_dim, output_lows, output_highs,
                 n_gaussians, full_covariance, hidden_layers, activation, lr,
                 device='cpu', **kwargs):
        """Same arguments as the reference (mdnn.py:26-52).  ``device`` must
        name a GPU for anything beyond construction."""
        super(MDNN, self).__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.output_lows = None
        self.output_highs = None
        if output_lows is not None:
            self.output_lows = torch.from_numpy(np.asarray(output_lows)).float().to(device)
            self.output_highs = torch.from_numpy(np.asarray(output_highs)).float().to(device)
        self.n_gaussians = n_gaussians
        self.activation = activation
        self.lr = lr
        self.device = device
        if activation not in _ACT_CODES:
            raise NotImplementedError('activation %r has no HIP epilogue' % (activation,))
        # Same submodules, created in the same order on the CPU so that the
        # torch-RNG initialisation matches the reference bit for bit
        # (mdnn.py:68-86), then flattened onto the device.
        net = OrderedDict()
        width = input_dim
        for l, layer_size in enumerate(hidden_layers):
            net['fcon%d' % l] = nn.Linear(width, layer_size)
            net['nl%d' % l] = activation()
            width = layer_size
        self.net = nn.Sequential(net) if len(hidden_layers) > 0 else None
        self.pi = nn.Linear(width, n_gaussians)
        self.mu = nn.Linear(width, output_dim * n_gaussians)
        self.Diag = nn.Sequential(nn.Linear(width, output_dim * n_gaussians))
        self.Lower = None
        self.L_size = int(0.5 * output_dim * (output_dim - 1))
        if self.L_size > 0 and full_covariance:
            self.Lower = nn.Linear(width, self.L_size * n_gaussians)
        self._hidden = [int(h) for h in hidden_layers]
        # A two-layer tanh trunk narrower than 128 is STORED zero-padded to [128, 128] (the
        # parameters are views of the leading blocks): the padding units see zero weights, put
        # out tanh(0) = 0 and receive exactly zero gradients, so Adam leaves them at zero and
        # the network is the reference's bit for bit -- and the persistent update kernel, which
        # is built for the reference's default [128, 128] trunk, covers it (the reference's own
        # tests/regression_tests.py:59 uses (24, 24)).
        self._hidden_stored = list(self._hidden)
        if (len(self._hidden) == 2 and max(self._hidden) <= self.PAD_TRUNK_TO and
                min(self._hidden) >= 1 and self._hidden != [self.PAD_TRUNK_TO] * 2 and
                activation is nn.Tanh and os.environ.get('BSIG_NO_TRUNK_PAD') != '1'):
            self._hidden_stored = [self.PAD_TRUNK_TO] * 2
        self._rff_feats = int(kwargs.get('_rff_feats', 0))
        self._rff_scale = float(kwargs.get('_rff_scale', 0.0))
        self._plan = None
        self._plan_key = None
        self._bufs = {}
        self._dp = None
        self._flatten(device)

    # ------------------------------------------------------------ plumbing
    def _cfg(self):
        cfg = _lib.MdnCfg()
        cfg.input_dim = int(self.input_dim)
        cfg.n_hidden = len(self._hidden)
        for i, h in enumerate(self._hidden_stored):
            cfg.hidden[i] = h
        cfg.activation = _ACT_CODES[self.activation]
        cfg.rff_feats = self._rff_feats
        cfg.rff_cos_only = 0
        cfg.rff_scale = self._rff_scale
        cfg.head.out_dim = int(self.output_dim)
        cfg.head.n_comp = int(self.n_gaussians)
        cfg.head.full_cov = 1 if self.Lower is not None else 0
        cfg.head.eps_noise = float(type(self).EPS_NOISE)
        cfg.head.min_weight = float(type(self).MIN_WEIGHT)
        cfg.head.ll_limit = float(type(self).LL_LIMIT)
        cfg.lr, cfg.beta1, cfg.beta2, cfg.adam_eps = float(self.lr), 0.9, 0.999, 1e-8
        return cfg

    def _linears(self):
        mods = [self.net[2 * l] for l in range(len(self._hidden))]
        mods += [self.pi, self.mu, self.Diag[0]]
        if self.Lower is not None:
            mods.append(self.Lower)
        return mods

    def _flatten(self, device):
        """Move every parameter into one flat fp32 buffer (layout:
        bsig_mdn_param_offsets) and re-point the nn.Parameters at views."""
        lib = _lib.load()
        cfg = self._cfg()
        total = int(lib.bsig_mdn_param_count(C.byref(cfg)))
        assert total > 0, lib.bsig_last_error().decode()
        n_off = 2 * (len(self._hidden) + 4)
        offs = (C.c_int64 * n_off)()
        _lib.check(lib.bsig_mdn_param_offsets(C.byref(cfg), offs, n_off))
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grads = torch.zeros(total, dtype=torch.float32, device=device)
        self._param_slices = []
        # stored input width of every Linear: the (possibly padded) width of the layer below
        widths = [self.input_dim if self._rff_feats == 0 else self._rff_feats] + self._hidden_stored
        with torch.no_grad():
            for i, mod in enumerate(self._linears()):
                l = min(i, len(self._hidden))            # trunk layer l, or the heads (on the last width)
                rows = self._hidden_stored[i] if i < len(self._hidden) else mod.weight.shape[0]
                for j, prm in enumerate((mod.weight, mod.bias)):
                    o = int(offs[2 * i + j])
                    full = (rows, widths[l]) if j == 0 else (rows,)
                    real = tuple(prm.shape)
                    sub = tuple(slice(0, d) for d in real)
                    self._param_slices.append((o, full, real))
                    view = flat[o:o + int(np.prod(full))].view(full)[sub]
                    view.copy_(prm.detach().to(dtype=torch.float32))
                    prm.data = view
                    prm.grad = grads[o:o + int(np.prod(full))].view(full)[sub]
        self._flat, self._flat_grad = flat, grads
        self._exp_avg = torch.zeros_like(flat)
        self._exp_avg_sq = torch.zeros_like(flat)
        if self.output_lows is not None:
            self.output_lows = self.output_lows.to(device)
            self.output_highs = self.output_highs.to(device)
        self.device = str(device) if not isinstance(device, str) else device
        self._drop_plan()

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        prm = next(self.parameters())
        if prm.dtype != torch.float32:
            raise NotImplementedError('the HIP estimator computes in fp32 only')
        self._flatten(prm.device)
        return self

    def _drop_plan(self):
        if getattr(self, '_plan', None):
            _lib.load().bsig_fit_destroy(self._plan)
        self._plan, self._plan_key = None, None
        self._bufs = {}

    def __del__(self):
        try:
            self._drop_plan()
        except Exception:
            pass

    # ---- robustness of the persistent launches: every workgroup of such a launch must be resident
    #      at once; when the GPU turns out to be shared, the bounded polls give up and the call is
    #      repeated from a snapshot on the per-phase kernels
    def _snapshot(self):
        """(parameters, Adam moments, numpy / torch RNG states) as of now."""
        return (self._flat.clone(), self._exp_avg.clone(), self._exp_avg_sq.clone(),
                np.random.get_state(), torch.get_rng_state())

    def _restore(self, snap):
        self._flat.copy_(snap[0]); self._exp_avg.copy_(snap[1]); self._exp_avg_sq.copy_(snap[2])
        np.random.set_state(snap[3]); torch.set_rng_state(snap[4])

    def _resident_calls(self):
        return self._dp.resident_calls() if self._dp is not None else 0

    def _give_up_a_level(self, resident_calls_before):
        """After a persistent launch timed out.  A data-parallel rank that stayed resident across the
        gradient exchange in the calls since ``resident_calls_before`` first goes back to one launch per
        update (the exchange stream was not served in time: the launch itself had the chip); a time-out
        after that, or without the resident exchange: the per-phase kernels."""
        if self._dp is not None and self._dp.resident_calls() > resident_calls_before:
            import warnings
            warnings.warn('bayes_sim_ig_amd: a data-parallel launch that stays resident across the gradient '
                          'exchange timed out waiting for the exchange; this model continues with one launch '
                          'per update', RuntimeWarning, stacklevel=3)
            self._dp.set_resident(False)
            return
        self._disable_persistent()

    def _disable_persistent(self):
        """From now on this model's plans use the per-phase kernels."""
        import warnings
        warnings.warn('bayes_sim_ig_amd: a persistent update launch timed out waiting for another workgroup '
                      '(GPU shared with another process?); this model continues on the per-phase kernels, '
                      'which are slower', RuntimeWarning, stacklevel=3)
        self._no_persistent = True
        self._drop_plan()

    def _gpu(self):
        lib = _lib.require_gpu()
        if not self._flat.is_cuda:
            raise RuntimeError("MDNN was built with device=%r: the estimator runs on "
                               "MI355X only, construct it with device='cuda:N' "
                               "(no CPU fallback)" % (self.device,))
        return lib

    def _buf(self, name, numel, dtype=torch.float32):
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self._flat.device)
            self._bufs[name] = t
        return t

    def _rff_args(self):
        return None, 0, None   # (coeff, ld_coeff, offset); MDRFF overrides

    def _seed(self):
        """A fresh Philox seed drawn from torch's global RNG (the reference
        consumes the torch RNG for rand_like, mdnn.py:116)."""
        if type(self).EPS_NOISE == 0.0:
            return 0
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    # ----------------------------------------------------------- forward
    @_on_model_device
    def _head_forward(self, x):
        lib = self._gpu()
        cfg = self._cfg()
        xs, ldx = _lib.as_f32_rows(x, self._flat.device)
        assert xs.shape[1] == self.input_dim
        b = xs.shape[0]
        nh = int(lib.bsig_head_width(C.byref(cfg.head)))
        out = torch.empty((b, nh), dtype=torch.float32, device=xs.device)
        ws_bytes = int(lib.bsig_mdn_workspace_bytes(C.byref(cfg), b))
        ws = self._buf('fwd_ws', ws_bytes // 4 + 1)
        coeff, ldc, off = self._rff_args()
        _lib.check(lib.bsig_mdn_head_forward(
            C.byref(cfg), _lib.ptr(self._flat), _lib.ptr(coeff), ldc, _lib.ptr(off),
            _lib.ptr(xs), ldx, None, b, _lib.ptr(out), nh, _lib.ptr(ws),
            ws.numel() * 4, _lib.stream()))
        return cfg, out

    @_on_model_device
    def forward(self, x, noise=None):
        """Reference mdnn.py:89-125 -> (weights[B,K], mu[B,D,K], L_d[B,D,K],
        L[B,L_size,K] | None).  ``noise`` injects the rand_like draw."""
        lib = self._gpu()
        cfg, out = self._head_forward(x)
        b, d, k = out.shape[0], self.output_dim, self.n_gaussians
        dev = out.device
        weights = torch.empty((b, k), dtype=torch.float32, device=dev)
        mu = torch.empty((b, d, k), dtype=torch.float32, device=dev)
        l_d = torch.empty((b, d, k), dtype=torch.float32, device=dev)
        low = None
        if self.Lower is not None:
            low = torch.empty((b, self.L_size, k), dtype=torch.float32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = self._buf('head_ws', 64 + int(lib.bsig_head_workspace_bytes(C.byref(cfg.head), b)) // 4)
        nz = None if noise is None else noise.to(dev, torch.float32).contiguous()
        seed = self._seed()
        _lib.check(lib.bsig_mdn_head_outputs(
            C.byref(cfg.head), _lib.ptr(out), out.stride(0), b, _lib.ptr(nz),
            seed, 0, _lib.ptr(weights), _lib.ptr(mu), _lib.ptr(l_d),
            _lib.ptr(low), _lib.ptr(flag), _lib.ptr(ws), ws.numel() * 4, _lib.stream()))
        assert int(flag.item()) == 0      # isfinite asserts, mdnn.py:120-124
        # remembered so that mdn_loss_fn(*model(x), y).backward() works (below)
        self._fwd_ctx = (weights, x, nz, seed) if torch.is_grad_enabled() else None
        return weights, mu, l_d, low

    @_on_model_device
    def mdn_loss_fn(self, weights, mu, L_d, L, y):
        """Reference mdnn.py:127-178 -> 0-dim loss tensor."""
        lib = self._gpu()
        cfg = self._cfg()
        dev = self._flat.device
        b = y.size()[0]
        ys, ldy = _lib.as_f32_rows(y, dev)
        w = weights.to(dev, torch.float32).contiguous()
        m = mu.to(dev, torch.float32).contiguous()
        s = L_d.to(dev, torch.float32).contiguous()
        lo = None if L is None else L.to(dev, torch.float32).contiguous()
        head = cfg.head
        head.full_cov = 0 if lo is None else 1
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = self._buf('head_ws', 64 + int(lib.bsig_head_workspace_bytes(C.byref(head), b)) // 4)
        _lib.check(lib.bsig_mdn_nll_from_tuple(
            C.byref(head), _lib.ptr(w), _lib.ptr(m), _lib.ptr(s), _lib.ptr(lo),
            _lib.ptr(ys), ldy, b, _lib.ptr(loss), _lib.ptr(flag), _lib.ptr(ws),
            ws.numel() * 4, _lib.stream()))
        assert int(flag.item()) == 0      # mdnn.py:172-174
        ctx = getattr(self, '_fwd_ctx', None)
        if torch.is_grad_enabled() and ctx is not None and ctx[0] is weights:
            # the reference pattern loss = mdn_loss_fn(*model(x), y); loss.backward():
            # the backward is the fused forward+NLL+backward pass on the same x, y, noise
            return _FusedNLL.apply(loss[0], self, ctx[1], ys, ctx[2], ctx[3], *self.parameters())
        return loss[0]

    @_on_model_device
    def loss_and_grad(self, x, y, rows=None, noise=None, norm_batch=None, seed=None,
                      grads_out=None):
        """forward + mdn_loss_fn + backward for one minibatch
        (mdnn.py:229-233): returns the 0-dim loss; gradients land in every
        parameter's ``.grad`` (views of the flat gradient buffer).  ``y`` is
        already normalised."""
        lib = self._gpu()
        cfg = self._cfg()
        dev = self._flat.device
        xs, ldx = _lib.as_f32_rows(x, dev)
        ys, ldy = _lib.as_f32_rows(y, dev)
        ridx = None
        b = xs.shape[0]
        if rows is not None:
            ridx = torch.as_tensor(rows, dtype=torch.int32, device=dev).contiguous()
            b = ridx.numel()
        ws_bytes = int(lib.bsig_mdn_workspace_bytes(C.byref(cfg), b))
        ws = self._buf('grad_ws', ws_bytes // 4 + 1)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        nz = None if noise is None else noise.to(dev, torch.float32).contiguous()
        coeff, ldc, off = self._rff_args()
        _lib.check(lib.bsig_mdn_loss_grad(
            C.byref(cfg), _lib.ptr(self._flat), _lib.ptr(coeff), ldc, _lib.ptr(off),
            _lib.ptr(xs), ldx, _lib.ptr(ys), ldy, _lib.ptr(ridx), b,
            int(norm_batch or b), _lib.ptr(nz), self._seed() if seed is None else int(seed), 0,
            _lib.ptr(self._flat_grad if grads_out is None else grads_out), _lib.ptr(loss),
            _lib.ptr(flag), _lib.ptr(ws),
            ws.numel() * 4, _lib.stream()))
        assert int(flag.item()) == 0
        return loss[0]

    @_on_model_device
    def adam_step(self, t):
        """One torch.optim.Adam step (defaults) over the flat buffers; t is
        the 1-based step number since the optimizer was created."""
        lib = self._gpu()
        _lib.check(lib.bsig_adam_flat(
            _lib.ptr(self._flat), _lib.ptr(self._flat_grad), _lib.ptr(self._exp_avg),
            _lib.ptr(self._exp_avg_sq), self._flat.numel(), float(self.lr), 0.9, 0.999,
            1e-8, int(t), _lib.stream()))

    # ---------------------------------------------------------- training
    def enable_data_parallel(self, group=None, transport=None):
        """Shard run_training's minibatch over the ranks of ``group``: every
        rank keeps a full replica, computes the gradient of its B/R rows and
        the flat gradient buffer is all-reduced (RCCL) before the identical
        Adam step.  Parameters are broadcast from rank 0.
        The exchange is the C ABI's communicator (bsig_comm_*: RCCL, or with
        ``transport='torch'`` the group's own collectives behind the same entry
        points) and run_training's update loop is bsig_fit_run_dp."""
        self._dp = _dp.DataParallel(group)
        if self._flat.is_cuda:
            with _lib.on_device(self._flat.device):
                self._dp.init_comm(self._flat.device, transport)
        self._dp.broadcast(self._flat)
        self._dp_sync_extra()
        return self

    def _dp_sync_extra(self):
        """Replica state outside the flat parameter buffer (MDRFF: the RFF frequencies)."""

    def run_training(self, x_data, y_data, n_updates, batch_size, test_frac=0.2,
                     ids_table=None, _defer=False, _feats=None):
        """Reference mdnn.py:180-243.  Returns {'train_loss': [...],
        'test_loss': [...]} with the same 6 logging points.  ``ids_table``
        [n_updates, batch] (optional) overrides the numpy-RNG minibatch draw
        (teacher forcing for parity tests)."""
        if _defer or not self._flat.is_cuda:
            # (deferred logs: the caller -- BayesSim.fit -- holds the snapshot and repeats its loop)
            return self._run_training_once(x_data, y_data, n_updates, batch_size, test_frac,
                                           ids_table, _defer, _feats)
        snap = self._snapshot() if self._may_time_out() else None
        for attempt in range(3):
            calls0 = self._resident_calls()
            try:
                return self._run_training_once(x_data, y_data, n_updates, batch_size, test_frac,
                                               ids_table, False, _feats)
            except PersistentTimeout:
                # (a data-parallel rank: the flag is the SUM over the ranks of the call's logs -- every
                # rank of the group arrives here in the same call and repeats it with its peers)
                if snap is None or attempt == 2:
                    raise
                self._restore(snap)
                self._give_up_a_level(calls0)

    def _may_time_out(self):
        """Could the next call run a persistent kernel?  (Unknown before the first plan exists.)"""
        if getattr(self, '_no_persistent', False) or os.environ.get('BSIG_NO_PERSISTENT') == '1':
            return False
        return self._plan is None or bool(_lib.load().bsig_fit_is_persistent(self._plan))

    @_on_model_device
    def _run_training_once(self, x_data, y_data, n_updates, batch_size, test_frac=0.2,
                           ids_table=None, _defer=False, _feats=None):
        assert x_data.shape[0] == y_data.shape[0]
        lib = self._gpu()
        self.train()
        cfg = self._cfg()
        dev = self._flat.device
        d, n_tot = self.output_dim, x_data.shape[0]
        n_train = max(int(n_tot * (1.0 - test_frac)), 1)
        n_test = n_tot - n_train
        st = _lib.stream()
        # plan (graphs, persistent-kernel geometry) keyed by everything baked into it
        key = (batch_size, max(n_test, self._bufs.get('cap_test', 0)), n_updates,
               cfg.head.eps_noise, cfg.lr, cfg.head.min_weight, cfg.head.ll_limit,
               max(n_train, self._bufs.get('cap_train', 0)))
        if self._plan is None or self._plan_key != key:
            if self._plan:
                lib.bsig_fit_destroy(self._plan)
            handle = C.c_void_p()
            # (a model that met a persistent-launch time-out stays on the per-phase kernels: a plan
            # option, not the process-wide environment switch)
            flags = _lib.PLAN_NO_PERSISTENT if getattr(self, '_no_persistent', False) else 0
            _lib.check(lib.bsig_fit_create_ex(C.byref(cfg), batch_size, key[7], key[1],
                                              n_updates, flags, C.byref(handle)))
            self._plan, self._plan_key = handle, key
            self._bufs['cap_test'], self._bufs['cap_train'] = key[1], key[7]
        # a cross-correlation summary may arrive as factor rows (summarizers.CrossCorrFactors):
        # plans whose first layer lives in the persistent kernel consume them as they are
        factored = isinstance(x_data, CrossCorrFactors)
        if factored and not lib.bsig_fit_accepts_factor_rows(self._plan, x_data.s_dim, x_data.a_dim):
            x_data, factored = x_data.materialize(), False
        ldy = _lib.round_up(d, 4)
        ys, ldy_src = _lib.as_f32_rows(y_data, dev)
        assert x_data.shape[1] == self.input_dim and ys.shape[1] == d
        cap = self._bufs.get('cap_rows', 0)
        if n_tot > cap:
            self._bufs.pop('x_stage', None), self._bufs.pop('y_stage', None)
            self._bufs['cap_rows'] = n_tot
        if factored:
            # bound where they lie: 1.3 KB per Ant row instead of a 47 KB summary row
            x_stage, _ = _lib.as_f32_rows(x_data.factors, dev)
            ldx = x_stage.stride(0) if n_tot > 1 else x_stage.shape[1]
            # the held-out rows, read once per evaluation, as summary rows -- unless the launch
            # evaluates from the held-out pairs' factor rows too (a streamed first layer): then no
            # [n, I] block exists at all
            bind_flags = (_lib.FIT_GRAPH if type(self).USE_GRAPH else 0) | \
                (_lib.FIT_SPLIT_ADAM if self._dp is not None else 0)
            eval_fac = bool(lib.bsig_fit_evaluates_from_factors(self._plan, x_data.s_dim, x_data.a_dim, bind_flags))
            x_held = x_data[n_train:].materialize() if n_test > 0 and not eval_fac else None
            self._bufs['x_keepalive'] = (x_stage, x_held)
        else:
            # chunk staging: fixed addresses (graph replay) and 16-B aligned rows
            xs, ldx_src = _lib.as_f32_rows(x_data, dev)
            ldx = _lib.round_up(self.input_dim, 4)
            x_stage = self._buf('x_stage', self._bufs['cap_rows'] * ldx)
            # (an MDRFF whose rows' features are handed over never reads the summaries themselves)
            if _feats is None or not lib.bsig_fit_takes_features(self._plan, n_train):
                _lib.check(lib.bsig_copy_rows(_lib.ptr(xs), ldx_src, None, _lib.ptr(x_stage), ldx,
                                              n_tot, self.input_dim, st))
        y_stage = self._buf('y_stage', self._bufs['cap_rows'] * ldy)
        if self.output_lows is not None:                       # mdnn.py:204-205
            _lib.check(lib.bsig_normalize_rows(
                _lib.ptr(ys), ldy_src, _lib.ptr(self.output_lows),
                _lib.ptr(self.output_highs), _lib.ptr(y_stage), ldy, n_tot, d, st))
        else:
            _lib.check(lib.bsig_copy_rows(_lib.ptr(ys), ldy_src, None, _lib.ptr(y_stage),
                                          ldy, n_tot, d, st))
        n_ids = n_updates * batch_size
        ids_dev = self._buf('ids', max(n_ids, 1), torch.int32)
        def upload_ids():
            """Minibatch ids in the reference's numpy-RNG order (mdnn.py:219-222), drawn on the host
            and uploaded through a pinned staging ring (a pageable source would make the copy wait
            for the stream to drain)."""
            if ids_table is None:
                # (dtype=int32: the SAME draws from the same stream as the reference's int64 default --
                # tests/test_ids_draw.py -- in half the host time and without the astype pass: at 122 x
                # 8192 ids the draw is what the GPU waits for when the host is busy)
                ids_np = np.random.randint(0, n_train, (n_updates, batch_size), dtype=np.int32)
            else:
                ids_np = np.asarray(ids_table)
                assert ids_np.shape == (n_updates, batch_size)
            # truly asynchronous upload: pinned staging ring (a pageable source would
            # make the copy wait for the stream to drain)
            ring = self._bufs.setdefault('ids_ring', {'slots': [], 'next': 0})
            if not ring['slots'] or ring['slots'][0][0].numel() < n_ids:
                ring['slots'] = [[torch.empty(max(n_ids, 1), dtype=torch.int32, pin_memory=True), None]
                                 for _ in range(4)]
            slot = ring['slots'][ring['next'] % 4]
            ring['next'] += 1
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0][:n_ids].copy_(torch.from_numpy(np.ascontiguousarray(ids_np, dtype=np.int32)).reshape(-1))
            # (a copy KERNEL reading the pinned buffer across PCIe instead of this DMA copy was measured in
            # round 5: 487.5 k against 489.0 k pairs/s -- the ~20 us "gap before the next fit_begin_kernel"
            # of the chunk timeline is not the copy engine's hand-off)
            ids_dev[:n_ids].copy_(slot[0][:n_ids], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()

        # A large table (the scaled-batch fit: 122 x 8192 ids, 8 ms of numpy) is drawn AFTER the
        # begin call has been enqueued, so that begin's device work -- the RFF projection of the
        # chunk's rows, 7 ms at 100k rows -- runs under it.  Only where begin does not read the table
        # (a plan that projects the gathered minibatch rows instead of each row once does).
        late_ids = n_ids >= (1 << 16) and (cfg.rff_feats == 0 or
                                           bool(lib.bsig_fit_takes_features(self._plan, n_train)))
        if not late_ids:
            upload_ids()
        every = max(n_updates // 5, 1)
        eval_its = [it for it in range(n_updates)
                    if it % every == 0 or it + 1 == n_updates]
        train_loss = self._buf('train_loss', n_updates)
        test_loss = self._buf('test_loss', len(eval_its))
        state = self._buf('state', 16, torch.int32)
        ws = self._buf('fit_ws', int(lib.bsig_fit_workspace_bytes(self._plan)) // 4 + 1)
        coeff, ldc, off = self._rff_args()
        fb = _lib.FitBuffers()
        fb.params, fb.grads = self._flat.data_ptr(), self._flat_grad.data_ptr()
        fb.exp_avg, fb.exp_avg_sq = self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr()
        fb.rff_coeff = None if coeff is None else coeff.data_ptr()
        fb.ld_coeff = ldc
        fb.rff_offset = None if off is None else off.data_ptr()
        fb.x_train, fb.ldx_train, fb.n_train = x_stage.data_ptr(), ldx, n_train
        fb.y_train, fb.ldy_train = y_stage.data_ptr(), ldy
        fb.x_test, fb.ldx_test, fb.n_test = x_stage.data_ptr() + 4 * n_train * ldx, ldx, n_test
        if factored and n_test > 0:
            if x_held is not None:
                fb.x_test, fb.ldx_test = x_held.data_ptr(), x_held.stride(0)
            else:
                fb.x_test, fb.ldx_test = None, 0
        fb.y_test, fb.ldy_test = y_stage.data_ptr() + 4 * n_train * ldy, ldy
        fb.ids_table = ids_dev.data_ptr()
        fb.train_loss, fb.test_loss = train_loss.data_ptr(), test_loss.data_ptr()
        fb.state = state.data_ptr()
        fb.workspace, fb.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if factored:
            fb.x_kind, fb.x_s, fb.x_a = _lib.X_CROSSCORR_FACTORS, x_data.s_dim, x_data.a_dim
            if n_test > 0:      # the held-out pairs' factor rows lie behind the training rows
                fb.x_test_factors, fb.ldx_test_factors = x_stage.data_ptr() + 4 * n_train * ldx, ldx
        flags = (_lib.FIT_GRAPH if type(self).USE_GRAPH else 0) | \
            (_lib.FIT_SPLIT_ADAM if self._dp is not None else 0)
        _lib.check(lib.bsig_fit_bind(self._plan, C.byref(fb), flags))
        if _feats is not None:
            # MDRFF: the rows' RFF features, already projected by the caller (BayesSim.fit)
            assert _feats.shape[0] == n_tot and _feats.is_cuda and _feats.dtype == torch.float32
            # (a plan without a per-row feature cache declines: bsig_fit_begin then projects)
            lib.bsig_fit_set_features(self._plan, _lib.ptr(_feats), _feats.stride(0), n_tot, st)
        world = 1 if self._dp is None else self._dp.world
        _lib.check(lib.bsig_fit_begin(self._plan, self._seed(), batch_size * world, st))
        if late_ids:
            upload_ids()
        if self._dp is None:
            _lib.check(lib.bsig_fit_run(self._plan, n_updates, st))
            # single read-back per call: 6+6 losses and the isfinite flag
            # (a fresh allocation per call -- no kernel: BayesSim.fit reads all chunks' logs at the end)
            packed = torch.empty(2 * len(eval_its) + 1, dtype=torch.float32, device=dev)
            _lib.check(lib.bsig_fit_pack_logs(self._plan, n_updates, _lib.ptr(packed), st))
            pending = PendingLogs(packed, len(eval_its), n_test, type(self).VERBOSE)
        else:
            # data parallel: grad -> all-reduce -> apply per update, driven from C
            self._dp._on_gpu = True
            logs = torch.empty(n_updates + len(eval_its) + 3, dtype=torch.float32, device=dev)
            _lib.check(lib.bsig_fit_run_dp(self._plan, self._dp.comm, n_updates, _lib.ptr(logs), st))
            if self._dp._error is not None:
                err, self._dp._error = self._dp._error, None
                raise err
            pending = PendingLogs(logs, len(eval_its), n_test, type(self).VERBOSE,
                                  dp=(eval_its, n_updates, world))
        return pending if _defer else pending.result()

    fit = run_training   # the north-star name for the same call

    @_on_model_device
    def normalize_samples(self, params):
        """Reference mdnn.py:245-248."""
        lib = self._gpu()
        ps, ldp = _lib.as_f32_rows(params, self._flat.device)
        out = torch.empty_like(ps)
        _lib.check(lib.bsig_normalize_rows(
            _lib.ptr(ps), ldp, _lib.ptr(self.output_lows), _lib.ptr(self.output_highs),
            _lib.ptr(out), out.stride(0), ps.shape[0], ps.shape[1], _lib.stream()))
        return out

    def predict_MoGs(self, xs, noise=None):
        """Reference mdnn.py:250-289: one pdf.MoG per row of ``xs`` with
        de-normalised means and [diag | strict-lower] factors.  The
        full-covariance row indexing bug of mdnn.py:281 (``L[:, :, comp_id]``
        instead of ``L[pt, :, comp_id]``) is not reproduced."""
        ntest, dim = xs.size()
        w, mu, l_d, low = self.forward(xs, noise=noise)
        w, mu, l_d = w.cpu().numpy(), mu.cpu().numpy(), l_d.cpu().numpy()
        low = None if low is None else low.cpu().numpy()
        normalize = self.output_lows is not None
        if normalize:
            lows = self.output_lows.cpu().numpy()
            rng = self.output_highs.cpu().numpy() - lows
        rows, _ = np.tril_indices(self.output_dim, -1)
        mogs = []
        for pt in range(ntest):
            ms, ls = [], []
            for k in range(self.n_gaussians):
                m = mu[pt, :, k]
                diag = l_d[pt, :, k]
                lower = None if low is None else low[pt, :, k]
                if normalize:           # Rng * T: row i scaled by rng[i]
                    m = m * rng + lows
                    diag = diag * rng
                    lower = None if lower is None else lower * rng[rows]
                ms.append(m.astype(np.float32))
                ls.append((diag if lower is None else np.concatenate([diag, lower]))
                          .astype(np.float32))
            mogs.append(pdf.MoG(a=w[pt, :], ms=ms, Ls=ls))
        return mogs
