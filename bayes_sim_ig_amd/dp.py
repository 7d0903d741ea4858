"""Data-parallel fit over the GPUs of one node.

The reference is single-device; this is the one exchange the path needs:
every rank holds a full replica and its shard of the (theta, trajectory)
pairs, computes the gradient of its minibatch rows scaled by 1/B_global and
the flat fp32 gradient buffer is summed with ONE all-reduce per update,
followed by the identical Adam step on every rank.  Summarizers and the RFF
projection are per-trajectory independent and need no communication.

The exchange lives in the C ABI (include/bsig.h ``bsig_comm_*``: RCCL over
xGMI, enqueued on the fit's stream) and the whole per-update loop
grad -> all-reduce -> apply is driven from C (``bsig_fit_run_dp``): Python
touches nothing between the updates of a run_training call.
``torch.distributed`` is only the rendezvous — it hands rank 0's RCCL unique id
to the other ranks.  When RCCL cannot carry the group (gloo groups: CPU tests,
several ranks sharing one GPU) the communicator wraps a torch.distributed
exchange behind the same C entry points (``bsig_comm_init_external``).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class _DevicePtr:
    """A raw fp32 device buffer as a ``__cuda_array_interface__`` object."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<f4',
                                         'data': (int(ptr), False), 'version': 2}


class DataParallel:
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.comm = None
        self._keep = None
        self.transport = None

    def _root(self):
        return dist.get_global_rank(self.group, 0) if self.group is not None else 0

    # ------------------------------------------------------------ communicator
    def init_comm(self, device=None, transport=None):
        """Create the C-ABI communicator.  ``transport``: 'rccl' (default for
        nccl groups: one rank per GPU) or 'torch' (the group's own collectives
        behind bsig_comm_init_external; default for gloo groups)."""
        if self.comm is not None:
            return self
        lib = _lib.load()
        if transport is None:
            transport = 'rccl' if dist.get_backend(self.group) == 'nccl' else 'torch'
        handle = C.c_void_p()
        if transport == 'rccl':
            dev = torch.device(device if device is not None else 'cuda')
            index = dev.index if dev.index is not None else torch.cuda.current_device()
            box = [None]
            if self.rank == 0:
                raw = (C.c_ubyte * _lib.COMM_ID_BYTES)()
                _lib.check(lib.bsig_comm_unique_id(raw))
                box[0] = bytes(raw)
            dist.broadcast_object_list(box, src=self._root(), group=self.group)
            raw = (C.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
            _lib.check(lib.bsig_comm_init(raw, self.world, self.rank, index, C.byref(handle)))
        elif transport == 'torch':
            self._keep = _lib.EXCHANGE_FN(self._torch_exchange)
            _lib.check(lib.bsig_comm_init_external(
                self.world, self.rank, C.cast(self._keep, C.c_void_p), None, C.byref(handle)))
        else:
            raise ValueError('unknown transport %r' % (transport,))
        self.comm, self.transport = handle, transport
        return self

    def _torch_exchange(self, ctx, op, buf, n, root, stream):
        """bsig_exchange_fn over the torch.distributed group (functional path:
        host-synchronous).  ``buf`` is a device pointer when the caller's tensors
        live on a GPU, a host pointer in the CPU tests."""
        try:
            if self._on_gpu:
                torch.cuda.current_stream().synchronize()
                dev_view = torch.as_tensor(_DevicePtr(buf, n), device='cuda')
                t = dev_view.cpu()
            else:
                t = torch.frombuffer((C.c_float * n).from_address(buf), dtype=torch.float32)
            if op == _lib.EXCHANGE_SUM:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            else:
                src = dist.get_global_rank(self.group, root) if self.group is not None else root
                dist.broadcast(t, src=src, group=self.group)
            if self._on_gpu:
                dev_view.copy_(t)
                torch.cuda.current_stream().synchronize()
            return 0
        except Exception as exc:        # never unwind through the C frames
            import traceback
            traceback.print_exc()
            self._error = exc
            return 1

    _on_gpu = False
    _error = None

    def resident_calls(self):
        """(diagnostics) run_training calls this rank ran RESIDENT across the gradient exchange: one
        launch per call, the all-reduces on a second stream (BSIG_DP_RESIDENT)."""
        return int(_lib.load().bsig_comm_resident_calls(self.comm)) if self.comm is not None else 0

    def set_resident(self, mode):
        """Over the BSIG_DP_RESIDENT policy, for this communicator: False never, True always where
        covered, None back to the policy."""
        if self.comm is not None:
            _lib.load().bsig_comm_set_resident(self.comm, -1 if mode is None else int(bool(mode)))

    def close(self):
        if self.comm is not None:
            _lib.load().bsig_comm_destroy(self.comm)
            self.comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------- collectives
    def _through_comm(self, fn, flat, *extra):
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        self._on_gpu = flat.is_cuda
        st = _lib.stream(flat.device) if flat.is_cuda else None
        _lib.check(fn(self.comm, C.c_void_p(flat.data_ptr()), flat.numel(), *extra, st))

    def broadcast(self, flat):
        """flat <- rank 0's values."""
        if self.comm is not None:
            return self._through_comm(_lib.load().bsig_comm_broadcast, flat, 0)
        dist.broadcast(flat, src=self._root(), group=self.group)

    def allreduce_sum(self, flat):
        if self.comm is not None:
            return self._through_comm(_lib.load().bsig_comm_allreduce, flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def gather_counts(self, value, device='cpu'):
        """Every rank's ``value`` (a non-negative integer below 2**48), on every rank.  Exact:
        the exchange sums fp32, so the value travels as three 16-bit digits, each in a slot only
        its own rank fills."""
        value = int(value)
        assert 0 <= value < (1 << 48)
        slots = torch.zeros(3 * self.world, dtype=torch.float32, device=device)
        for d in range(3):
            slots[3 * self.rank + d] = float((value >> (16 * d)) & 0xFFFF)
        self.allreduce_sum(slots)
        got = [int(v) for v in slots.cpu().tolist()]
        return [got[3 * r] | (got[3 * r + 1] << 16) | (got[3 * r + 2] << 32) for r in range(self.world)]

    def mean_losses(self, train_loss, test_loss, n_test):
        """Global mean train loss per update (equal local batches) and the
        count-weighted mean of the held-out NLL over all shards."""
        tl = train_loss.clone()
        self.allreduce_sum(tl)
        tl /= self.world
        packed = torch.cat([test_loss * float(n_test),
                            torch.full((1,), float(n_test), dtype=test_loss.dtype,
                                       device=test_loss.device)])
        self.allreduce_sum(packed)
        return tl, packed[:-1] / packed[-1].clamp_min(1.0)


def run_updates(n_updates, eval_set, grad, allreduce, apply, evaluate):
    """mdnn.py:228-242 with the gradient exchange between backward and the
    optimizer step — the schedule bsig_fit_run_dp issues from C for the HIP
    engine, restated for the CPU test of the orchestration (tests/test_dp_gloo.py
    drives a CPU gradient engine through it and the exchange through bsig_comm_*)."""
    for it in range(n_updates):
        grad()
        allreduce()
        apply()
        if it in eval_set:
            evaluate()


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n pairs for ``rank`` (first n % world
    ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def equal_shards(n, world):
    """Pairs per rank when every rank must run the same chunk schedule (one
    all-reduce per update: a rank with an extra chunk would wait forever)."""
    return n // world
