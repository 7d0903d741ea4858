#!/usr/bin/env python3
"""Which earlier activity of a process breaks the resident exchange (profiles/r05_NOTES.md, "Resident
exchange", item 7)?  argv: steps before the resident fit on a 1-rank RCCL group --
'train[_nograph][_nopersist][_del]' (a single-process MDRFF-512 run_training), 'fit_nodp', 'model_only',
'libkernel', 'pinned' / 'pageable' (a host-to-device copy), 'nbstream' (a kernel on a side stream),
'stream', 'spawn' (two child processes on the GPU), 'sync', 'init' (init_process_group here, not at the
end), 'comm' (the C-ABI communicator here).  Prints whether the fit timed out and the resident calls.
With the exchange stream chosen by a probe (comm.cpp: comm_xr) every combination runs clean."""
import os
import sys
import warnings

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))


def child(rank):
    x = torch.randn(1000, 1000, device='cuda:0')
    (x @ x).sum().item()


def main():
    import bench
    import bayes_sim_ig_amd as pkg
    keep = []
    for step in sys.argv[1:]:
        if step.startswith('train'):
            pkg.MDNN.USE_GRAPH = 'nograph' not in step
            if 'nopersist' in step:
                os.environ['BSIG_NO_PERSISTENT'] = '1'
            import test_gpu_dp2 as t
            m = t._model(pkg, 512, 1e-5)
            x, y, ids = t._data(0)
            m.run_training(x.cuda(), y.cuda(), t.NU, t.B, test_frac=0.2, ids_table=ids)
            torch.cuda.synchronize()
            os.environ.pop('BSIG_NO_PERSISTENT', None)
            pkg.MDNN.USE_GRAPH = True
            if 'del' in step:
                del m
                import gc
                gc.collect()
            else:
                keep.append(m)
        elif step == 'nbstream':
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                a = torch.randn(256, 256, device='cuda:0'); (a @ a).sum().item()
            keep.append(st)
        elif step == 'pinned':
            h = torch.randn(1 << 20).pin_memory()
            d = h.to('cuda:0', non_blocking=True); torch.cuda.synchronize(); keep += [h, d]
        elif step == 'pageable':
            d = torch.randn(1 << 20).cuda(); torch.cuda.synchronize(); keep.append(d)
        elif step == 'libkernel':
            theta, states, actions = bench.synth_pairs(dict(bench.CONFIGS['cfg5']), 100, 3, 'cuda:0')
            keep.append(pkg.summary_corrdiff(states, actions)); torch.cuda.synchronize()
        elif step == 'model_only':
            import test_gpu_dp2 as t
            keep.append(t._model(pkg, 512, 1e-5))
        elif step == 'fit_nodp':
            cfg0 = dict(bench.CONFIGS['cfg5'])
            th, stt, ac = bench.synth_pairs(cfg0, 1000, 5, 'cuda:0')
            b0 = bench.build_gpu_model(pkg, cfg0, 'cuda:0', 3)
            b0.fit(th, stt, ac); torch.cuda.synchronize(); keep.append(b0)
        elif step == 'init':
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29583')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
        elif step == 'comm':       # the C-ABI communicator (and RCCL's own init) ahead of everything else
            from bayes_sim_ig_amd import dp as _dp
            keep.append(_dp.DataParallel(None).init_comm('cuda:0'))
        elif step == 'spawn':
            mp.spawn(child, nprocs=2, join=True)
        elif step == 'stream':
            keep += [torch.cuda.Stream() for _ in range(6)]
        elif step == 'sync':
            torch.cuda.synchronize()
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29583')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    cfg = dict(bench.CONFIGS['cfg5'])
    theta, states, actions = bench.synth_pairs(cfg, 3000, 21, 'cuda:0')
    os.environ['BSIG_DP_RESIDENT'] = '1'
    bs = bench.build_gpu_model(pkg, cfg, 'cuda:0', 31)
    bs.model.enable_data_parallel()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        bs.fit(theta, states, actions)
        torch.cuda.synchronize()
    print('steps', sys.argv[1:], '->', 'TIMED OUT' if any('timed out' in str(x.message) for x in w) else 'ok',
          'resident calls', bs.model._dp.resident_calls(), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
