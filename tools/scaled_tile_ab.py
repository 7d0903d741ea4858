"""Scaled-batch fit with the 96-wide GEMM tiles (default) and with the whole-width 128 x 288 / 288 x 128
tiles (BSIG_GEMM_WIDE_TILE=1, library built with BSIG_BUILD_WIDE_TILES=1 ./build.sh) at one and two
workgroups per CU."""
import os, sys, subprocess, json
sys.path.insert(0, '.')
for env in ({}, {'BSIG_GEMM_WIDE_TILE': '1', 'BSIG_GEMM_WIDE_WGS': '256'}, {'BSIG_GEMM_WIDE_TILE': '1', 'BSIG_GEMM_WIDE_WGS': '512'}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, 'bench.py', '--only-scaled-batch'], env=e, capture_output=True, text=True).stdout.strip().split('\n')[-1]
    d = json.loads(out)
    r = d['roofline_scaled']
    print(env, 'pairs/s %.0f  eff TF %.1f (%.3f)  fwd %.1f us %.3f  dW %.1f us %.3f  nll %.1e' % (
        d['pairs_per_s'], d['effective_tflops'], d['frac_of_fp32_mfma_peak'], r['forward']['avg_us'], r['forward']['frac'],
        r['weight_gradient']['avg_us'], r['weight_gradient']['frac'], d['nll_match']['max_rel_diff_all_logs']), flush=True)
