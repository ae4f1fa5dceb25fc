#!/bin/bash
# error of the fp32 MFMA path and of the exact-split bf16 path against the fp64 torch golden (reduced segment) and
# against the fp32 oracle (full segment): tap by tap
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in f32 bf16x3; do
DMX_GEMM=$g python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as orc, parity_utils as pu
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
orc.lib().orc_set_num_threads(32)
res = {}
for ns, seed, fn in ((4, 0, 'golden_seg_4s.npz'), (6, 3, 'golden_seg_6s.npz')):
    g = np.load('tests/golden/' + fn)
    p = f'/tmp/m{ns}.bin'; write_synthetic_model(p, ns, seed)
    m = dmx.Model(p); ctx = dmx.Context(m, int(g['seg']), 1)
    out = ctx.segment(g['mix'])
    res[f'{ns}s_vs_fp64'] = pu.relerr(out, g['out'])
    worst = 0
    for k in ('x_3', 'xt_3', 'ct_x', 'ct_xt', 'dec_0', 'dec_2'):
        a = pu.gpu_tap_as_oracle(ctx, k)
        ref = np.zeros(int(np.prod(g[f'tap_{k}_shape'])))  # only the subsampled points are stored
        got = a.reshape(-1)[g[f'tap_{k}_idx']]
        want = g[f'tap_{k}_val']
        if k.startswith('dec_'):
            continue  # (decoder taps of the product include the fused skip add)
        worst = max(worst, float(np.abs(got - want).max() / g[f'tap_{k}_absmax']))
    res[f'{ns}s_taps_vs_fp64'] = worst
    ctx.close(); m.close()
p = '/tmp/m4.bin'
m = dmx.Model(p); ctx = dmx.Context(m, 0, 1); om = orc.OracleModel(p)
mix = (0.1 * np.random.default_rng(4).standard_normal((2, 343980))).astype(np.float32)
errs, out, ref = pu.compare_segment(ctx, om, mix)
res['full_vs_oracle_out'] = errs['out']; res['full_vs_oracle_worst_tap'] = max(errs.values())
res['full_local'] = dict(pu.LAST_LOCAL).get('out_block'); res['full_min_sdr'] = dict(pu.LAST_LOCAL).get('out_min_stem_sdr_db')
print(os.environ['DMX_GEMM'], {k: (float('%.3g' % v) if v is not None else None) for k, v in res.items()})
PY
done
