"""Ad-hoc first GPU check: GPU vs oracle at 1 spp on the four named scenes, plus timing."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle as O
from rtm_loader import load_package
R = load_package()
R.init(0)
import torch
for name in ['simple_sphere', 'multisphere', 'suzanne', 'bunny']:
    s = R.Scene(path=f'{ROOT}/assets/scene/{name}.scene', base_dir=f'{ROOT}/assets')
    o = O.OracleScene(path=f'{ROOT}/assets/scene/{name}.scene', base_dir=f'{ROOT}/assets')
    g = s.render(spp=1, seed=0, stats=True)
    c = o.render(spp=1, seed=0)
    eq_obj = np.array_equal(g['obj_id'], c['obj_id']); eq_tri = np.array_equal(g['tri_id'], c['tri_id'])
    err = np.abs(g['rgb'] - c['rgb']).max()
    bits = (g['rgb'].view(np.uint64) != c['rgb'].view(np.uint64)).sum()
    print(name, 'obj', eq_obj, 'tri', eq_tri, 'maxerr', err, 'differing f64 words', bits, 'stats eq', g['counters'] == c['counters'], flush=True)
    if g['counters'] != c['counters']:
        print('  gpu', g['counters']); print('  cpu', c['counters'])
    for spp, v1 in ((20, True), (1, False), (20, False)):
        p = R.make_params(s.width, s.height, spp=spp, seed=0, kernel_v1=v1)
        buf = torch.empty(s.height * s.width * 3, dtype=torch.float64, device='cuda')
        s.time_render_device(p, buf.data_ptr(), 1)
        ms = s.time_render_device(p, buf.data_ptr(), 3)
        rays = sum(g['counters'][k] for k in ('rays_camera', 'rays_bounce', 'rays_shadow')) * spp
        print(f'  spp={spp} v1={v1}: {ms:.2f} ms/frame  ~{rays/ms/1e3:.1f} Mrays/s (ray count scaled from 1spp)', flush=True)
