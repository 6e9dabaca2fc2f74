import numpy as np, time, sys
sys.path.insert(0,'.')
from oracle import oracle as O
from pilot_amd import engine
from tests.test_gpu_cell_w2 import cohort
X, offs, scale = cohort(4, 40, 5, seed=45)
for reg in (0.5, 0.1):
    for i in range(4):
        r = [O.cell_w2(X[offs[i]:offs[i+1]], X[offs[j]:offs[j+1]], scale, reg, return_info=True)[1] for j in range(4)]
        print(reg, i, [(q['iters'], '%.1e' % q['err']) for q in r])
    Wg, info = engine.cell_w2_grid(X, offs, scale, reg, return_info=True)
    print(info['iters']); print(info['err'])
    Wg2, info2 = engine.cell_w2_grid(X, offs, scale, reg, f32_floor_ulps=4.0, return_info=True)
    print(info2['iters']); print(np.abs(Wg2-Wg).max())
# timing at scale
for (N, cells, D) in [(32, 1000, 30), (64, 2000, 30), (32, 5000, 50)]:
    rng = np.random.default_rng(0)
    sizes = np.full(N, cells); offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    X = (rng.standard_normal((N, 1, D)) * 0.5 + rng.standard_normal((N, cells, D))).reshape(-1, D).astype(np.float32)
    scale = 2.0 * float(((X - X.mean(0)) ** 2).sum(1).mean())
    engine.cell_w2_grid(X[:offs[2]], offs[:3], scale, 0.1)
    t = time.time(); W, info = engine.cell_w2_grid(X, offs, scale, 0.1, return_info=True); dt = time.time() - t
    its = info['iters'].sum()
    fl = 2.0 * its * 2 * cells * cells * D
    print(N, cells, D, 'time %.3f s' % dt, 'pairs/s %.1f' % (N * N / dt), 'updates', its, 'dot TF/s %.2f' % (fl / dt / 1e12), 'iters med', np.median(info['iters']), flush=True)
