import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as O
from ultrare_amd import engine, rng
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'toy', '0_train.csv')
raw = O.load_csv(G)
part = O.partition(*raw, O.uniform_groups(1508, 1))[0]
N = len(part[0])
k, E = 16, 1
torch.manual_seed(11)
init = rng.mf_init(1508, 2071, k)
perms = rng.epoch_perms(rng.epoch_seeds(E, True), N)
B = N
job = engine.TrainJob([engine.ShardData(*part, 1508, 2071)], [init], [perms], k, B, E, 1e-3, 0.1, 0.9, 0.95, touch='index')
job.run()
U, V = job.tables(0)
U, V = U.cpu().numpy(), V.cpu().numpy()
U0, V0 = init[0].numpy().astype(np.float64), init[1].numpy().astype(np.float64)
u, i, r = part
for user in (1, 4, 7):
    js = np.flatnonzero(u == user)
    print('user', user, 'interactions', js, 'items', i[js], 'r', r[js])
    w = U0[user]
    for name, rr, f in (('ok', r[js], 2.0), ('r=0', 0 * r[js], 2.0), ('ge=e', r[js], 1.0)):
        acc = sum(f * (w @ V0[i[j]] - rj) * V0[i[j]] for j, rj in zip(js, rr))
        g = 0.1 * w + acc
        print('  ', name, np.abs((w - 1e-3 * g) - U[user]).max())
    print('   delta gpu', (U[user] - w)[:4], 'expected', (-1e-3 * (0.1 * w + sum(2 * (w @ V0[i[j]] - r[j]) * V0[i[j]] for j in js)))[:4])
