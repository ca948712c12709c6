"""oracle/torch_port.py -- TEST INFRASTRUCTURE ONLY: the CPU baseline of bench.py.

A torch-CPU restatement of the reference's *structure* (not only its arithmetic), so
that bench.py can time "what the reference does" on the GPU box's host cores, where the
reference's own files cannot travel:

  read.py:108-124    per-sample Dataset: three scalar tensors per __getitem__
  read.py:127-133    DataLoader(batch_size, shuffle=True, num_workers)
  utils.py:30-43     two nn.Embedding tables, forward = (U[u] * V[i]).sum(1)
  utils.py:58-91     per batch: MSELoss(sum), zero_grad, backward, step, loss.item()
  scratch.py:64-69   SGD(lr, weight_decay, momentum) -- dense gradients
  utils.py:115-187   per-epoch Python evaluation (optional here)

Validated in the build container against the real reference: with the same
torch.manual_seed it reproduces tests/golden/full_mf_toy.npz (tests/test_oracle_golden.py
::test_torch_port_matches_reference) and its wall time tracks the reference's.
"""
import time

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset


class _Triples(Dataset):
    def __init__(self, uid, iid, rating):
        self.u = np.asarray(uid).astype(int)
        self.i = np.asarray(iid).astype(int)
        self.r = np.asarray(rating).astype(float)

    def __len__(self):
        return len(self.u)

    def __getitem__(self, j):
        return (torch.tensor(self.u[j], dtype=torch.long), torch.tensor(self.i[j], dtype=torch.long),
                torch.tensor(self.r[j], dtype=torch.float32))


class _Model(nn.Module):
    def __init__(self, n_user, n_item, k):
        super().__init__()
        self.user_mat = nn.Embedding(n_user, k)
        self.item_mat = nn.Embedding(n_item, k)
        nn.init.normal_(self.user_mat.weight, std=1)
        nn.init.normal_(self.item_mat.weight, std=1)

    def forward(self, u, i):
        return (self.user_mat(u) * self.item_mat(i)).sum(1)


def train_shard(data, n_user, n_item, k, batch, epochs, lr=1e-3, lam=0.1, momentum=0.9, workers=0,
                budget_s=None):
    """One Scratch.train-like run on CPU without the per-epoch tests.  Returns
    (model, interactions processed, seconds inside the batch loop, epoch losses).
    `budget_s` stops after the epoch that exceeds the time budget (bounded sample)."""
    loader = DataLoader(_Triples(*data), batch_size=batch, shuffle=True, num_workers=workers)
    model = _Model(n_user, n_item, k)
    opt = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=lam, momentum=momentum)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.95)
    loss_fn = nn.MSELoss(reduction='sum')
    seen, spent, losses = 0, 0.0, []
    for _ in range(epochs):
        t0 = time.perf_counter()
        total = 0.0
        for u, i, r in loader:
            loss = loss_fn(model(u, i), r)
            total += loss.item()
            opt.zero_grad()
            loss.backward()
            opt.step()
            seen += len(u)
        sched.step()
        spent += time.perf_counter() - t0
        losses.append(float(np.sqrt(total / len(loader.dataset))))
        if budget_s is not None and spent >= budget_s:
            break
    return model, seen, spent, losses


def prebatched_rate(data, n_user, n_item, k, batch, epochs=1, lr=1e-3, lam=0.1, momentum=0.9):
    """The same arithmetic on pre-built index tensors (no DataLoader): reported beside
    the baseline so the speed-up is not credited to removing Python plumbing alone."""
    u = torch.as_tensor(np.asarray(data[0]), dtype=torch.long)
    i = torch.as_tensor(np.asarray(data[1]), dtype=torch.long)
    r = torch.as_tensor(np.asarray(data[2]), dtype=torch.float32)
    model = _Model(n_user, n_item, k)
    opt = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=lam, momentum=momentum)
    loss_fn = nn.MSELoss(reduction='sum')
    n = len(u)
    t0 = time.perf_counter()
    for _ in range(epochs):
        perm = torch.randperm(n)
        for b0 in range(0, n, batch):
            idx = perm[b0:b0 + batch]
            loss = loss_fn(model(u[idx], i[idx]), r[idx])
            loss.item()
            opt.zero_grad()
            loss.backward()
            opt.step()
    dt = time.perf_counter() - t0
    return n * epochs / dt
