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


def _dcg(r):
    return r[0] + np.sum(r[1:] / np.log2(np.arange(2, len(r) + 1)))


def base_test(loader, models, top_k=10):
    """utils.py:115-187 in the reference's structure: sequential DataLoader, mean of the models'
    scores, per batch a Python loop over the batch's users with boolean masks and a `uid in list`
    membership test, then per user two argsorts, HR@10 and the reference's NDCG@10."""
    loss_fn = nn.MSELoss(reduction='sum')
    size = len(loader.dataset)
    test_loss, ndcg, hr = 0, [], []
    with torch.no_grad():
        all_user, rating_dict = [], {}
        for user, item, rating in loader:
            uni_user = user.unique().tolist()
            batch_user = user.numpy().astype(np.int32)
            batch_rating = rating.numpy().astype(np.float32)
            pred = torch.stack([m(user, item) for m in models]).mean(dim=0)
            test_loss += loss_fn(pred, rating).item()
            batch_pred = pred.numpy().astype(np.float32).reshape(-1)
            for uid in uni_user:
                cur_rating = batch_rating[batch_user == uid].tolist()
                cur_pred = batch_pred[batch_user == uid].tolist()
                if uid in all_user:
                    rating_dict[uid]['rating'] += cur_rating
                    rating_dict[uid]['pred'] += cur_pred
                else:
                    all_user.append(uid)
                    rating_dict[uid] = {'rating': cur_rating, 'pred': cur_pred}
        test_loss = np.sqrt(test_loss / size)
        for uid in all_user:
            uid_rating = np.array(rating_dict[uid]['rating'])
            uid_pred = np.array(rating_dict[uid]['pred'])
            top_rating = np.argsort(uid_rating, kind='stable')[::-1][:top_k]
            top_pred = np.argsort(uid_pred, kind='stable')[::-1][:top_k]
            relevance = uid_rating[top_pred]
            hr.append(sum(relevance >= (4 / 5)) / top_k)
            common = np.isin(top_rating, top_pred)
            relevance = relevance * (relevance >= (4 / 5)) * common
            r = np.concatenate([relevance, np.zeros(top_k - len(relevance))])
            ndcg.append(_dcg(r) / _dcg(np.ones(top_k)) if len(relevance) else 0)
    return test_loss, float(np.mean(ndcg)), float(np.mean(hr))


def train_shard_with_tests(data, test, n_user, n_item, k, batch, epochs, lr=1e-3, lam=0.1, momentum=0.9, workers=0,
                           budget_s=None):
    """The end-to-end flavour (scratch.py:72-97): every epoch = the training pass + the shard test +
    the total test (here the same test set twice, as for a one-shard job).  Returns (training
    interactions processed, seconds) -- the figure the >= 50x target of BASELINE.json is stated on."""
    loader = DataLoader(_Triples(*data), batch_size=batch, shuffle=True, num_workers=workers)
    tloader = DataLoader(_Triples(*test), batch_size=batch, shuffle=False, num_workers=workers)
    model = _Model(n_user, n_item, k)
    opt = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=lam, momentum=momentum)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=50, gamma=0.95)
    loss_fn = nn.MSELoss(reduction='sum')
    seen, spent = 0, 0.0
    for _ in range(epochs):
        t0 = time.perf_counter()
        for u, i, r in loader:
            loss = loss_fn(model(u, i), r)
            loss.item()
            opt.zero_grad()
            loss.backward()
            opt.step()
            seen += len(u)
        sched.step()
        base_test(tloader, [model])
        base_test(tloader, [model])
        spent += time.perf_counter() - t0
        if budget_s is not None and spent >= budget_s:
            break
    return seen, spent
