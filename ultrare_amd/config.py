"""InsParam / Instance with the reference's names, arguments and artifact layout
(config.py:16-200), running on the MI355X engine.

Differences from the reference, all defect fixes or optional additions:
  D1  n_del = int(del_per / 100 * n_user)   (the shipped formula asks for 12,080 of 6,040 users)
  D2  dis_type = 'nor', attr = []            (set here; Scratch needs them)
  D5  data / result roots are configurable   ($ULTRARE_DATA_DIR, $ULTRARE_SAVE_DIR, default ./data, ./result)
  +   InsParam(..., k=16, parallel=False)    optional embedding width and shard-parallel mode
  +   dataset 'toy' is usable end to end     (the reference only has its batch size)
"""
import os
import warnings
from os import mkdir
from os.path import exists

import numpy as np

from .group import DATA_DIR, SAVE_DIR, Group
from .method.scratch import Scratch
from .method.sisa import Sisa
from .method.utils import atomic_save, dist_rank, saveObject
from .read import RatingData, loadData, readRating, readSparseMat

DATASETS = {
    # name: (train csv, test csv, n_user, n_item)   relative to DATA_DIR   (config.py:40-44)
    'ml1m': ('ml1m/squ0_train.csv', 'ml1m/squ0_test.csv', 6040, 3416),
    'toy': ('toy/0_train.csv', 'toy/0_test.csv', 1508, 2071),
}


class InsParam(object):
    def __init__(self, dataset='toy', epochs=50, n_worker=24, layers=[32], n_group=2, del_per=2, del_type='test',
                 k=16, parallel=False, data_dir=None):
        # model param
        self.k = k  # dimension of embedding (config.py:19 hard-codes 16)
        self.lam = 0.1  # regularization coefficient
        self.layers = layers  # unused by MF (structure of FC layers in DMF)

        # training param
        self.seed = 42
        self.n_worker = n_worker
        self.batch = 3000 if dataset == 'toy' else 30000
        self.lr = 0.001
        self.lr_decay = 0.95
        self.momentum = 0.9
        self.epochs = epochs
        self.n_group = n_group
        self.dis_type = 'nor'   # D2
        self.attr = []          # D2
        self.parallel = parallel

        # dataset-varied param
        self.del_rating = []  # 2d array/list [[uid, iid], ...]
        self.dataset = dataset
        self.max_rating = 5
        self.del_per = del_per
        self.del_type = del_type
        self.del_user = []

        if dataset in DATASETS:
            root = data_dir or DATA_DIR
            tr, te, self.n_user, self.n_item = DATASETS[dataset]
            self.train_dir = root + '/' + tr
            self.test_dir = root + '/' + te
            if self.del_type == 'rand':
                np.random.seed(0)
                n_del = int(self.del_per / 100 * self.n_user)            # D1
                self.del_user = np.random.choice(self.n_user, n_del, replace=False)
        else:
            raise ValueError(f'unknown dataset {dataset!r}; known: {sorted(DATASETS)}')

    def info(self):
        print(self.dataset, '-----------')
        print('Path of training data:', self.train_dir)
        print('Path of testing data:', self.test_dir)
        print('Number of users:', self.n_user)
        print('Number of items:', self.n_item)


class Instance(object):
    def __init__(self, param, save_dir=None):
        self.param = param
        self.save_root = save_dir or SAVE_DIR
        prefix = '/test/' if self.param.del_type == 'test' else '/' + str(self.param.del_per) + '/' + self.param.del_type + '/'
        self.name = prefix + self.param.dataset + '_g' + str(self.param.n_group)
        param_dir = self.save_root + self.name
        os.makedirs(param_dir, exist_ok=True)
        rank, dist = dist_rank()
        if rank == 0:          # several ranks (torch.distributed.run) run the same Instance: one of them writes
            # save param
            saveObject(param_dir + '/param', self.param)  # loadObject(dir + '/param')
            # save deletion
            deletion = [self.param.del_user, self.param.del_rating]
            arr = np.empty(2, dtype=object)
            arr[0], arr[1] = deletion

            def write(tmp):
                with warnings.catch_warnings(), open(tmp, 'wb') as f:
                    warnings.simplefilter('ignore')
                    np.save(f, arr)  # np.load('deletion.npy', allow_pickle=True)
            atomic_save(param_dir + '/deletion.npy', write)
        if dist is not None:
            dist.barrier()

    # read raw data (config.py:80-96)
    def _read(self, is_del=False, n_group=1, group_index=[]):
        del_user = self.param.del_user if is_del else []
        del_rating = self.param.del_rating if is_del else []
        train_rating, train_index = readRating(self.param.train_dir, self.param.n_user, self.param.max_rating,
                                               del_user, del_rating, n_group, group_index, 'a')
        # no deletion for testing data
        test_rating, _ = readRating(self.param.test_dir, self.param.n_user, self.param.max_rating,
                                    [], [], n_group, train_index)
        return train_rating, train_index, test_rating

    def _save_dir(self, is_save, saving_name):
        if not is_save:
            return ''
        save_dir = self.save_root + self.name + '/' + saving_name
        os.makedirs(save_dir, exist_ok=True)
        return save_dir

    # sub function of self.runFull (config.py:99-120)
    def _full(self, is_save, saving_name, model_type='mf', is_del=False, verbose=1):
        print(self.name, saving_name, 'begin:')
        train_rating, _, test_rating = self._read(is_del)
        train_data = loadData(RatingData(train_rating[0]), self.param.batch, self.param.n_worker)
        test_data = loadData(RatingData(test_rating[0]), self.param.batch, self.param.n_worker, False)
        save_dir = self._save_dir(is_save, saving_name)
        model = Scratch(self.param, model_type)
        trained = model.train(train_data, test_data, [], verbose, save_dir)
        print('End of training', self.name, saving_name)
        print()
        return trained

    def _full_user_mat(self):
        """config.py:132 loads result/2/rand/ml1m_g0/MF_full_train/user_mat0.npy -- the
        product of an earlier `--group 0` run.  Look beside this instance first."""
        p = self.param
        cands = [f'{self.save_root}/{p.del_per}/{p.del_type}/{p.dataset}_g0/MF_full_train/user_mat0.npy',
                 f'{self.save_root}/2/rand/ml1m_g0/MF_full_train/user_mat0.npy']
        for c in cands:
            if exists(c):
                return np.load(c, allow_pickle=True)
        raise FileNotFoundError(f'{cands[0]} not found: run the full-MF stage first (main.py --group 0), '
                                'its user matrix is the embedding the OT grouping clusters (config.py:132)')

    # sub function of self.runGroup (config.py:123-174)
    def _group(self, model_list, is_save, learn_type, saving_name, model_type='mf',
               is_del=False, group_type='uniform', n_group=5, verbose=1):
        print(self.name, saving_name, 'begin:')
        if group_type == 'uniform':
            group_index = []
        else:
            val_mat = readSparseMat(self.param.train_dir, self.param.n_user, self.param.n_item) \
                if group_type.startswith('rating') else None
            user_mat = self._full_user_mat()
            rank, dist = dist_rank()
            grouper = Group(val_mat, self.param.dataset, user_mat)
            kw = dict(verbose=False, data_dir=os.path.dirname(os.path.dirname(self.param.train_dir)))
            if dist is None:
                group_index = grouper.grouping(self.param.dataset, n_group, group_type, **kw)
            else:               # rank 0 clusters and writes the label cache (atomically); the others read it after the barrier
                if rank == 0:
                    group_index = grouper.grouping(self.param.dataset, n_group, group_type, **kw)
                dist.barrier()
                if rank != 0:
                    group_index = grouper.grouping(self.param.dataset, n_group, group_type, **kw)

        train_rating, train_index, test_rating = self._read(is_del, n_group, group_index)

        train_dlist, test_dlist = [], []
        assert learn_type in ['sisa']
        for i in range(n_group):
            train_dlist.append(loadData(RatingData(train_rating[i]), self.param.batch, self.param.n_worker))
            test_dlist.append(loadData(RatingData(test_rating[i]), self.param.batch, self.param.n_worker, False))
        test_total = np.hstack(test_rating)
        test_data = loadData(RatingData(test_total), self.param.batch, self.param.n_worker, False)

        save_dir = self._save_dir(is_save, saving_name)

        model = Sisa(self.param, model_type, n_group, train_index)
        if not is_del:
            model.learn(train_dlist, test_dlist, test_data, verbose, save_dir)
        else:
            del_user = list(self.param.del_user)
            for rating in self.param.del_rating:
                if rating[0] not in del_user:
                    del_user.append(rating[0])
            model.unlearn(model_list, train_dlist, test_dlist, test_data, del_user, verbose, save_dir)
        self.last = model
        return model.model_list

    #########################
    # runFull, runGroup
    #########################
    def runFull(self, is_save=True, verbose=1):
        '''model MF'''
        # full train without deletion
        self._full(is_save, 'MF_full_train', 'mf', False, verbose)
        # retrain from scratch after deletion
        self._full(is_save, 'MF_retrain', 'mf', True, verbose)

    def runGroup(self, is_save=True, learn_type='seq', group_type='uniform', n_group=5, verbose=1):
        '''model MF'''
        if learn_type == 'seq':
            learn_type = 'sisa'      # the published tree only ships the SISA learner (main.py:45)
        saving_name = 'MF_' + group_type + '_' + learn_type + '_learn'
        model_list = self._group([], is_save, learn_type, saving_name, 'mf', False, group_type, n_group, verbose)
        saving_name = 'MF_' + group_type + '_' + learn_type + '_unlearn'
        return self._group(model_list, is_save, learn_type, saving_name, 'mf', True, group_type, n_group, verbose)
