"""Command line of the reference (main.py:5-72) on the MI355X engine.

Same flags, defaults and assertions; additions are optional:
  --k N          embedding width (the reference hard-codes 16, config.py:19)
  --parallel 0   train the shards of a SISA call one after the other, testing after every epoch as the reference does.  The default
                 (1) trains them side by side (and across ranks when launched with torch.distributed.run, one process per GPU) and
                 rebuilds the per-epoch tests from end-of-epoch snapshots: models, log0 and every log series are bit-identical
                 (tests/test_gpu_surface.py::test_parallel_equals_sequential_bitwise), the same lines are printed -- after the
                 call instead of during it -- and a 5-shard ml-1m learn takes 10 ms instead of 50.  One exception to
                 "bit-identical": a shard's own test_rmse comes from the total set's predictions (ure_eval_subset) when the shard's test
                 set is the total set's rows of its users, and is then summed pair by pair in double instead of by float wave partials --
                 equal to rounding (1e-12 relative), not to the bit, with a run that evaluates the shard's set on its own (another test
                 set, URE_EVAL_SUBSET=0, snapshots beyond URE_SNAPSHOT_LIMIT_GB); NDCG and HR are the same bits on both routes
  --dataset toy  the small rating set shipped with the reference (data/toy)
  --data-dir / --save-dir   roots of data/ and result/ (default: ./data, ./result)
"""
import argparse
import os

parser = argparse.ArgumentParser()
parser.add_argument('--dataset', type=str, default='ml1m', help='dataset name')
parser.add_argument('--epoch', type=int, default=50, help='number of epochs')
parser.add_argument('--worker', type=int, default=24, help='number of CPU workers (accepted, unused: no DataLoader workers)')
parser.add_argument('--verbose', type=int, default=1, help='verbose type')
parser.add_argument('--group', type=int, default=2, help='number of groups')
parser.add_argument('--layer', nargs='+', default=[64, 32], help='setting of layers')
parser.add_argument('--learn', type=str, default='sisa', help='type of learning and unlearning')
parser.add_argument('--delper', type=int, default=2, help='deleted user proportion')
parser.add_argument('--deltype', type=str, default='rand', help='deletion type')
parser.add_argument('--k', type=int, default=16, help='embedding width')
parser.add_argument('--parallel', type=int, default=1, help='1 (default): shards side by side / across GPUs; 0: one after the other')
parser.add_argument('--group-type', type=str, default='emb-ot', help="'emb-ot' (reference) or 'uniform'")
parser.add_argument('--data-dir', type=str, default=None)
parser.add_argument('--save-dir', type=str, default=None)


def main(argv=None):
    args = parser.parse_args(argv)

    assert args.dataset in ['ml1m', 'toy']
    assert args.epoch > 0
    assert args.worker > 0
    assert args.verbose in [0, 1, 2]
    assert args.group >= 0
    for i in args.layer:
        assert type(i) == int
    assert args.learn in ['sisa']
    assert args.delper in [2, 5]
    assert args.deltype in ['rand']
    assert args.group_type in ['emb-ot', 'uniform']

    import torch
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get('URE_DIST_BACKEND', 'nccl'), device_id=torch.device('cuda', local)
                                if os.environ.get('URE_DIST_BACKEND', 'nccl') == 'nccl' else None)
        if args.group > 0 and not args.parallel:
            # one process per GPU only makes sense with the shards spread over the ranks: without it every rank would train
            # every shard and write the same files
            if dist.get_rank() == 0:
                print('WORLD_SIZE > 1: --parallel 1 implied (shards placed over the ranks)')
            args.parallel = 1

    from .config import InsParam, Instance

    torch.manual_seed(42)   # the reference never seeds the CPU generator; a fixed run needs it
    param = InsParam(args.dataset, args.epoch, args.worker, args.layer, args.group, args.delper, args.deltype,
                     k=args.k, parallel=bool(args.parallel), data_dir=args.data_dir)
    ins = Instance(param, save_dir=args.save_dir)

    if args.group == 0:
        ins.runFull(is_save=True, verbose=args.verbose)
    else:
        ins.runGroup(is_save=True, learn_type=args.learn, group_type=args.group_type, n_group=args.group,
                     verbose=args.verbose)


if __name__ == '__main__':
    main()
