#!/usr/bin/env python3
"""A/B of two source trees on one box: tools/e2e_sisa.py of each tree alternately, several rounds; medians of learn / unlearn.

    python tools/ab_trees.py TREE_A TREE_B [--rounds 6] [--shards 5] [--k 32]
"""
import json, os, statistics, subprocess, sys

args = sys.argv[1:]
trees = [os.path.abspath(a) for a in args[:2]]
rounds = int(args[args.index('--rounds') + 1]) if '--rounds' in args else 6
shards = args[args.index('--shards') + 1] if '--shards' in args else '5'
k = args[args.index('--k') + 1] if '--k' in args else '32'
res = {t: {'learn': [], 'unlearn': []} for t in trees}
for r in range(rounds):
    for t in trees:
        out = subprocess.run([sys.executable, os.path.join(t, 'tools', 'e2e_sisa.py'), '--shards', shards, '--k', k, '--reps', '5'],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=t)
        if out.returncode:
            sys.exit(out.stderr[-2000:])
        out = out.stdout
        d = json.loads(out)
        res[t]['learn'].append(d['learn_s'] * 1e3)
        res[t]['unlearn'].append(d['unlearn_s'] * 1e3)
for t, v in res.items():
    print(t, 'learn median %.2f (min %.2f)' % (statistics.median(v['learn']), min(v['learn'])),
          'unlearn median %.2f (min %.2f)' % (statistics.median(v['unlearn']), min(v['unlearn'])), flush=True)
