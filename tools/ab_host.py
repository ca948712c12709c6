#!/usr/bin/env python3
"""A/B of host-side knobs on one box: runs tools/e2e_sisa.py alternately under each environment variant, several rounds, and
prints the median learn / unlearn wall times per variant (single runs differ by more than the effects: neighbours on the host).

    python tools/ab_host.py [--shards 5 --k 32] [--rounds 5] VAR=a,b[,c] [VAR2=x,y]     (variants = the cross product)
"""
import itertools, json, os, statistics, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args[args.index('--rounds') + 1]) if '--rounds' in args else 5
shards = args[args.index('--shards') + 1] if '--shards' in args else '5'
k = args[args.index('--k') + 1] if '--k' in args else '32'
knobs = [a.split('=', 1) for a in args if '=' in a]
variants = [dict(zip([n for n, _ in knobs], combo)) for combo in itertools.product(*[v.split(',') for _, v in knobs])] or [{}]
res = {json.dumps(v): {'learn': [], 'unlearn': []} for v in variants}
for r in range(rounds):
    for v in variants:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'e2e_sisa.py'), '--shards', shards, '--k', k, '--reps', '4'],
                             env=dict(os.environ, **v), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        d = json.loads(out)
        res[json.dumps(v)]['learn'].append(d['learn_s'] * 1e3)
        res[json.dumps(v)]['unlearn'].append(d['unlearn_s'] * 1e3)
for v, t in res.items():
    print(v, 'learn median %.1f (min %.1f)' % (statistics.median(t['learn']), min(t['learn'])),
          'unlearn median %.1f (min %.1f)' % (statistics.median(t['unlearn']), min(t['unlearn'])), flush=True)
