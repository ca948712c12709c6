#!/usr/bin/env python3
"""The figures of a bench.py line one looks at first.   python tools/bench_summary.py bench_n1.json [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print('==', f)
    for k in ['value', 'ms_per_step', 'steps', 'device_ms_timed', 'shuffles_in_timed_region', 'value_resident_tags', 'ms_per_step_resident_tags', 'value_request']:
        print(' ', k, d.get(k))
    print('  tag_chunks', [(c['epochs'], c['by']) for c in (d.get('tag_chunks') or [])])
    r = d.get('roofline') or {}
    print('  roofline', {k: r.get(k) for k in ['frac', 'avg_launch_us', 'stream_waited_for_tags_ms', 'avg_launch_us_region', 'avg_launch_us_resident_tags', 'event_pair_pass_us', 'launches_timed']})
    u = d.get('unlearn') or {}
    print('  request', {k: u.get(k) for k in ['learn_wall_s', 'unlearn_wall_s', 'learn_wall_s_all', 'unlearn_wall_s_all', 'cold_request_s']})
    c4 = u.get('config4_16_shards_k16') or {}
    print('  config4', c4.get('learn_wall_s_all'), c4.get('unlearn_wall_s_all'))
    rf = u.get('run_full') or {}
    print('  run_full', rf.get('wall_s'), rf.get('wall_s_all'), rf.get('interactions_per_s'))
    h = d.get('roofline_hbm')
    if h:
        print('  hbm', h['value'], h['frac'], h['avg_launch_us'], 'k16', h['k16']['value'], h['k16']['frac'], h['k16']['prep_share_of_device_time'])
        fm = h['full_mf']
        print('  full_mf', {k: fm.get(k) for k in ['batch_tags', 'shuffle_ms_per_epoch', 'avg_launch_us', 'frac', 'epoch_start_ms', 'epoch_start_share_of_device_time', 'value']})
    cb = d.get('cpu_baseline')
    if cb:
        print('  cpu', cb.get('value'), cb.get('prebatched_value'), cb.get('end_to_end_value'))
