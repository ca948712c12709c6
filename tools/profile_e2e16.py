#!/usr/bin/env python3
"""Host timeline of Sisa.learn at BASELINE configs[4]'s shape (16 shards, k = 16) -- tools/profile_e2e.py with other numbers."""
import os, sys
os.environ.setdefault('URE_HOST_TRACE', '1')
sys.argv = [sys.argv[0]]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profile_e2e.py')).read()
src = src.replace('S = 5', 'S = 16').replace('= 32, 0.1, 42, 30000', '= 16, 0.1, 42, 30000')
exec(compile(src, 'profile_e2e.py', 'exec'))
