"""Compile the C restatement (oracle/mf_oracle.c) into oracle/liburoracle.so with gcc.

Plain -O2, no -ffast-math and no FMA contraction, so every fp32 operation rounds
once in the order the source states.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'mf_oracle.c')
LIB = os.path.join(HERE, 'liburoracle.so')


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ['gcc', '-O2', '-fPIC', '-shared', '-std=c99', '-ffp-contract=off', '-Wall', '-o', LIB, SRC, '-lm']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
