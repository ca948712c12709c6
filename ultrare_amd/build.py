"""Build libultrare_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m ultrare_amd.build [--force] [--timeline OUT.so] [--out OUT.so -DNAME=VALUE ...]

The library is compiled for MI355X only (--offload-arch=gfx950); hipcc
cross-compiles without a GPU.  The built .so is git-ignored but travels with the
tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libultrare_hip.so')
SOURCES = ['ure_common.hip', 'mf_train.hip', 'tag_prep.hip', 'mf_eval.hip', 'job_io.hip', 'perm_tags.hip', 'perm_chain.hip', 'mf_init.hip', 'ot.hip', 'ot_solver.cpp', 'host_rng.cpp', 'mt_jump.cpp', 'host_layout.cpp']
# -amdgpu-kernarg-preload-count: gfx950 delivers the first kernel arguments in SGPRs at wave launch, which
# removes the first of the step kernel's dependent scalar-load rounds (bench: 13.1 -> 12.6 us per launch)
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-munsafe-fp-atomics',
         '-ffp-contract=off', '-Wall', '-Wno-unused-result', '-pthread', '-mllvm', '-amdgpu-kernarg-preload-count=8']


# host_normal_avx2.cpp is built apart: -mavx2 -mfma -ffp-contract=fast (its arithmetic must contract as PyTorch's own build of the same
# header does: bit-identical normals) against the include directory of the installed PyTorch
AVX2_SOURCE = 'host_normal_avx2.cpp'
AVX2_FLAGS = ['-O3', '-std=c++17', '-fPIC', '-mavx2', '-mfma', '-ffp-contract=fast', '-c']


def torch_include():
    """The installed PyTorch's include directory when it ships ATen/native/cpu/avx_mathfun.h, else None."""
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        inc = os.path.join(os.path.dirname(spec.origin), 'include')
        return inc if os.path.exists(os.path.join(inc, 'ATen', 'native', 'cpu', 'avx_mathfun.h')) else None
    except Exception:
        return None


def source_hash(extra=()):
    """sha256 (16 hex digits) over every file of csrc/, the public header and the compiler flags: the
    identity of the code a library, a profile or a counter file belongs to."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith('.')) + [os.path.join(ROOT, 'include', 'ultrare_hip.h')]
    for path in files:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(list(FLAGS) + list(extra) + AVX2_FLAGS + [str(torch_include() is not None)]).encode())
    return h.hexdigest()[:16]


STEP_KERNEL_FILES = ['mf_train.hip', 'mf_touch.h', 'mf_index.h', 'tag_prep.h', 'tag_prep.hip', 'ure_internal.h']


def step_kernel_hash():
    """Hash of the sources the training step kernels are compiled from (+ the public header and the flags): what a
    PMC counter file of those kernels is tied to (bench.py quotes `roofline.traffic` only on a match)."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, f) for f in STEP_KERNEL_FILES]:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    # of the public header only what those kernels compile against: the shard descriptor
    with open(os.path.join(ROOT, 'include', 'ultrare_hip.h')) as f:
        text = f.read()
    a, b = text.index('typedef struct ure_shard'), text.index('} ure_shard_t;')
    h.update(text[a:b].encode())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def built_hash(lib=None):
    """The hash a built library reports (ure_source_hash), read without loading it into this process
    through ctypes' global namespace: a throw-away handle is enough, the symbol is plain C."""
    lib = lib or LIB
    if not os.path.exists(lib):
        return None
    with open(lib, 'rb') as f:
        blob = f.read()
    i = blob.find(b'URE_SRC_HASH=')
    return blob[i + 13:i + 29].decode() if i >= 0 else None


def _stale():
    return built_hash() != source_hash()


def build(force=False, verbose=False, timeline=None, defines=(), out=None):
    """timeline=PATH builds a diagnostic library there instead (per-workgroup timestamps inside the
    step kernel, -DURE_TIMELINE; see tools/exp_timeline.py) and leaves the product library alone."""
    if not timeline and not out and not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libultrare_hip.so cannot be built')
    out = timeline or out or LIB
    extra = (['-DURE_TIMELINE'] if timeline else []) + ['-D' + d for d in defines]
    # csrc/perm_tags.hip exchanges data between the waves of a workgroup with the workgroup-scope release / acquire of NON-tgsplit mode
    # (LLVM AMDGPU memory model); in threadgroup-split mode it would read stale lines silently
    if any('tgsplit' in f and not f.startswith('-mno') for f in FLAGS + extra + os.environ.get('HIPCC_COMPILE_FLAGS_APPEND', '').split()):
        raise RuntimeError('libultrare_hip.so must not be built with -mtgsplit (csrc/perm_tags.hip: MEMORY MODEL)')
    inc = torch_include()
    obj = out + '.avx2.o'
    cmd0 = [hipcc, '-x', 'c++'] + AVX2_FLAGS + (['-DURE_HAVE_AVX_MATHFUN', '-I', inc] if inc else []) + ['-o', obj, os.path.join(CSRC, AVX2_SOURCE)]
    cmd = [hipcc] + FLAGS + extra + [f'-DURE_SOURCE_HASH="URE_SRC_HASH={source_hash(extra)}"'] + ['-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-o', out] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ['-Wl,' + obj]          # (as a linker argument: hipcc takes every plain input for HIP source)
    if verbose:
        print(' '.join(cmd0), flush=True)
        print(' '.join(cmd), flush=True)
    try:
        subprocess.check_call(cmd0)
        subprocess.check_call(cmd)
    finally:
        if os.path.exists(obj):
            os.remove(obj)
    return out


if __name__ == '__main__':
    tl = sys.argv[sys.argv.index('--timeline') + 1] if '--timeline' in sys.argv else None
    out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
    defs = [a[2:] for a in sys.argv if a.startswith('-D')]
    print(build(force='--force' in sys.argv, verbose=True, timeline=tl, defines=defs, out=out))
