"""Per-kernel register / spill / scratch / LDS figures of the gfx950 code objects inside a built library.

    python tools/isa_report.py [LIB.so] [--json] [--filter SUBSTR]

The library holds one clang offload bundle per HIP translation unit; each bundle carries one ELF
code object for gfx950 whose NT_AMDGPU_METADATA note (msgpack) lists every kernel with its
.vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size ...  This reads the
bundles directly (no GPU, no roc-obj tooling) and prints the table tests/test_cpu_host.py checks:
no step kernel may spill.
"""
import json
import os
import shutil
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
DEMANGLE = shutil.which('c++filt') or '/opt/rocm/lib/llvm/bin/llvm-cxxfilt'


def code_objects(lib):
    """The gfx950 ELF images bundled in `lib` (bytes objects)."""
    with open(lib, 'rb') as f:
        blob = f.read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from('<Q', blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if 'gfx950' in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def _msgpack(b, i=0):
    """Minimal msgpack decoder (the subset LLVM's metadata writer emits)."""
    t = b[i]
    if t <= 0x7f: return t, i + 1
    if t >= 0xe0: return t - 256, i + 1
    if 0x80 <= t <= 0x8f: return _map(b, i + 1, t & 15)
    if 0x90 <= t <= 0x9f: return _arr(b, i + 1, t & 15)
    if 0xa0 <= t <= 0xbf: return b[i + 1:i + 1 + (t & 31)].decode(), i + 1 + (t & 31)
    if t == 0xc0: return None, i + 1
    if t == 0xc2: return False, i + 1
    if t == 0xc3: return True, i + 1
    if t in (0xc4, 0xd9): n = b[i + 1]; s = b[i + 2:i + 2 + n]; return (s.decode() if t == 0xd9 else s), i + 2 + n
    if t in (0xc5, 0xda): n, = struct.unpack_from('>H', b, i + 1); s = b[i + 3:i + 3 + n]; return (s.decode() if t == 0xda else s), i + 3 + n
    if t in (0xc6, 0xdb): n, = struct.unpack_from('>I', b, i + 1); s = b[i + 5:i + 5 + n]; return (s.decode() if t == 0xdb else s), i + 5 + n
    if t == 0xca: return struct.unpack_from('>f', b, i + 1)[0], i + 5
    if t == 0xcb: return struct.unpack_from('>d', b, i + 1)[0], i + 9
    if t == 0xcc: return b[i + 1], i + 2
    if t == 0xcd: return struct.unpack_from('>H', b, i + 1)[0], i + 3
    if t == 0xce: return struct.unpack_from('>I', b, i + 1)[0], i + 5
    if t == 0xcf: return struct.unpack_from('>Q', b, i + 1)[0], i + 9
    if t == 0xd0: return struct.unpack_from('>b', b, i + 1)[0], i + 2
    if t == 0xd1: return struct.unpack_from('>h', b, i + 1)[0], i + 3
    if t == 0xd2: return struct.unpack_from('>i', b, i + 1)[0], i + 5
    if t == 0xd3: return struct.unpack_from('>q', b, i + 1)[0], i + 9
    if t == 0xdc: return _arr(b, i + 3, struct.unpack_from('>H', b, i + 1)[0])
    if t == 0xdd: return _arr(b, i + 5, struct.unpack_from('>I', b, i + 1)[0])
    if t == 0xde: return _map(b, i + 3, struct.unpack_from('>H', b, i + 1)[0])
    if t == 0xdf: return _map(b, i + 5, struct.unpack_from('>I', b, i + 1)[0])
    raise ValueError(f'msgpack type {t:#x}')


def _arr(b, i, n):
    out = []
    for _ in range(n):
        v, i = _msgpack(b, i)
        out.append(v)
    return out, i


def _map(b, i, n):
    out = {}
    for _ in range(n):
        k, i = _msgpack(b, i)
        v, i = _msgpack(b, i)
        out[k] = v
    return out, i


def metadata(elf):
    """The amdhsa metadata map of one ELF64 code object (NT_AMDGPU_METADATA = 32, owner 'AMDGPU')."""
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
    for k in range(shnum):
        _, typ, _, _, off, size = struct.unpack_from('<IIQQQQ', elf, shoff + k * shentsize)
        if typ != 7:          # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b'\0')
            d0 = p + 12 + ((namesz + 3) & ~3)
            if name == b'AMDGPU' and ntype == 32:
                return _msgpack(elf[d0:d0 + descsz])[0]
            p = d0 + ((descsz + 3) & ~3)
    return None


def kernel_descriptors(elf):
    """{kernel symbol: its 64-byte kernel descriptor} of one ELF64 code object (the `<kernel>.kd` objects of .rodata)."""
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', elf, 0x3A)
    secs = [struct.unpack_from('<IIQQQQIIQQ', elf, shoff + k * shentsize) for k in range(shnum)]
    out = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ not in (2, 11) or not entsize:          # SHT_SYMTAB / SHT_DYNSYM
            continue
        str_off = secs[link][4]
        for i in range(size // entsize):
            name_i, info, other, shndx, value, sym_size = struct.unpack_from('<IBBHQQ', elf, off + i * entsize)
            end = elf.index(b'\0', str_off + name_i)
            name = elf[str_off + name_i:end].decode()
            if name.endswith('.kd') and sym_size == 64 and 0 < shndx < shnum:
                sec = secs[shndx]
                at = sec[4] + (value - sec[3])
                out[name[:-3]] = elf[at:at + 64]
    return out


def demangle(names):
    if not os.path.exists(DEMANGLE):
        return names
    out = subprocess.run([DEMANGLE], input='\n'.join(names), stdout=subprocess.PIPE, text=True).stdout.split('\n')
    return out[:len(names)]


def kernels(lib=None):
    """-> list of dicts {name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, wg_max} for every kernel."""
    lib = lib or os.path.join(ROOT, 'ultrare_amd', 'libultrare_hip.so')
    rows = []
    for elf in code_objects(lib):
        md = metadata(elf)
        kds = kernel_descriptors(elf)
        for k in (md or {}).get('amdhsa.kernels', []):
            kd = kds.get(k['.name'])
            # COMPUTE_PGM_RSRC3 (descriptor bytes 44-47), gfx90a and later: bit 16 = TG_SPLIT (the waves of a workgroup may be placed on
            # different compute units: workgroup-scope acquire / release then need cache maintenance -- csrc/perm_tags.hip must not be built so)
            rows.append(dict(symbol=k['.name'], tg_split=(struct.unpack_from('<I', kd, 44)[0] >> 16) & 1 if kd else None, vgpr=k.get('.vgpr_count', 0), agpr=k.get('.agpr_count', 0), sgpr=k.get('.sgpr_count', 0),
                             vgpr_spill=k.get('.vgpr_spill_count', 0), sgpr_spill=k.get('.sgpr_spill_count', 0),
                             scratch=k.get('.private_segment_fixed_size', 0), lds=k.get('.group_segment_fixed_size', 0),
                             wg_max=k.get('.max_flat_workgroup_size', 0)))
    for r, n in zip(rows, demangle([r['symbol'] for r in rows])):
        r['name'] = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('ure::', '')
    return rows


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    if flt in args:
        args.remove(flt)
    rows = [r for r in kernels(args[0] if args else None) if flt in r['name']]
    if '--json' in sys.argv:
        print(json.dumps(rows, indent=1))
    else:
        print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
        for r in sorted(rows, key=lambda r: r['name']):
            print(f"{r['name'][:58]:58s} {r['vgpr']:5d} {r['agpr']:5d} {r['sgpr']:5d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch']:7d} {r['lds']:6d}")
