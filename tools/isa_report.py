"""Static report on the gfx950 code objects inside libsg2im_hip.so (no GPU needed): registers, LDS, scratch and resident waves
per kernel, plus two lints that this round's small-kernel sweep was made of:

  * scratch > 0            : the kernel spills registers to memory
  * serialized loads       : a loop whose body issues a global / buffer load and waits for ALL outstanding loads
                             (``s_waitcnt vmcnt(0)``) with at most two loads per wait -- one memory round trip per element

usage: python tools/isa_report.py [--lib PATH] [--all] [--loops]
The library is taken apart with objcopy (.hip_fatbin section), clang-offload-bundler and llvm-readelf / llvm-objdump from ROCm.
Importable: ``kernels(lib_path)`` -> [dict(name, vgpr, agpr, sgpr, lds, scratch, waves_per_simd)]."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, 'scene_generation_amd', 'csrc', 'libsg2im_hip.so')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def _run(cmd):
    return subprocess.run(cmd, capture_output=True, text=True)


def code_objects(lib_path, workdir):
    """extract every gfx950 code object of the fat binary; -> [path]"""
    fat = os.path.join(workdir, 'fat.bin')
    r = _run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', lib_path, fat])
    if r.returncode != 0 or not os.path.isfile(fat):
        raise RuntimeError('objcopy failed: %s' % r.stderr)
    data = open(fat, 'rb').read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(data)
        bundle, elf = os.path.join(workdir, 'b%d.bin' % k), os.path.join(workdir, 'c%d.elf' % k)
        with open(bundle, 'wb') as f:
            f.write(data[a:b])
        r = _run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + bundle,
                  '--targets=' + TARGET, '--output=' + elf])
        if r.returncode == 0 and os.path.isfile(elf) and os.path.getsize(elf) > 0:
            out.append(elf)
    return out


def demangle(names):
    r = _run(['c++filt'] + list(names))
    out = r.stdout.strip().split('\n') if r.returncode == 0 else list(names)
    return [re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', n)) for n in out]


def _short(n):
    n = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', n)          # drop the argument list
    return re.sub(r'TileCfg<(\d+), (\d+), \d+, \d+(?:, \d+)*>', r'T\1x\2', n)


def kernels_of(elf):
    notes = _run([os.path.join(LLVM, 'llvm-readelf'), '--notes', elf]).stdout
    ks = []
    for blk in notes.split('- .agpr_count:')[1:]:
        g = lambda key: int(re.search(r'\.%s:\s*(\d+)' % key, blk).group(1))
        ks.append({'mangled': re.search(r'\.name:\s*(\S+)', blk).group(1), 'agpr': int(re.match(r'\s*(\d+)', blk).group(1)),
                   'vgpr': g('vgpr_count'), 'sgpr': g('sgpr_count'), 'lds': g('group_segment_fixed_size'),
                   'scratch': g('private_segment_fixed_size'), 'wg': g('max_flat_workgroup_size')})
    for k, n in zip(ks, demangle([k['mangled'] for k in ks])):
        k['name'] = _short(n)
        regs = k['vgpr'] + k['agpr']                      # unified 512-entry file per SIMD lane, granules of 8
        regs = max(8, (regs + 7) // 8 * 8)
        k['waves_per_simd'] = min(8, 512 // regs)
    return ks


def kernels(lib_path=DEFAULT_LIB):
    with tempfile.TemporaryDirectory() as wd:
        out = []
        for elf in code_objects(lib_path, wd):
            out.extend(kernels_of(elf))
        return out


def serialized_loops(elf):
    """-> {kernel: [(body length, loads, full waits)]} for loops that wait for every load before issuing the next"""
    dis = _run([os.path.join(LLVM, 'llvm-objdump'), '-d', elf]).stdout.split('\n')
    cur, body, res = None, [], {}
    blocks = {}
    for ln in dis:
        m = re.match(r'^[0-9a-f]+ <(.*)>:', ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
        elif cur and ln.strip():
            m2 = re.search(r'//\s*([0-9A-Fa-f]+):', ln)
            blocks[cur].append((int(m2.group(1), 16) if m2 else None, ln.split('//')[0].strip()))
    for name, ins in blocks.items():
        addr_to_i = {a: i for i, (a, _) in enumerate(ins) if a is not None}
        found = []
        for i, (a, text) in enumerate(ins):
            m = re.match(r's_c?branch\w* (\d+)', text)
            if not m or a is None:
                continue
            off = int(m.group(1))
            if off < 32768:
                continue                                   # forward branch
            tgt = a + 4 + (off - 65536) * 4                 # simm16 counts dwords from the next instruction
            j = addr_to_i.get(tgt)
            if j is None or j >= i:
                continue
            seg = [t for _, t in ins[j:i]]
            loads = sum(1 for t in seg if re.match(r'(global|buffer)_load', t))
            waits = sum(1 for t in seg if 'vmcnt(0)' in t)
            if loads and waits and loads <= 2 * waits:
                found.append((i - j, loads, waits))
        if found:
            res[name] = found
    return res


def main():
    lib = DEFAULT_LIB
    if '--lib' in sys.argv:
        lib = sys.argv[sys.argv.index('--lib') + 1]
    with tempfile.TemporaryDirectory() as wd:
        elfs = code_objects(lib, wd)
        ks = []
        for e in elfs:
            ks.extend(kernels_of(e))
        print('%d kernels in %d code objects of %s' % (len(ks), len(elfs), lib))
        spills = [k for k in ks if k['scratch']]
        print('kernels with scratch (register spills): %d' % len(spills))
        for k in spills:
            print('  scratch %5d B  %s' % (k['scratch'], k['name'][:150]))
        if '--all' in sys.argv:
            print('| kernel | vgpr | agpr | sgpr | LDS B | scratch B | waves/SIMD (registers) |')
            print('|---|---|---|---|---|---|---|')
            seen = set()
            for k in sorted(ks, key=lambda k: k['name']):
                key = (k['name'], k['vgpr'], k['agpr'], k['lds'])
                if key in seen:
                    continue
                seen.add(key)
                print('| %s | %d | %d | %d | %d | %d | %d |' % (k['name'][:140], k['vgpr'], k['agpr'], k['sgpr'], k['lds'],
                                                              k['scratch'], k['waves_per_simd']))
        if '--loops' in sys.argv:
            print('loops that wait for every load before the next one (body length, loads, full waits):')
            for e in elfs:
                res = serialized_loops(e)
                names = demangle(list(res.keys()))
                for n, (_, v) in zip(names, res.items()):
                    print('  %-90s %s' % (_short(n)[:90], v[:4]))


if __name__ == '__main__':
    main()
