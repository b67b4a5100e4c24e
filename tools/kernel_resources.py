#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel of a HIP object or of libssmq.so, from the code object's metadata
(clang-offload-bundler + llvm-readelf --notes; no GPU needed).  With --isa also counts VALU / SALU / branch / scratch
instructions in the disassembly of the kernels that match.

  tools/kernel_resources.py ssmtoybox_amd/csrc/ssmq_filter_fused.o --match 'k_filter_fused<5, 4, 11, 11' --isa
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def code_object(path, tmp):
    """The gfx950 code object inside a host object / shared library (or `path` itself if it already is one)."""
    out = os.path.join(tmp, 'k.co')
    for kind in ('o', 'so'):
        r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=' + kind,
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + path, '--output=' + out, '--unbundle'],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            return out
    # a shared library: the fat binary sits in .hip_fatbin
    fat = os.path.join(tmp, 'fat.bin')
    r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', path, fat])
    if r.returncode == 0 and os.path.exists(fat) and os.path.getsize(fat) > 0:
        r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                            '--input=' + fat, '--output=' + out, '--unbundle'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            return out
    return path


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def kernels(co):
    txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], stdout=subprocess.PIPE, text=True).stdout
    recs = []
    for blk in re.split(r'\n\s+- \.agpr_count:', txt)[1:]:
        blk = '.agpr_count:' + blk
        get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, None])[1]      # noqa: E731
        recs.append({'sym': get('name'), 'vgpr': int(get('vgpr_count') or 0), 'agpr': int(blk.split()[1]),
                     'sgpr': int(get('sgpr_count') or 0), 'scratch': int(get('private_segment_fixed_size') or 0),
                     'lds': int(get('group_segment_fixed_size') or 0), 'vgpr_spill': int(get('vgpr_spill_count') or 0),
                     'sgpr_spill': int(get('sgpr_spill_count') or 0), 'wg': int(get('max_flat_workgroup_size') or 0)})
    for r, n in zip(recs, demangle([r['sym'] for r in recs])):
        r['name'] = re.sub(r'\s*\[clone.*', '', n)
    return recs


def isa_counts(co, sym):
    txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--disassemble-symbols=' + sym, co], stdout=subprocess.PIPE,
                         text=True).stdout
    c = dict(valu=0, salu=0, branch=0, scratch=0, mfma=0, ds=0, vmem=0, total=0)
    for ln in txt.splitlines():
        m = re.match(r'\s+([a-z_0-9]+)\s', ln)
        if not m:
            continue
        op = m.group(1)
        c['total'] += 1
        if op.startswith('scratch_'):
            c['scratch'] += 1
        elif 'mfma' in op:
            c['mfma'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith(('s_cbranch', 's_branch')):
            c['branch'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        elif op.startswith('ds_'):
            c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')):
            c['vmem'] += 1
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('path')
    ap.add_argument('--match', default='', help='substring of the demangled kernel name')
    ap.add_argument('--isa', action='store_true')
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(a.path, tmp)
        for r in kernels(co):
            if a.match and a.match not in r['name']:
                continue
            line = '{name}\n    vgpr {vgpr} agpr {agpr} sgpr {sgpr} scratch {scratch} B lds {lds} B spills v{vgpr_spill}/s{sgpr_spill} wg {wg}'.format(**r)
            if a.isa:
                line += '\n    isa ' + ' '.join('{}={}'.format(k, v) for k, v in isa_counts(co, r['sym']).items())
            print(line)
    return 0


if __name__ == '__main__':
    sys.exit(main())
