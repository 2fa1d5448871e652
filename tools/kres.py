#!/usr/bin/env python3
"""tools/kres.py [unit ...] [--rev GIT_REV] -- registers, spills, scratch, LDS and occupancy of every kernel of the given units
(default: all .hip units), from the compiler's own report (-Rpass-analysis=kernel-resource-usage).  No GPU needed.
--rev: the same for the sources of a git revision (side by side: what a change did to the allocation)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ['mc_k0', 'mc_scan', 'mc_emit', 'mc_fused', 'mc_literal', 'mc_classify', 'mc_train', 'mc_stream']


def report(src_dir, unit):
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-c',
           os.path.join(src_dir, 'mcaller_amd', 'csrc', unit + '.hip'), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        if m.group(1) == 'Function Name':
            name = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(anonymous namespace\)::', '', name)
            name = re.sub(r'^void ', '', name)
            cur = re.sub(r'\(.*', '', name)
            out[cur] = {}
        elif cur:
            out[cur][m.group(1).split(' [')[0]] = m.group(2)
    return out


def main():
    args = sys.argv[1:]
    rev = None
    if '--rev' in args:
        i = args.index('--rev')
        rev = args[i + 1]
        del args[i:i + 2]
    units = args or UNITS
    old_dir = None
    if rev:
        old_dir = tempfile.mkdtemp(prefix='kres_')
        subprocess.check_call('git -C %s archive %s mcaller_amd/csrc include | tar -x -C %s' % (REPO, rev, old_dir), shell=True)
    keys = ['VGPRs', 'TotalSGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize', 'LDS Size', 'Occupancy']
    print('%-44s %s' % ('kernel', ' '.join('%12s' % k for k in keys)))
    for u in units:
        new = report(REPO, u)
        old = report(old_dir, u) if old_dir and os.path.exists(os.path.join(old_dir, 'mcaller_amd', 'csrc', u + '.hip')) else {}
        for name, v in new.items():
            cells = []
            for k in keys:
                a, b = v.get(k, '?'), old.get(name, {}).get(k)
                cells.append('%12s' % (a if b is None or b == a else '%s<-%s' % (a, b)))
            print('%-44s %s' % (name[:44], ' '.join(cells)))


if __name__ == '__main__':
    main()
