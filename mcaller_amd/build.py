"""Build libmcaller_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['csrc/mc_device.hip', 'csrc/mc_train.hip', 'csrc/mc_parse.cpp', 'csrc/mc_format.cpp', 'csrc/mc_fastq.cpp',
           'csrc/mc_common.cpp', 'csrc/mc_synth.cpp']
OUT = os.path.join(HERE, 'libmcaller_hip.so')


def build_lib(force=False, verbose=True, out=None, defines=()):
    """out / defines: a variant build for kernel experiments (tools/variants.sh), e.g. defines=('MC_TILE=2048',)."""
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    if out is not None:
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-pthread', '-o', out]
        cmd += ['-D' + d for d in defines] + srcs + ['-ldl']
        subprocess.check_call(cmd)
        return out
    deps = srcs + [os.path.join(os.path.dirname(HERE), 'include', 'mcaller_hip.h')]
    deps += [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc')) if f.endswith(('.inc', '.h'))]     # (included files)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
           '-Wall', '-Wno-unused-function', '-Wno-unused-const-variable', '-pthread', '-o', OUT] + srcs + ['-ldl']
    for macro in ('MC_TILE',):
        if os.environ.get(macro):
            cmd.insert(1, '-D%s=%s' % (macro, os.environ[macro]))
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
