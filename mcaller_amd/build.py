"""Build libmcaller_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is a translation unit of its own (objects under mcaller_amd/build/, kept out of history): a change to one kernel unit
recompiles that unit and links -- the scan alone is ~15 s, the whole device side ~35 s when the units build side by side."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['csrc/mc_stream.hip', 'csrc/mc_k0.hip', 'csrc/mc_scan.hip', 'csrc/mc_emit.hip', 'csrc/mc_fused.hip', 'csrc/mc_literal.hip', 'csrc/mc_classify.hip', 'csrc/mc_rowtext.hip',
           'csrc/mc_train.hip', 'csrc/mc_parse.cpp', 'csrc/mc_format.cpp', 'csrc/mc_fastq.cpp', 'csrc/mc_common.cpp', 'csrc/mc_synth.cpp']
OUT = os.path.join(HERE, 'libmcaller_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-pthread']
WARN = ['-Wall', '-Wno-unused-function', '-Wno-unused-const-variable']


def shared_deps():
    """Headers and included files: every unit depends on them."""
    deps = [os.path.join(os.path.dirname(HERE), 'include', 'mcaller_hip.h')]
    deps += [os.path.join(HERE, 'csrc', f) for f in os.listdir(os.path.join(HERE, 'csrc')) if f.endswith(('.inc', '.h'))]
    return deps


def build_lib(force=False, verbose=True, out=None, defines=()):
    """out / defines: a variant build for kernel experiments (tools/variants.sh), e.g. defines=('MC_TILE=2048',)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    macros = ['-D' + d for d in defines]
    for macro in ('MC_TILE',):
        if os.environ.get(macro):
            macros.append('-D%s=%s' % (macro, os.environ[macro]))
    if out is not None:                       # (a variant: one command, nothing cached)
        subprocess.check_call([hipcc] + FLAGS + ['-shared', '-o', out] + macros + srcs + ['-ldl'])
        return out
    common = shared_deps()
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time (several workers of a sharded run, or pytest-xdist processes, may all find the library stale at the
    # same moment); objects and the library are written beside their place and renamed into it
    import fcntl
    with open(os.path.join(objdir, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(hipcc, srcs, macros, common, objdir, force, verbose)


def _build_locked(hipcc, srcs, macros, common, objdir, force, verbose):
    stamp = os.path.join(objdir, 'macros.txt')           # (objects built with other macros are stale)
    if not os.path.exists(stamp) or open(stamp).read() != ' '.join(macros):
        force = True
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        newest = max(os.path.getmtime(d) for d in [src] + common)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            tmp = '%s.tmp.%d' % (obj, os.getpid())
            jobs.append(([hipcc] + FLAGS + WARN + macros + ['-c', src, '-o', tmp], tmp, obj))
    if not jobs and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(o) for o in objs):
        return OUT
    # the units side by side, at most eight compilers at a time and never more than the host has CPUs (each is one thread, ~1.5 GB)
    width = max(1, min(len(jobs), len(os.sched_getaffinity(0)), int(os.environ.get('MCALLER_BUILD_JOBS', '8'))))
    failed, running, todo = [], [], list(jobs)
    while todo or running:
        while todo and len(running) < width:
            cmd, tmp, obj = todo.pop(0)
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            running.append((cmd, tmp, obj, subprocess.Popen(cmd)))
        cmd, tmp, obj, proc = running.pop(0)
        rc = proc.wait()
        if rc != 0:
            failed.append((rc, cmd))
            if os.path.exists(tmp):
                os.remove(tmp)
        else:
            os.replace(tmp, obj)
    if failed:
        for rc, cmd in failed[1:]:
            print('build failed (%d): %s' % (rc, ' '.join(cmd)), file=sys.stderr)
        raise subprocess.CalledProcessError(failed[0][0], failed[0][1])
    open(stamp, 'w').write(' '.join(macros))
    out_tmp = '%s.tmp.%d' % (OUT, os.getpid())
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', out_tmp] + objs + ['-ldl']
    if verbose:
        print(' '.join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(out_tmp, OUT)
    return OUT


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
