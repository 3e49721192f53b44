"""Build libcurious_hip.so (gfx950, hipcc) and libcurious_torch.so (the TORCH_LIBRARY face, g++) in-tree.
`python -m curious_amd.build [--force]`."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libcurious_hip.so')
SOURCES = ['api.cpp', 'her_sample.hip', 'store.hip', 'normalizer.hip', 'optim.hip', 'ipc.hip', 'actor.hip', 'env.hip',
           'mlp.hip']
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
    [os.path.join(os.path.dirname(HERE), 'include', 'curious_hip.h')]      # every header takes part in the rebuild check
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-x', 'hip']
# kernarg preloading (gfx940+): the leading scalar arguments of a kernel arrive in SGPRs with the wave instead of being
# fetched from the (uncached) kernarg segment -- mlp_lean_gemm.h "DwMap"
SOURCE_FLAGS = {'mlp.hip': ['-mllvm', '-amdgpu-kernarg-preload-count=14']}


def source_digest():
    return _digest()


def _digest():
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(SOURCE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, 'build.sha256')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIBDIR, s.rsplit('.', 1)[0] + '.o')
        # the source digest is compiled into the library (curious_build_digest): _lib.lib() refuses a binary that was
        # built from other sources than the ones next to it (the stamp file below is only the fast path of this function)
        extra = ['-DCURIOUS_BUILD_DIGEST="%s"' % dig] if s == 'api.cpp' else []
        cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(s, []) + extra + ['-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(obj)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % s)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(dig)
    return LIB


TORCH_LIB = os.path.join(LIBDIR, 'libcurious_torch.so')
TORCH_SRC = os.path.join(CSRC, 'torch_library.cpp')


def _torch_digest():
    import torch
    h = hashlib.sha256()
    for p in (TORCH_SRC, os.path.join(os.path.dirname(HERE), 'include', 'curious_hip.h')):
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def build_torch_library(force=False, verbose=True):
    """libcurious_torch.so: TORCH_LIBRARY(curious_hip, ...) over the C ABI (csrc/torch_library.cpp; host-only C++, g++ against
    the headers of the installed torch; linked to libcurious_hip.so next to it).  Loaded by curious_amd/torch_ops.py."""
    import torch
    from torch.utils import cpp_extension
    build(force=False, verbose=verbose)                           # (what it links to)
    stamp = os.path.join(LIBDIR, 'build_torch.sha256')
    dig = _torch_digest()
    if not force and os.path.exists(TORCH_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return TORCH_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = [os.environ.get('CXX', 'g++'), '-shared', '-fPIC', '-std=c++17', '-O2', '-Wall', '-Wno-unused-function',
           '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ['-I' + p for p in cpp_extension.include_paths()] + ['-I/opt/rocm/include']
    cmd += [TORCH_SRC, '-o', TORCH_LIB, '-L' + tlib, '-lc10', '-ltorch_cpu', '-ltorch', '-lc10_hip', '-ltorch_hip',
            '-L' + LIBDIR, '-lcurious_hip', '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + tlib]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, 'w') as f:
        f.write(dig)
    return TORCH_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_torch_library(force='--force' in sys.argv))
