"""Lab: build the library with extra -D flags into abtest/NAME.so (A/B runs through CURIOUS_LIB, tools/ab_bench.sh).
    python tools/build_variant.py NAME [-DFLAG ...]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curious_amd import build as B  # noqa: E402


def main():
    name, defs = sys.argv[1], sys.argv[2:]
    out = os.path.join(os.path.dirname(B.HERE), 'abtest')
    tmp = os.path.join('/tmp', 'variant_' + name)
    os.makedirs(out, exist_ok=True)
    os.makedirs(tmp, exist_ok=True)
    procs, objs = [], []
    for s in B.SOURCES:
        obj = os.path.join(tmp, s.rsplit('.', 1)[0] + '.o')
        extra = ['-DCURIOUS_BUILD_DIGEST="variant-%s"' % name] if s == 'api.cpp' else []
        procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + B.FLAGS + B.SOURCE_FLAGS.get(s, []) + defs + extra + ['-c', os.path.join(B.CSRC, s), '-o', obj]))
        objs.append(obj)
    for p in procs:
        assert p.wait() == 0
    lib = os.path.join(out, name + '.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main()
