"""Runs last in the GPU session (file name): every kernel of libcurious_hip.so must have been launched by some test."""
import pytest

pytestmark = pytest.mark.gpu


def test_every_kernel_of_the_library_was_launched(request):
    from curious_amd import ops
    counts = ops.prof_launch_counts()
    print('\nlaunches per kernel in this session:')
    for name, n in sorted(counts.items(), key=lambda kv: -kv[1]):
        print('  %-28s %d' % (name, n))
    full = request.config.getoption('-m') == 'gpu' and not request.config.getoption('-k') and \
        len(request.session.items) > 60
    if not full:
        pytest.skip('only meaningful after the whole `-m gpu` session (this run selected a subset)')
    idle = [name for name, n in counts.items() if n == 0]
    assert not idle, 'kernels no GPU test launched: %s' % idle
