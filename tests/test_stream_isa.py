"""The streamed-weights sparse-conv kernel issues its loads from inline asm with hand-counted
s_waitcnt values (csrc/sparse_conv.hip).  That is only sound while the compiler keeps its hands off
registers with a load in flight; tools/check_stream_isa.py verifies it on the generated gfx950 code.
Needs hipcc only (no GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_stream_kernel_isa_keeps_in_flight_registers_untouched():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_stream_isa.py')],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert 'touches of in-flight registers:   0' in r.stdout


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_sir_grid_barrier_drains_vector_memory_before_it_arrives():
    """bar_arrive of csrc/sir_fused_impl.hpp: `s_waitcnt vmcnt(0)` in front of the workgroup barrier of every arrival
    (the no-return atomics on maxima / arg-max rows / collected gradients must have landed before another workgroup
    passes the grid barrier); tools/check_sir_barrier_isa.py walks the generated gfx950 code of all three tile sizes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_sir_barrier_isa.py')],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert 'grid-barrier arrivals checked: 72' in r.stdout
    assert 'arrivals without vmcnt(0) in front of the barrier: 0' in r.stdout
