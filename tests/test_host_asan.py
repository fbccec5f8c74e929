"""CPU: the launcher code of libvargp_hip (argument checks, workspace carving, descriptor handling -- everything that runs
before a kernel launch) under host-side AddressSanitizer (sanitizers; device ASan needs xnack, which the GPU
pool does not offer).  `make -C tests/native asan` builds the library with -fsanitize=address -fno-gpu-sanitize and a
driver that walks every entry point's error paths and size queries without a GPU."""
import os
import subprocess

from conftest import ROOT


def test_launcher_code_under_host_asan():
    nat = os.path.join(ROOT, 'tests', 'native')
    build = subprocess.run(['make', '-C', nat, '-j8', 'asan'], capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0')   # (the HIP runtime keeps its own allocations)
    run = subprocess.run([os.path.join(nat, 'asan_build', 'asan_host')], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and 'asan_host: ok' in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
    assert 'AddressSanitizer' not in run.stderr, run.stderr[-4000:]
