"""The in-launch K slices (k_gemm_h4.hip) while several PROCESSES share the GPU: their grids compete for the CUs, so a slice's peers
may start late -- the case the bounded wait / abandon / close protocol exists for (a workgroup never waits without bound for one that
may not have started, and every block of every tile is still finished exactly once).  Three workers at a time, integer operands,
every product bit-exact; once with the ordinary patience (30 us) and once with none (every early slice abandons: the last slice
finishes everything from memory while other processes' grids interleave)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("wait_us", [30, 0])
def test_sliced_gemms_from_three_processes_on_one_gpu(wait_us):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "h4_shared_worker.py"), str(seed), "60", str(wait_us)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for seed in (1, 2, 3)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()                                              # the exact process started above
            out, _ = p.communicate()
            outs.append((None, out.decode(errors="replace")))
            continue
        outs.append((p.returncode, out.decode(errors="replace")))
    assert all(rc == 0 for rc, _ in outs), outs
