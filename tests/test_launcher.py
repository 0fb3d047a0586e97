"""CPU: the one-process-per-GPU launcher (aero_amd/launcher.py, replacing src/ddp/executor.py:13-75) --
N supervised ranks with the torchrun environment contract, gloo standing in for RCCL."""
import os
import sys
import textwrap
import time

from aero_amd import launcher
from conftest import ROOT


def _script(tmp_path, body):
    p = tmp_path / 'worker.py'
    p.write_text('import os, sys\nsys.path.insert(0, %r)\n' % ROOT + textwrap.dedent(body))
    return str(p)


def test_two_ranks_join_one_process_group(tmp_path):
    out = tmp_path / 'out'
    w = _script(tmp_path, f'''
        from aero_amd import distrib
        distrib.init_from_env(backend='gloo')
        n = distrib.count_ranks()
        t = distrib.max_over_ranks(float(distrib.rank))
        open({str(out)!r} + os.environ['RANK'], 'w').write(f"{{n}} {{t}} {{os.environ['LOCAL_RANK']}} {{os.environ['MASTER_ADDR']}}")
        distrib.close()
    ''')
    assert launcher.spawn_ranks([w], 2, timeout_s=300)
    for r in range(2):
        assert open(str(out) + str(r)).read() == f'2 1.0 {r} 127.0.0.1'


def test_a_dead_worker_stops_the_others(tmp_path):
    w = _script(tmp_path, '''
        import time
        if os.environ['RANK'] == '1':
            sys.exit(3)
        time.sleep(120)
    ''')
    t0 = time.monotonic()
    assert launcher.spawn_ranks([w], 2, timeout_s=300) is False
    assert time.monotonic() - t0 < 60                     # rank 0 was terminated, not waited for


def test_reference_import_path():
    from src.ddp import executor
    assert executor.start_ddp_workers is launcher.start_ddp_workers and hasattr(executor, 'ChildrenManager')
    assert not launcher.under_launcher() or 'RANK' in os.environ
