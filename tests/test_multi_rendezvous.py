"""Host logic of the torch-free multi-process path (pilot_amd/multi.py): the RCCL unique id reaches every rank through
an atomically renamed temp file.  CPU only -- two real processes, no GPU, no RCCL call."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT
from pilot_amd import multi

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    from pilot_amd import multi
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    uid, path = multi.exchange_unique_id(rank, world, lambda: bytes(range(128)), directory=%(tmp)r, timeout=30)
    assert uid == bytes(range(128)), uid
    print("rank", rank, "ok", os.path.basename(path))
""")


def test_unique_id_reaches_every_rank_of_one_launch(tmp_path):
    env = dict(os.environ, WORLD_SIZE="3", MASTER_PORT="29999", TORCHELASTIC_RUN_ID="t1")
    code = WORKER % dict(root=ROOT, tmp=str(tmp_path))
    # ranks 1 and 2 start first and must wait for rank 0's file
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in (1, 2, 0)]
    outs = [p.communicate(timeout=60) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se
    names = {so.split()[-1] for so, _ in outs}
    assert len(names) == 1                           # same launcher parent, port and run id -> same file
    assert "29999_t1_%d" % os.getpid() in names.pop()


def test_waiting_rank_times_out_with_a_clear_error(tmp_path):
    with pytest.raises(TimeoutError, match="no unique id"):
        multi.exchange_unique_id(1, 2, lambda: b"", key="never", timeout=0.2, directory=str(tmp_path))


def test_device_list_validation():
    with pytest.raises(ValueError):
        multi._devices(None, None)
    assert list(multi._devices(None, 3)) == [0, 1, 2]
    assert list(multi._devices([0, 0], None)) == [0, 0]


def test_stdout_to_stderr_moves_c_level_prints_and_restores():
    """multi.stdout_to_stderr(): what bench.py wraps communicator creation in (RCCL prints its version banner with a plain
    printf).  In a child process: a C-level write to fd 1 inside the block lands on stderr, stdout works again afterwards."""
    import subprocess
    import sys
    code = (
        "import os, sys, ctypes\n"
        "sys.path.insert(0, %r)\n"
        "from pilot_amd import multi\n"
        "libc = ctypes.CDLL(None)\n"
        "print('before', flush=True)\n"
        "with multi.stdout_to_stderr():\n"
        "    libc.printf(b'banner from C\\n')\n"
        "    os.write(1, b'raw write\\n')\n"
        "print('after', flush=True)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["before", "after"]
    assert "banner from C" in r.stderr and "raw write" in r.stderr
