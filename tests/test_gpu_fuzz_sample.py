"""A sample of the randomised runs (tests/fuzz/fuzz_q8.py, tests/fuzz/fuzz_shards.py) inside the suite: the seeds that found the two
bugs of round 3 (5002: a voice whose IRs differ in length, a memory access fault; 5106 / 5143: 1024-frame periods entering the
Q8 regime earlier than 256-frame ones) and a few of every mode.  Each script runs in a process of its own and exits non-zero on
any run above 1e-5 RMS against oracle.RefCompat."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), *map(str, args)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, env=e, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "fault" not in out, out[-1500:]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("args", [(5002, 1, "general"), (5106, 1, "general"), (5143, 1, "general"), (1, 4), (5000, 6, "general"),
                                  (70000, 2, "jack"), (80000, 1, "jack512"), (20001, 2, "long")],
                         ids=["unequal_irs_in_a_voice", "period1024_a", "period1024_b", "q8", "general", "jack", "jack512", "long_batches"])
def test_randomised_runs_against_the_oracle(gpu_lib, args):
    out = _run("fuzz_q8.py", *args)
    assert f"{args[1]} runs, 0 above" in out, out[-800:]


@pytest.mark.gpu
def test_randomised_multi_gpu_layouts_on_one_card(gpu_lib):
    out = _run("fuzz_shards.py", 1, 9)
    assert "9 runs" in out and ", 0 above" in out, out[-800:]


@pytest.mark.gpu
def test_randomised_runs_single_transform_form(gpu_lib):
    out = _run("fuzz_q8.py", 30000, 4, "general", env={"MCCONV_FORM": "single"})
    assert "4 runs, 0 above" in out, out[-800:]
