"""Controller values reach the process path without a lock.

The reference stores into the public cc[i].value fields from main() and from the MIDI thread while the JACK thread reads
them, unsynchronised (main.cu:49-70, conv.cu:255-285, midi.cu:22-59).  The engine's hand-off (csrc/params_handoff.h):
writers publish whole pairs of values under a writers-only mutex; the process call copies the current pair lock-free
and runs on exactly that pair.

CPU: the hand-off alone under -fsanitize=thread (two writers, one reader).  GPU: a controller thread sends MIDI
controller messages at random while 2000 periods run; every period's output equals the oracle's for the pair the
period reports it sampled.
"""
import os
import random
import subprocess
import threading
import time

import numpy as np
import pytest

from helpers import BASE, RMS_TOL, apply_params, rms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_handoff_under_thread_sanitizer(tmp_path):
    """tests/stub/handoff_stress.cpp: no data race (ThreadSanitizer is silent), no mixed pair, generations never go back,
    and a select's reset of vsteps that lands between a period's sample and its count-down survives (conv.cu:261, 345)."""
    exe = str(tmp_path / "handoff_stress")
    src = os.path.join(ROOT, "tests", "stub", "handoff_stress.cpp")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-o", exe, src, "-lpthread"], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT)
    if cc.returncode != 0 and b"tsan" in cc.stdout.lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert cc.returncode == 0, cc.stdout.decode(errors="replace")[-600:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    p = subprocess.run([exe, "300000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr.decode(errors="replace")[-800:])
    word = p.stdout.decode().split()
    assert word[0] == "ok" and int(word[1]) == 300000 and int(word[2]) > 1000  # the writers really ran beside the reader
    assert b"ThreadSanitizer" not in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("period", [256, 1024])
def test_controller_thread_beside_the_process_thread(oracle_mod, gpu_lib, period):
    """A second thread sends controller messages (dry, wet, both pans, level of both halves: mc_handle_cc through
    Convolution.onMidiMessage) at random while the first runs 2000 periods, as the MIDI thread does beside the JACK
    thread in the reference.  Each period reports the generation of the parameter pair it sampled; the controller thread
    logs every pair it published (single writer: generation -> values).  Replayed on the oracle - for period k, the
    logged pair of the generation period k reports - the output is the oracle's: every period ran on ONE published pair,
    never a mixture, and no period was lost.  The periods are also parked ahead while nothing moves (JACK path), so the
    give-up route runs under the same traffic."""
    from cuda_audio_amd.engine import Convolution
    from cuda_audio_amd.synth import make_input, make_ir

    n_ref, nper = 8192, 2000
    x = make_input(nper * period, seed=77)
    irs = [make_ir(6000, seed=81, norm=0.05), make_ir(5200, seed=83, norm=0.05)]
    c = Convolution("handoff", n_ref, max_batch=8, period=period, stream_threshold=8)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    p0, p1 = dict(BASE, select=0, wet=0.6), dict(BASE, select=1, level=0.9)
    apply_params(c, p0, p1, False)
    dev = object()
    names = ("dry", "wet", "panDry", "panWet", "level")
    for half in (0, 1):
        cc = c.cc[half]
        cc.device, cc.message = dev, 176
        cc.select, cc.predelay, cc.speed = 100 + half, 102 + half, 104 + half  # (not sent: no IR switch, no predelay change)
        cc.dry, cc.wet, cc.panDry, cc.panWet, cc.level = (10 + 5 * half + k for k in range(5))

    def snapshot():
        return [{k: getattr(c.cc[h].value, k) for k in names} for h in (0, 1)]

    published = {c.param_generation(published=True): snapshot()}
    stop = threading.Event()
    rng = random.Random(5)

    def controller():
        while not stop.is_set():
            half = rng.randrange(2)
            ctl = 10 + 5 * half + rng.randrange(5)
            c.onMidiMessage(dev, bytes([176, ctl, rng.randrange(20, 110)]))
            published[c.param_generation(published=True)] = snapshot()  # (single writer: nobody publishes in between)
            time.sleep(rng.choice((0.0, 0.0002, 0.001, 0.004)))

    th = threading.Thread(target=controller)
    got = np.zeros((2, nper * period), np.float32)
    gens = []
    th.start()
    try:
        for k in range(nper):
            s = slice(k * period, (k + 1) * period)
            got[:, s] = np.stack(c.onProcess(x[0, s], x[1, s]))
            gens.append(c.param_generation())
            if k % 40 == 0:
                time.sleep(0.003)  # (the controller thread gets its turns; periods are parked ahead in between)
    finally:
        stop.set()
        th.join()
    stats = c.park_stats()
    c.close()
    assert len(set(gens)) > 20, "the controller thread hardly ran"
    assert all(b >= a for a, b in zip(gens, gens[1:]))
    assert all(g in published for g in gens)

    ref = oracle_mod.RefCompat(n_ref, True)
    for i, ir in enumerate(irs):
        ref.prepare(i, ir)
    apply_params(ref, p0, p1, True)
    want = np.zeros((2, nper * period))
    k = 0
    while k < nper:  # runs of periods on one generation
        j = k
        while j < nper and gens[j] == gens[k]:
            j += 1
        for half in (0, 1):
            ref.set(half, **published[gens[k]][half])
        s = slice(k * period, j * period)
        want[:, s] = ref.process(x[0, s], x[1, s], block=period)
        k = j
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e} over {len(set(gens))} generations (parked periods: {stats})"
