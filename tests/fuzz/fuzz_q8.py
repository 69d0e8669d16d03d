#!/usr/bin/env python3
"""Randomised runs in and around the Q8 regime against oracle.RefCompat (GPU box; not part of the test suite: minutes of oracle time).
Every run: random IR lengths up to n_ref - 1024 (several exactly there: the shipped shape), random controller traffic (select, predelay
- often back to 1024 -, wet, speed, pans, level), and a random mix of batches (up to 64 blocks: longer than the reach of the cut terms, so the
forward transforms sum them where the shape allows) and single periods.  usage: fuzz_q8.py [first_seed] [runs] [general]
`general`: also periods of 512 / 1024 frames, up to 7 IRs of any length (more than the engine has voices: they merge), n_ref up to 16384, any predelay.
`jack`: single 256-frame periods only, 1200 of them, controller events 40-250 calls apart: the parked path (and, in the Q8 regime, the cut terms
carried by the launch before) under parameter changes.
`long`: `general` with batches of up to 1500 calls and the switch-over to the second-level transform lowered (MCCONV_FFT2_WORK=1, set here): the
kernels of the headline (k_g2_mac; k_f2_* where gains differ per block and the IRs have >= 256 partitions: every fifth run is at n_ref = 131072).
`os` (round 4): `long` on the LAB build (MCCONV_LIB=build_ab/lib_lab.so) with the overlap-save form taken from 48 blocks on (MCCONV_OS_MIN=48, set here; the
product takes it from 12288): every settled batch of a random stream runs as one 512 x 8192-frame segment - any IR set, predelay, period size, the Q8
regime's one-term shape through the forward transforms - between batches and periods of every other form."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as oracle_mod  # noqa: E402
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
os_mode = len(sys.argv) > 3 and sys.argv[3] == "os"
long_ = os_mode or (len(sys.argv) > 3 and sys.argv[3] == "long")
jack = len(sys.argv) > 3 and sys.argv[3].startswith("jack")  # (jack512 / jack1024: periods of that many frames)  # single periods only, events far enough apart that the periods in between are parked one call ahead
general = long_ or (len(sys.argv) > 3 and sys.argv[3] == "general")
if long_:
    os.environ["MCCONV_FFT2_WORK"] = "1"
if os_mode:
    os.environ["MCCONV_OS_MIN"] = "48"
BMAX = 1500 if long_ else 64  # calls per batch
trace = os.environ.get("FUZZ_TRACE") == "1"  # every call printed before it runs and synchronised after it (fault triage)
TOL = 1e-5
cmap = (21, 22, 23, 24, 25, 26, 27, 28)
arr = (C.c_uint8 * 8)(*cmap)
bad = 0
tot = dict(drop_fft=0, forward_transforms=0, tiles=0, carried_periods=0, fused=0, split=0, resident=0, batches=0, spectra_builds=0)
for seed in range(first, first + runs):
    rng = np.random.default_rng(seed)
    n_ref = int(rng.choice([4096, 8192, 16384] if general else [4096, 8192]))
    if long_:
        n_ref = 131072 if seed % 5 == 0 else int(rng.choice([8192, 16384]))
    period = int(rng.choice([256, 256, 512, 1024])) if general else 256
    pm = period // 256
    nb = (6 * n_ref // 256 + 200) // pm  # calls
    if jack:
        n_ref, nb = 4096, 1200
        period = int(sys.argv[3][4:] or 256)
        pm = period // 256
    if long_:
        nb = (2600 if n_ref == 131072 else 4000) // pm
    nirs = int(rng.integers(2, 8 if general else 5))
    full = n_ref - 1024
    if general:
        lens = [full if rng.random() < 0.3 else int(rng.integers(300, full + 1)) for _ in range(nirs)]
    else:
        lens = [full if rng.random() < 0.6 else int(rng.integers(full // 2, full + 1)) for _ in range(nirs)]
    irs = []
    for L in lens:
        h = rng.standard_normal((L, 2)) * np.exp(-np.arange(L) / (1.5 * L))[:, None]
        irs.append((h * np.sqrt(0.003 / L)).astype(np.float32))
    x = make_input(nb * period, seed=100 + seed)
    ref = oracle_mod.RefCompat(n_ref, True)
    c = Convolution("fuzz", n_ref, max_batch=BMAX * pm, stream_threshold=8, period=period, pipeline=os.environ.get("FUZZ_PIPELINE") == "1",
                    precision=os.environ.get("FUZZ_PRECISION", "fp32"))
    for i, ir in enumerate(irs):
        ref.prepare(i, ir)
        c.prepare(i, ir)
    # start at the shipped predelay (controller value 16 -> 1024 frames)
    for half in (0, 1):
        pd0 = 16 if not general or rng.random() < 0.5 else int(rng.integers(0, 128))
        oracle_mod.handle_cc(ref.cc(half), cmap, 22, pd0, ref.num_irs())
        assert c._L.mc_handle_cc(c._h, half, arr, 22, pd0) == 0
    events, q = {}, int(rng.integers(20, 60))
    while q < nb:
        ctl = int(rng.choice(cmap))
        val = int(rng.integers(0, 128))
        if ctl == 22:
            val = 16 if rng.random() < 0.5 else int(rng.integers(0, 128))
        if ctl == 25:
            val = int(rng.integers(0, 4))
        if ctl == 28:
            val = int(rng.integers(64, 128))
        events.setdefault(q, []).append((int(rng.integers(0, 2)), ctl, val))
        q += int(rng.integers(3, 45)) if not (long_ or jack) else int(rng.integers(3, 900) if long_ else rng.integers(40, 250))
    got = np.zeros((2, nb * period), np.float32)
    want = np.zeros((2, nb * period))
    q = 0
    while q < nb:
        for half, ctl, val in events.get(q, []):
            oracle_mod.handle_cc(ref.cc(half), cmap, ctl, val, ref.num_irs())
            assert c._L.mc_handle_cc(c._h, half, arr, ctl, val) == 0
        nxt = min([e for e in events if e > q] + [nb])
        n = 1 if jack or rng.random() < (0.1 if long_ else 0.25) else int(min(rng.integers(2, BMAX + 1), nxt - q))
        s = slice(q * period, (q + n) * period)
        want[:, s] = ref.process(x[0, s], x[1, s], block=period)
        if trace:
            print(f"  call at {q}: {n} period(s), events {events.get(q, [])}, predelay {c.cc[0].value.predelay}/{c.cc[1].value.predelay} select {c.cc[0].value.select}/{c.cc[1].value.select}", flush=True)
        if n == 1:
            got[0, s], got[1, s] = c.onProcess(x[0, s], x[1, s])
        else:
            got[:, s] = c.process(x[0, s], x[1, s])
        if trace:
            c.sync()
        q += n
    try:
        st = c.drop_stats()
        st.update(c.mac_stats())
        st.update(c.os_stats())  # (overlap-save form: batches that took it, builds of its spectra)
        if os_mode and not c.lab_build():
            raise SystemExit("the os mode needs the lab build: MCCONV_LIB=build_ab/lib_lab.so")
    except Exception:  # (the single-transform form, MCCONV_FORM=single, keeps no such counters)
        st = {}
    c.close()
    for k in tot:
        tot[k] += st.get(k, 0)
    err = float(np.sqrt(np.mean((got - want) ** 2)))
    tol = TOL if os.environ.get("FUZZ_PRECISION", "fp32") == "fp32" else 2e-3 * float(np.sqrt(np.mean(want ** 2)))  # (fp16 storage: the stated relative bar)
    flag = "" if err <= tol else "   <-- FAIL"
    bad += err > tol
    print(f"seed {seed}: n_ref {n_ref}, period {period}, IRs {lens}, {len(events)} event calls, rms {err:.3e} (signal {np.sqrt(np.mean(want ** 2)):.3e}, peak {np.abs(want).max():.2f}) {st}{flag}", flush=True)
print(f"{runs} runs, {bad} above {TOL}; batches by form of the cut terms and of the partition sums: {tot}")
sys.exit(1 if bad else 0)
