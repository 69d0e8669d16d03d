"""The path in the reference's own shape (mc_config.form = 1, cuda_audio_amd/csrc/singlefft.hip.h): one n_ref-point
transform per call, live IR spectra stepped per bin, n_ref-long running accumulator clamped every call - BASELINE
config 2.  Checked against oracle.RefCompat, the float64 restatement of Convolution::onProcess with the same buffers
(conv.cu:287-466); the oracle is "parity unpinned" (oracle/oracle.h), so is everything here."""
import numpy as np
import pytest

from helpers import BASE, RMS_TOL, apply_params, rms

pytestmark = pytest.mark.gpu


def _single(n_ref, **kw):
    from cuda_audio_amd.engine import Convolution

    return Convolution("single", n_ref, form="single", **kw)


def _oracle_stream(oracle_mod, n_ref, irs, x, events, block=256):
    """events: {call index: (half, params)} applied before that call; returns float64 [2, n]"""
    r = oracle_mod.RefCompat(n_ref, True)
    for i, ir in enumerate(irs):
        r.prepare(i, ir)
    out = np.zeros((2, x.shape[1]))
    for q in range(x.shape[1] // block):
        for half, p in events.get(q, ()):
            r.set(half, **p)
        s = slice(q * block, (q + 1) * block)
        out[:, s] = r.process(x[0, s], x[1, s], block)
    r.close()
    return out


def _engine_stream(c, x, events, block=256):
    out = np.zeros((2, x.shape[1]), np.float32)
    for q in range(x.shape[1] // block):
        for half, p in events.get(q, ()):
            c.cc[half].value.update(**p)
        s = slice(q * block, (q + 1) * block)
        out[:, s] = np.stack(c.onProcess(x[0, s], x[1, s]))
    return out


def test_ir_spectra_carry_the_split_quirks(oracle_mod, gpu_lib):
    """Convolution::prepare (conv.cu:207-253): truncation to n_ref - nframes taps, packed transform, two-for-one split
    whose s == 0 shortcut leaves H_L[0] = sum L + j sum R and H_R[0] = 0 (Q1); against numpy's transform."""
    from cuda_audio_amd.synth import make_ir

    n_ref = 8192
    ir = make_ir(9000, seed=3)  # longer than n_ref - 1024: truncated
    c = _single(n_ref)
    c.prepare(0, ir)
    got = c.debug_read(0, 0, np.float32, 0, 2 * n_ref).reshape(2, n_ref // 2, 2)
    got = got[..., 0] + 1j * got[..., 1]
    M = n_ref // 512  # the engine keeps bin d + M c at [d][c] (the order its four-step passes touch them)
    got = got.reshape(2, M, 256).transpose(0, 2, 1).reshape(2, n_ref // 2)
    info = c.ir_info(0)
    c.close()
    n = n_ref - 1024
    assert info["taps"] == n
    lr = np.asarray(ir, np.float64).reshape(-1, 2)[:n]
    want = [np.fft.fft(lr[:, ch], n_ref)[: n_ref // 2] for ch in range(2)]
    scale = max(np.abs(w).max() for w in want)
    for ch in range(2):
        assert np.abs(got[ch, 1:] - want[ch][1:]).max() <= 2e-6 * scale
    assert abs(got[0, 0] - (lr[:, 0].sum() + 1j * lr[:, 1].sum())) <= 1e-5 * scale
    assert got[1, 0] == 0


@pytest.mark.parametrize("n_ref,taps,nb,stockham", [(4096, 2500, 64, False), (16384, 14000, 120, False), (16384, 14000, 70, True),
                                                    (131072, 88200, 40, False), (524288, 441000, 10, False)],
                         ids=["N4096", "N16384", "N16384_lds_transform", "config2_N131072", "N524288"])
def test_cold_start_and_steady_state(oracle_mod, gpu_lib, monkeypatch, n_ref, taps, nb, stockham):
    """Cold start (the live spectra ramp up from zero, Q7) into steady state, unequal halves, predelay; the 2 s IR at
    the reference's default size is BASELINE config 2 (conv.cu:287-466).  Sizes above 262144 run the second inverse
    pass as a radix-2 transform in LDS (k_sf_inv2), the others on one wavefront per row (k_sf_inv2w); the LDS path
    is also forced at a short size."""
    from cuda_audio_amd.synth import make_input, make_ir

    if stockham:
        monkeypatch.setenv("MCCONV_SF_STOCKHAM", "1")
    else:
        monkeypatch.delenv("MCCONV_SF_STOCKHAM", raising=False)

    irs = [make_ir(taps, seed=21, norm=0.3), make_ir(taps - 300, seed=23, norm=0.3)]
    x = make_input(nb * 256, seed=5)
    x[0] += 0.05
    x[1, ::2] += 0.03
    p0 = dict(BASE, select=0, predelay=300, wet=0.6, panWet=0.3, panDry=-0.2)
    p1 = dict(BASE, select=1, wet=0.4, level=0.8, panWet=-0.5, dry=0.3)
    ev = {0: ((0, p0), (1, p1))}
    want = _oracle_stream(oracle_mod, n_ref, irs, x, ev)
    c = _single(n_ref)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    got = _engine_stream(c, x, ev)
    c.close()
    assert rms(want) > 0.02
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e} (signal {rms(want):.3e})"


def test_ir_switches_and_controller_changes(oracle_mod, gpu_lib):
    """Live IR switching: the live spectra move bin by bin towards the newly selected IR (f_interpolate,
    conv.cu:15-32, 339-353) - any number of switches in flight, no voices involved - with wet / pan / level / predelay
    changes in between."""
    from cuda_audio_amd.synth import make_input, make_ir

    n_ref, nb = 8192, 160
    irs = [make_ir(5000 + 400 * k, seed=40 + k, norm=0.2) for k in range(4)]
    x = make_input(nb * 256, seed=9)
    ev = {
        0: ((0, dict(BASE, select=0)), (1, dict(BASE, select=1))),
        20: ((0, dict(select=2, vsteps=30, speed=30)),),
        25: ((1, dict(select=3, vsteps=12, speed=12)),),
        31: ((0, dict(select=1, vsteps=30, speed=30)),),  # a second switch while the first is under way
        60: ((0, dict(wet=0.9, panWet=0.7)), (1, dict(level=0.5))),
        90: ((0, dict(predelay=1000)),),
        120: ((0, dict(predelay=64, dry=0.1)),),
    }
    want = _oracle_stream(oracle_mod, n_ref, irs, x, ev)
    c = _single(n_ref)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    got = _engine_stream(c, x, ev)
    c.close()
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e} (signal {rms(want):.3e})"


def test_running_accumulator_saturates_like_the_reference(oracle_mod, gpu_lib):
    """Q4 (conv.cu:98): the reference clamps its running accumulator at every call.  The partitioned engine clamps
    the finished sample and parts from the reference at sample 1298 of this case (test_gpu_parity.py); this form keeps
    the accumulator itself and follows the reference through saturation."""
    from test_oracle import q4_case

    n_ref, x, ir, p = q4_case()
    ev = {0: ((0, p), (1, p))}
    want = _oracle_stream(oracle_mod, n_ref, [ir], x, ev)
    c = _single(n_ref)
    c.prepare(0, ir)
    got = _engine_stream(c, x, ev)
    c.close()
    assert np.abs(want).max() == 1.0 and (np.abs(want) == 1.0).sum() > 100  # it does saturate
    # (a sum that lands within rounding of +-1 may clamp on one side only: a few samples differ by what was cut)
    assert rms(got - want) <= 1e-4, f"rms {rms(got - want):.3e}"
    assert np.abs(got[:, :1298] - want[:, :1298]).max() < 1e-5
    assert np.median(np.abs(got - want)) < 1e-6


def test_tail_beyond_the_accumulator_is_dropped(oracle_mod, gpu_lib):
    """Q8 (conv.cu:94-98): a call's contribution is added at the predelay and what then falls past n_ref is lost -
    an IR of n_ref - 1024 taps with the largest predelay loses its last 8192 + samples of tail."""
    from cuda_audio_amd.synth import make_input

    n_ref, nb = 16384, 100
    taps = n_ref - 1024
    rng = np.random.default_rng(77)  # a slow decay (-9 dB at the end): the dropped tail is loud
    ir = (rng.standard_normal((taps, 2)) * np.exp(-np.arange(taps) / taps)[:, None]).astype(np.float32)
    ir *= np.float32(np.sqrt(0.5 / (ir.astype(np.float64) ** 2).sum(axis=0).max()))
    x = make_input(nb * 256, seed=11)
    x[:, 30 * 256:] = 0  # the tail rings out over silence
    p = dict(BASE, predelay=8192, wet=1.0, dry=0.0)
    ev = {0: ((0, p), (1, p))}
    want = _oracle_stream(oracle_mod, n_ref, [ir], x, ev)
    c = _single(n_ref)
    c.prepare(0, ir)
    got = _engine_stream(c, x, ev)
    c.close()
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e} (signal {rms(want):.3e})"


@pytest.mark.parametrize("period", [512, 1024])
def test_longer_periods(oracle_mod, gpu_lib, period):
    """onProcess with 512 / 1024 frames per call (conv.cu:287-466 takes any nframes <= 1024): the forward transform
    folds the second half of the period into the 512-point rows."""
    from cuda_audio_amd.synth import make_input, make_ir

    n_ref, ncalls = 8192, 40
    irs = [make_ir(6000, seed=51, norm=0.2), make_ir(5500, seed=53, norm=0.2)]
    x = make_input(ncalls * period, seed=13)
    ev = {0: ((0, dict(BASE, select=0, predelay=77)), (1, dict(BASE, select=1))), 10: ((1, dict(select=0, vsteps=8, speed=8)),)}
    want = _oracle_stream(oracle_mod, n_ref, irs, x, ev, block=period)
    c = _single(n_ref, period=period, max_batch=8)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    got = _engine_stream(c, x, ev, block=period)
    c.close()
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e} (signal {rms(want):.3e})"


def test_batches_equal_single_calls(oracle_mod, gpu_lib):
    """mc_process_batch (host buffers, chunks of max_batch) and mc_process_batch_device run the same calls back to
    back: the same bits as one mc_process per period; slices and partition shards are refused."""
    import torch

    from cuda_audio_amd._lib import McError
    from cuda_audio_amd.synth import make_input, make_ir

    n_ref, nb = 4096, 96
    ir = make_ir(3000, seed=5, norm=0.3)
    x = make_input(nb * 256, seed=17)
    p = dict(BASE, predelay=100, wet=0.7)
    outs = []
    for mode in ("calls", "host", "device"):
        c = _single(n_ref, max_batch=40)
        c.prepare(0, ir)
        apply_params(c, p, p, False)
        if mode == "calls":
            out = _engine_stream(c, x, {})
        elif mode == "host":
            out = c.process(x[0], x[1])  # 96 blocks: chunks of 40, 40, 16
        else:
            dev = torch.device("cuda:0")
            d_in = torch.from_numpy(x).to(dev)
            d_out = torch.zeros(2, nb * 256, device=dev)
            for o, n in ((0, 40), (40, 40), (80, 16)):
                c.process_device(d_in[0, o * 256:].data_ptr(), d_in[1, o * 256:].data_ptr(), d_out[0, o * 256:].data_ptr(),
                                 d_out[1, o * 256:].data_ptr(), n)
            c.sync()
            out = d_out.cpu().numpy()
            with pytest.raises(McError):
                c.process_slice_device(d_in[0].data_ptr(), d_in[1].data_ptr(), d_out[0].data_ptr(), d_out[1].data_ptr(), 8, 0, 4)
            with pytest.raises(McError):
                c.partial_device(d_in[0].data_ptr(), d_in[1].data_ptr(), d_out[0].data_ptr(), 8)
        outs.append(np.asarray(out))
        c.close()
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], outs[2])


def test_reset_and_form_from_the_environment(oracle_mod, gpu_lib, monkeypatch):
    """mc_reset returns to the cold state; MCCONV_FORM=single turns an unmodified host's engine into this form."""
    from cuda_audio_amd.engine import Convolution
    from cuda_audio_amd.synth import make_input, make_ir

    n_ref = 4096
    ir = make_ir(2000, seed=1, norm=0.3)
    x = make_input(30 * 256, seed=2)
    c = _single(n_ref)
    c.prepare(0, ir)
    a = _engine_stream(c, x, {})
    c.reset()
    b = _engine_stream(c, x, {})
    c.close()
    assert np.array_equal(a, b)
    monkeypatch.setenv("MCCONV_FORM", "single")
    c = Convolution("env", n_ref)
    c.prepare(0, ir)
    assert c.algorithmic_bytes_per_block() == 24 * n_ref
    d = _engine_stream(c, x, {})
    c.close()
    assert np.array_equal(a, d)
