"""C++ drop-in host (cuda_audio_amd/host): WAV loader, settings parser, MIDI
reassembly on CPU; the `Convolution` class driven through fake JACK on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from helpers import RMS_TOL, rms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cuda_audio_amd", "host")
TOOL = os.path.join(HOST, "mcconv_host_tool")
DEMO = os.path.join(HOST, "mcconv_host_demo")


@pytest.fixture(scope="module")
def host_built():
    if not (os.path.exists(TOOL) and os.path.exists(DEMO)):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    return True


def _write_wav(path, lr, bits):
    """Independent little WAV writer (stereo PCM, full scale +-0.5 like wav.cu's decode)."""
    scale = 65536.0 if bits == 16 else 16777216.0
    q = np.clip(np.rint(lr.astype(np.float64) * scale), -(2 ** (bits - 1)), 2 ** (bits - 1) - 1).astype(np.int64)
    if bits == 16:
        data = q.astype("<i2").tobytes()
    else:
        b = (q & 0xFFFFFF).astype("<u4").reshape(-1)
        data = b.view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    align = 2 * bits // 8
    hdr = b"RIFF" + np.uint32(36 + len(data)).tobytes() + b"WAVEfmt " + np.uint32(16).tobytes()
    hdr += np.uint16(1).tobytes() + np.uint16(2).tobytes() + np.uint32(44100).tobytes()
    hdr += np.uint32(44100 * align).tobytes() + np.uint16(align).tobytes() + np.uint16(bits).tobytes()
    # an extra chunk before "data": the loader must search chunks by id
    extra = b"LIST" + np.uint32(4).tobytes() + b"abcd"
    open(path, "wb").write(hdr + extra + b"data" + np.uint32(len(data)).tobytes() + data)
    return q


@pytest.mark.parametrize("bits", [16, 24])
def test_wav_loader_scaling(host_built, oracle_mod, tmp_path, bits):
    """Q5: s16 / 65536, s24 / 2^24 — C++ loader == oracle decode of the same PCM words."""
    from cuda_audio_amd.synth import make_ir

    ir = make_ir(777, seed=3, norm=0.05)
    wav = str(tmp_path / f"ir{bits}.wav")
    q = _write_wav(wav, ir, bits)
    out = str(tmp_path / "dump.f32")
    info = subprocess.check_output([TOOL, "wavdump", wav, out], text=True)
    assert f"frames=777" in info and f"bits={bits}" in info
    got = np.fromfile(out, np.float32).reshape(-1, 2)
    if bits == 16:
        want = oracle_mod.wav_decode_s16(q.astype(np.int16))
    else:
        raw = (q & 0xFFFFFF).astype("<u4").reshape(-1).view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
        want = oracle_mod.wav_decode_s24(raw)
    np.testing.assert_array_equal(got, want)
    # the C++ writer round-trips through the C++ loader
    wav2 = str(tmp_path / "rt.wav")
    ir.astype(np.float32).tofile(str(tmp_path / "ir.f32"))
    subprocess.check_call([TOOL, "wavwrite", str(tmp_path / "ir.f32"), wav2, str(bits)])
    subprocess.check_call([TOOL, "wavdump", wav2, out], stdout=subprocess.DEVNULL)
    np.testing.assert_array_equal(np.fromfile(out, np.float32).reshape(-1, 2), want)


def test_settings_grammar(host_built, tmp_path):
    """`key value` pairs, '#' comments, printf-formatted keys, throw on missing key (settings.cu:4-24)."""
    p = tmp_path / "settings.txt"
    p.write_text("# comment line\nconv.count 2\n\nconv[0].fftSize \t131072\t\n# another\nconv[1].value.dry\t0.25\n"
                 "conv[0].cc.device hw:2,0\n#trailing comment without newline")
    out = subprocess.check_output([TOOL, "settings", str(p), "conv[0].fftSize", "conv[0].cc.device"], text=True)
    assert "conv[0].fftSize=131072" in out and "conv[0].cc.device=hw:2,0" in out
    assert "u32(conv.count)=2" in out and "f32(conv[1].value.dry)=0.25" in out
    assert "missing=throw" in out


def test_midi_running_status(host_built):
    """CC with running status and an interleaved real-time byte -> two complete 3-byte messages."""
    out = subprocess.check_output([TOOL, "midi", "b01540f8163f"], text=True)
    assert out.splitlines() == ["msg 176 21 64", "msg 176 22 63"]


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["partitioned", "single"])
def test_convolution_class_through_fake_jack(host_built, oracle_mod, tmp_path, form):
    """The C++ `Convolution` (conv.h surface) driven by the fake JACK server: WAV IRs, public cc[] values,
    a MIDI controller change mid-stream, output vs the restatement run with the same events.  MCCONV_FORM=single:
    the same unmodified host on the engine's single-transform form (the reference's own shape)."""
    from cuda_audio_amd.synth import make_input, make_ir

    nb, n_ref = 48, 8192
    x = make_input(nb * 256)
    irs = [make_ir(3000, seed=11, norm=0.05), make_ir(2000, seed=22, norm=0.05)]
    wavs = []
    decoded = []
    for i, (ir, bits) in enumerate(zip(irs, (16, 24))):
        w = str(tmp_path / f"ir{i}.wav")
        _write_wav(w, ir, bits)
        wavs.append(w)
        subprocess.check_call([TOOL, "wavdump", w, str(tmp_path / "d.f32")], stdout=subprocess.DEVNULL)
        decoded.append(np.fromfile(str(tmp_path / "d.f32"), np.float32).reshape(-1, 2))
    x.tofile(str(tmp_path / "in.f32"))
    cc_block, cc_val = 20, 96  # dry controller (23) of both halves -> 0.75 at block 20
    cmd = [DEMO, str(n_ref), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), str(nb)] + wavs
    cmd += ["--set", "1", "select", "1", "--set", "0", "predelay", "512", "--set", "0", "panWet", "0.5",
            "--cc", "0", "23", str(cc_val), f"@{cc_block}"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MCCONV_FORM=form))
    assert res.returncode == 0, res.stderr + res.stdout
    got = np.fromfile(str(tmp_path / "out.f32"), np.float32).reshape(2, -1)

    ref = oracle_mod.RefCompat(n_ref, True)
    for i, d in enumerate(decoded):
        ref.prepare(i, d)
    ref.set(1, select=1)
    ref.set(0, predelay=512, panWet=0.5)
    want = np.zeros((2, nb * 256))
    for b in range(nb):
        if b == cc_block:
            ref.set(None, dry=cc_val / 128.0)  # both halves map controller 23 to dry
        s = slice(b * 256, (b + 1) * 256)
        want[:, s] = ref.process(x[0, s], x[1, s])
    err = rms(got - want)
    assert err <= RMS_TOL, f"rms {err:.3e}"
    assert "avg_runtime_ms" in res.stdout


@pytest.mark.gpu
def test_convolution_class_with_512_frame_period(host_built, oracle_mod, tmp_path):
    """jackd -p512 (run_x64_86.sh): the C++ Convolution adopts the period jackd calls it with."""
    from cuda_audio_amd.synth import make_input, make_ir

    ncalls, n_ref, period = 40, 8192, 512
    x = make_input(ncalls * period)
    ir = make_ir(3000, seed=11, norm=0.05)
    w = str(tmp_path / "ir.wav")
    _write_wav(w, ir, 24)
    subprocess.check_call([TOOL, "wavdump", w, str(tmp_path / "d.f32")], stdout=subprocess.DEVNULL)
    dec = np.fromfile(str(tmp_path / "d.f32"), np.float32).reshape(-1, 2)
    x.tofile(str(tmp_path / "in.f32"))
    res = subprocess.run([DEMO, str(n_ref), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), str(ncalls), w,
                          "--period", str(period), "--set", "0", "predelay", "1024"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr + res.stdout
    got = np.fromfile(str(tmp_path / "out.f32"), np.float32).reshape(2, -1)
    ref = oracle_mod.RefCompat(n_ref, True)
    ref.prepare(0, dec)
    ref.set(0, predelay=1024)
    want = ref.process(x[0], x[1], block=period)
    assert rms(got - want) <= RMS_TOL


@pytest.mark.gpu
def test_convolution_class_over_several_devices(host_built, oracle_mod, tmp_path):
    """The C++ Convolution with a device list (conv.h): IR partitions sharded over the devices by the native driver
    (libmcconv_rccl.so, include/mcconv_group.h), offline rendering through processBatch.  Two virtual ranks on the one card of
    this box (--devices 0,0: the sum kernel stands in for RCCL) against the oracle."""
    from cuda_audio_amd.synth import make_input, make_ir

    nb, n_ref = 700, 16384
    x = make_input(nb * 256)
    wavs, decoded = [], []
    for i, taps in enumerate((9000, 7000)):
        w = str(tmp_path / f"ir{i}.wav")
        _write_wav(w, make_ir(taps, seed=11 + i, norm=0.05), 24)
        wavs.append(w)
        subprocess.check_call([TOOL, "wavdump", w, str(tmp_path / "d.f32")], stdout=subprocess.DEVNULL)
        decoded.append(np.fromfile(str(tmp_path / "d.f32"), np.float32).reshape(-1, 2))
    x.tofile(str(tmp_path / "in.f32"))
    cmd = [DEMO, str(n_ref), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), str(nb)] + wavs
    cmd += ["--devices", "0,0", "--set", "1", "select", "1", "--set", "0", "predelay", "512", "--set", "0", "panWet", "0.5"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr + res.stdout
    got = np.fromfile(str(tmp_path / "out.f32"), np.float32).reshape(2, -1)
    ref = oracle_mod.RefCompat(n_ref, True)
    for i, d in enumerate(decoded):
        ref.prepare(i, d)
    ref.set(1, select=1)
    ref.set(0, predelay=512, panWet=0.5)
    want = ref.process(x[0], x[1])
    assert rms(want) > 0.05
    assert rms(got - want) <= RMS_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("period", [256, 512])
def test_main_flow_with_settings_file(host_built, oracle_mod, tmp_path, period):
    """mcconv_host = the reference's main() order (main.cu:18-116): selectGpu, settings.txt, one Convolution per
    channel pair, MIDI mapping + initial values, IR bank from an index file, start, connect, run, avg runtime.
    Two pairs (conv.count 4) run AT THE SAME TIME on one GPU, each on its own driver thread of the fake JACK server - what
    jackd does with the reference's instances (main.cu:31-39, jackclient.cu:4-11): two engines each parking a period on the
    card, polling their own doorbells.  Every instance's output (dumped by the host together with the noise it was fed)
    is the oracle's for that input, IR bank and initial values; the same again with the instances one after the other."""
    from cuda_audio_amd.synth import make_ir

    wavs, decoded = [], []
    for j, bits in enumerate((16, 24, 16)):
        w = tmp_path / f"ir{j}.wav"
        _write_wav(str(w), make_ir(3000 + 700 * j, seed=70 + j, norm=0.05), bits)
        wavs.append(str(w))
        subprocess.check_call([TOOL, "wavdump", str(w), str(tmp_path / "d.f32")], stdout=subprocess.DEVNULL)
        decoded.append(np.fromfile(str(tmp_path / "d.f32"), np.float32).reshape(-1, 2))
    index = tmp_path / "all.index"
    index.write_text("\n".join(wavs) + "\n")
    n_ref = 16384
    lines = ["# generated by the test", "conv.count 4"]
    for i in range(4):
        lines += [f"conv[{i}].fftSize {n_ref}", f"conv[{i}].maxPredelay 8192", f"conv[{i}].index {index}",
                  f"conv[{i}].input system:capture_{i + 1}", f"conv[{i}].output system:playback_{i + 1}",
                  f"conv[{i}].cc.device hw:2,0", f"conv[{i}].cc.message 176", f"conv[{i}].cc.select 21",
                  f"conv[{i}].cc.predelay 22", f"conv[{i}].cc.dry 23", f"conv[{i}].cc.wet 24", f"conv[{i}].cc.speed 25",
                  f"conv[{i}].cc.panDry {26 + i % 2}", f"conv[{i}].cc.panWet {26 + i % 2}", f"conv[{i}].cc.level 28",
                  f"conv[{i}].value.select {i % 3}", f"conv[{i}].value.predelay 1024", f"conv[{i}].value.dry 0.5",
                  f"conv[{i}].value.wet {0.5 + 0.1 * (i // 2)}", f"conv[{i}].value.speed 100", f"conv[{i}].value.panDry 0",
                  f"conv[{i}].value.panWet {0.25 * (i % 2)}", f"conv[{i}].value.level 1.0"]
    settings = tmp_path / "settings.txt"
    settings.write_text("\n".join(lines) + "\n")
    nper = 420  # (the cold-start ramp settles after ~160 calls: from then on both instances park their periods)
    for mode in ("concurrent", "sequential"):
        prefix = str(tmp_path / f"{mode}_")
        cmd = [os.path.join(HOST, "mcconv_host"), "--settings", str(settings), "--periods", str(nper), "--period", str(period), "--dump", prefix]
        if mode == "sequential":
            cmd.append("--sequential")
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
        assert res.returncode == 0, res.stderr[-2000:]
        out = res.stdout + res.stderr
        assert out.count("Average convolution runtime") == 2
        assert "Selected GPU" in out
        assert out.count("us per period inside the process callback") == 2
        for n in range(2):
            io = [np.fromfile(f"{prefix}{n}.{e}", np.float32) for e in ("in1", "in2", "outL", "outR")]
            assert all(len(a) == nper * period for a in io)
            ref = oracle_mod.RefCompat(n_ref, True)
            for j, d in enumerate(decoded):
                ref.prepare(j, d)
            for h in range(2):
                i = 2 * n + h
                ref.set(h, select=i % 3, predelay=1024, dry=0.5, wet=0.5 + 0.1 * (i // 2), speed=100, panDry=0.0, panWet=0.25 * (i % 2), level=1.0)
            want = ref.process(io[0], io[1], block=period)
            err = rms(np.stack(io[2:]) - want)
            assert rms(want) > 0.05
            assert err <= RMS_TOL, f"{mode}, instance {n}, period {period}: rms {err:.3e}"
        print(mode, period, [ln for ln in out.splitlines() if "us per period" in ln])


def test_operators_subset(host_built):
    """operators.h subset (reference operators.h:74,84,99,321,325,338,552,786,790,1015,1229): the dim3
    arithmetic of the kernels' grid-stride idiom and the float2 operators, on fixed operands."""
    out = subprocess.check_output([TOOL, "operators"]).decode().split()
    got = dict(line.split("=") for line in out)
    assert got["offset"] == "1297,1,0"  # blockDim * blockIdx + threadIdx = 256 * 5 + 17, 1 * 1 + 0, 1 * 0 + 0
    assert got["stride"] == "16384,2,1"  # blockDim * gridDim
    assert got["add"] == "1.75,2" and got["sub"] == "1.25,-6"
    assert got["mul"] == "3,-4" and got["lmul"] == "3,-4" and got["div"] == "0.375,-0.5"
    assert got["addeq"] == "1.75,2" and got["addeqs"] == "2.25,2.5"  # scalar added to both components (operators.h:338)
    assert got["clamp"] == "1,-1"
