#include "conv.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "../../include/mcconv.h"

namespace {
void check(int rc, const char* what) {
    // the reference asserts on every CUDA/cuFFT return code (asserts enabled): abort with a message
    if (rc != MC_OK) {
        Log::error("conv", "%s failed: %s", what, mc_last_error());
        std::abort();
    }
}
}  // namespace

Convolution::Convolution(const std::string& name, size_t fftSize)
    : JackClient(name), capture{nullptr, nullptr}, playback{nullptr, nullptr}, _fftSize(fftSize) {
    mc_config cfg;
    mc_default_config(&cfg);
    cfg.n_ref = fftSize;
    cfg.max_batch = 256;
    check(mc_create(&cfg, &_engine), "mc_create");
}

Convolution::~Convolution() {
    mc_destroy(_engine);
    _engine = nullptr;
}

void Convolution::onStart() {
    // reference conv.cu:197-204: activate first, then register 2 outputs and 2 inputs
    activate();
    playback[0] = addOutput("playback_1");
    playback[1] = addOutput("playback_2");
    capture[0] = addInput("capture_1");
    capture[1] = addInput("capture_2");
}

void Convolution::prepare(size_t idx, const WavFile& wav, size_t nframes) {
    check(mc_load_ir(_engine, idx, &wav.buffer[0].x, wav.numFrames, nframes), "mc_load_ir");
    if (idx + 1 > _nirs) _nirs = idx + 1;
}

// cc[i].value is plain public data written by main() and by the MIDI thread
// (main.cu:49-70, conv.cu:255-276); it is handed to the engine at each block.
void Convolution::pushParams() {
    for (int i = 0; i < 2; i++) {
        mc_cc_value v;
        v.select = cc[i].value.select;
        v.predelay = cc[i].value.predelay;
        v.speed = cc[i].value.speed;
        v.vsteps = _pushedVsteps[i] = cc[i].value.vsteps;
        v.dry = cc[i].value.dry;
        v.wet = cc[i].value.wet;
        v.panDry = cc[i].value.panDry;
        v.panWet = cc[i].value.panWet;
        v.level = cc[i].value.level;
        check(mc_set_params(_engine, i, &v), "mc_set_params");
    }
}

void Convolution::pullVsteps() {
    for (int i = 0; i < 2; i++) {
        mc_cc_value v;
        check(mc_get_params(_engine, i, &v), "mc_get_params");
        // counts down once per block (conv.cu:345,353).  A select that arrived from the MIDI thread while the engine
        // was processing has reset vsteps to speed (conv.cu:261): like the reference's in-place decrement, the
        // write-back must not undo it - only the value that was handed to the engine is replaced
        if (cc[i].value.vsteps == _pushedVsteps[i]) cc[i].value.vsteps = v.vsteps;
    }
}

void Convolution::onProcess(size_t nframes) {
    auto in1 = capture[0] ? (const float*)jack_port_get_buffer(capture[0], nframes) : nullptr;
    auto in2 = capture[1] ? (const float*)jack_port_get_buffer(capture[1], nframes) : nullptr;
    auto L = playback[0] ? (float*)jack_port_get_buffer(playback[0], nframes) : nullptr;
    auto R = playback[1] ? (float*)jack_port_get_buffer(playback[1], nframes) : nullptr;
    if (!in1 || !in2 || !L || !R) return;  // conv.cu:297
    if (nframes != _period) {  // jackd decides the period (256 on the README's target, 512 / 1024 in the run scripts)
        check(mc_set_period(_engine, (uint32_t)nframes), "mc_set_period");
        _period = nframes;
    }
    pushParams();
    check(mc_process(_engine, in1, in2, L, R, nframes), "mc_process");
    pullVsteps();
}

void Convolution::processBatch(const float* in1, const float* in2, float* outL, float* outR, size_t nblocks) {
    pushParams();
    for (size_t done = 0; done < nblocks;) {
        const size_t n = nblocks - done < 256 ? nblocks - done : 256;
        check(mc_process_batch(_engine, in1 + done * 256, in2 + done * 256, outL + done * 256, outR + done * 256, n),
              "mc_process_batch");
        done += n;
        pullVsteps();
        pushParams();
    }
}

double Convolution::avgRuntime() const { return mc_avg_runtime_ms(_engine); }

void Convolution::onMidiMessage(const RawMidi::Device* sender, const uint8_t* buffer, size_t len) {
    // conv.cu:278-285 + handleCC :255-276, applied to the public cc[] values
    if (len < 3) return;
    for (int i = 0; i < 2; i++) {
        CC& c = cc[i];
        if (c.device != sender || c.message != buffer[0]) continue;
        const uint8_t m2 = buffer[1];
        const int v = buffer[2];
        if (c.select == m2) {
            c.value.select = (size_t)v * _nirs / 0x80;
            c.value.vsteps = c.value.speed;
            Log::info("conv", "Selected IR %zu", c.value.select);
        }
        if (c.predelay == m2) c.value.predelay = (size_t)v * CONV_MAX_PREDELAY / 0x80;
        if (c.dry == m2) c.value.dry = v / 128.0f;
        if (c.wet == m2) c.value.wet = v / 128.0f;
        if (c.panDry == m2) c.value.panDry = v / 64.0f - 1;
        if (c.panWet == m2) c.value.panWet = v / 64.0f - 1;
        if (c.level == m2) c.value.level = v / 128.0f;
        if (c.speed == m2) {
            c.value.speed = ((size_t)v * CONV_MAX_SPEED) / 0x80;
            if (c.value.vsteps > c.value.speed) c.value.vsteps = c.value.speed;
        }
    }
}
