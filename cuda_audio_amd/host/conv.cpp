#include "conv.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "../../include/mcconv_group.h"

// the multi-device driver is optional at link time (libmcconv_rccl.so): the single-device host and the stub host do without it
extern "C" {
int mc_group_create(const mc_config*, const int32_t*, uint32_t, mc_group**) __attribute__((weak));
void mc_group_destroy(mc_group*) __attribute__((weak));
mc_engine* mc_group_engine(mc_group*, uint32_t) __attribute__((weak));
int mc_group_load_ir(mc_group*, uint64_t, const float*, uint64_t, uint64_t) __attribute__((weak));
int mc_group_set_params(mc_group*, int, const mc_cc_value*) __attribute__((weak));
int mc_group_process_batch(mc_group*, const float*, const float*, float*, float*, uint64_t) __attribute__((weak));
const char* mc_group_last_error(void) __attribute__((weak));
}

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

Convolution::Convolution(const std::string& name, size_t fftSize, const std::vector<int>& devices, size_t maxBatch)
    : JackClient(name), capture{nullptr, nullptr}, playback{nullptr, nullptr}, _fftSize(fftSize), _maxBatch(maxBatch) {
    mc_config cfg;
    mc_default_config(&cfg);
    cfg.n_ref = fftSize;
    cfg.max_batch = (uint32_t)maxBatch;
    if (devices.size() <= 1) {
        if (!devices.empty()) cfg.device = devices[0];
        check(mc_create(&cfg, &_engine), "mc_create");
        return;
    }
    if (!mc_group_create) {
        Log::error("conv", "%zu devices requested but the multi-device driver (libmcconv_rccl.so) is not linked in", devices.size());
        std::abort();
    }
    std::vector<int32_t> devs(devices.begin(), devices.end());
    if (mc_group_create(&cfg, devs.data(), (uint32_t)devs.size(), &_group) != MC_OK) {
        Log::error("conv", "mc_group_create failed: %s", mc_group_last_error());
        std::abort();
    }
    _engine = mc_group_engine(_group, 0);
}

Convolution::~Convolution() {
    if (_group) mc_group_destroy(_group);  // (owns its engines)
    else mc_destroy(_engine);
    _group = nullptr;
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
    if (_group) {
        if (mc_group_load_ir(_group, idx, &wav.buffer[0].x, wav.numFrames, nframes) != MC_OK) {
            Log::error("conv", "mc_group_load_ir failed: %s", mc_group_last_error());
            std::abort();
        }
    } else
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
        if (_group) {
            if (mc_group_set_params(_group, i, &v) != MC_OK) {
                Log::error("conv", "mc_group_set_params failed: %s", mc_group_last_error());
                std::abort();
            }
        } else
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
    if (_group) {
        Log::error("conv", "a Convolution over several devices renders batches only (processBatch)");
        std::abort();
    }
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
        const size_t n = nblocks - done < _maxBatch ? nblocks - done : _maxBatch;
        if (_group) {
            if (mc_group_process_batch(_group, in1 + done * 256, in2 + done * 256, outL + done * 256, outR + done * 256, n) != MC_OK) {
                Log::error("conv", "mc_group_process_batch failed: %s", mc_group_last_error());
                std::abort();
            }
        } else
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
