// conv.h — the reference's `Convolution` class surface (src/conv.h:30-86) as a
// thin C++ front of the MI355X engine: same base classes, same public members
// (cc[2] with the CC struct, capture/playback ports, onProcess, onStart,
// avgRuntime, prepare, onMidiMessage), same macros.  All arithmetic happens in
// libmcconv.so through the C ABI of include/mcconv.h; there are no device
// buffers or kernels on this side.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "gpu.h"
#include "jackclient.h"
#include "midi.h"
#include "wav.h"

#ifndef CONV_DEFAULT_FFTSIZE
#define CONV_DEFAULT_FFTSIZE (512 * 256)
#endif
#ifndef CONV_MAX_SPEED
#define CONV_MAX_SPEED 1024
#endif
#ifndef CONV_MAX_PREDELAY
#define CONV_MAX_PREDELAY 8192
#endif
// CONV_GRIDSIZE / CONV_BLOCKSIZE (conv.h:14-20) configured the reference's
// grid-stride launches; launch shapes are internal to the engine here.

struct mc_engine;
struct mc_group;

class Convolution : public JackClient, public RawMidi::MessageHandler {
public:
    struct CC {
        RawMidi::Device* device = nullptr;
        uint8_t message = 0;
        uint8_t select = 0, predelay = 0, dry = 0, wet = 0, speed = 0, panDry = 0, panWet = 0, level = 0;
        struct {
            size_t select = 0;    // [0, number of IRs)
            size_t predelay = 0;  // [0, 8192]
            size_t speed = 100;   // [0, 1024]
            size_t vsteps = 0;
            float dry = 0.5f;     // [0, 1]
            float wet = 0.5f;     // [0, 1]
            float panDry = 0.0f;  // [-1, 1]
            float panWet = 0.0f;  // [-1, 1]
            float level = 1.0f;   // [0, 1]
        } value;
    } cc[2];

    Convolution(const std::string& name = "Conv", size_t fftSize = CONV_DEFAULT_FFTSIZE);
    // More than one device (no reference equivalent: gpu.cu:38-90 selects one): the IR partitions are sharded over `devices`
    // and the partial wet blocks summed over RCCL (include/mcconv_group.h; needs libmcconv_rccl.so linked in).  Offline
    // rendering (processBatch) only: a JACK period does not wait for a collective.  One device = the constructor above.
    Convolution(const std::string& name, size_t fftSize, const std::vector<int>& devices, size_t maxBatch = 4096);
    ~Convolution();

    JackPort capture[2];
    JackPort playback[2];

    void onProcess(size_t nframes) override;
    void onStart() override;
    double avgRuntime() const;

    void prepare(size_t idx, const WavFile& wav, size_t nframes = 1024);

    void onMidiMessage(const RawMidi::Device* sender, const uint8_t* buffer, size_t len) override;

    // offline rendering through the same engine: nblocks * 256 frames per channel
    void processBatch(const float* in1, const float* in2, float* outL, float* outR, size_t nblocks);
    size_t numIrs() const { return _nirs; }

private:
    mc_engine* _engine = nullptr;
    mc_group* _group = nullptr;  // != null: several devices; _engine is rank 0's engine (parameters are read back from it)
    size_t _fftSize;
    size_t _maxBatch = 256;
    size_t _nirs = 0;
    size_t _period = 256;
    size_t _pushedVsteps[2] = {0, 0};  // cc[i].value.vsteps as last handed to the engine (see pullVsteps)
    void pushParams();
    void pullVsteps();
};
