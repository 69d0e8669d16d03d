// host_demo.cpp — offline harness used by tests/test_host_cpp.py: drives the
// C++ Convolution through the fake JACK server exactly as jackd would
// (processCallback -> onProcess), with IRs loaded from WAV files, optional
// MIDI controller messages, and raw float32 input/output files.
//
//   mcconv_host_demo <fftSize> <in.f32> <out.f32> <nperiods> <ir0.wav> [<ir1.wav>]
//                    [--period 256|512|1024] [--devices a,b,...] [--set half key value]... [--cc half controller value @period]...
// in.f32 holds [2][nperiods*period] floats (channel-major); out.f32 likewise.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "conv.h"

struct CcEvent {
    int half, controller, value;
    uint64_t block;
};

struct Io {
    std::vector<float> in, out;
    size_t n = 0;
    std::vector<CcEvent> events;
    Convolution* conv = nullptr;
    RawMidi::Device* midi = nullptr;
};

static void feed(uint64_t period, float** bufs, size_t nb, jack_nframes_t nframes, void* user) {
    Io* io = static_cast<Io*>(user);
    for (auto& e : io->events)
        if (e.block == period) {
            uint8_t msg[3] = {176, (uint8_t)e.controller, (uint8_t)e.value};
            io->midi->feed(msg, 3);
        }
    for (size_t b = 0; b < nb && b < 2; b++) memcpy(bufs[b], io->in.data() + b * io->n + period * nframes, nframes * sizeof(float));
}

static void drain(uint64_t period, float** bufs, size_t nb, jack_nframes_t nframes, void* user) {
    Io* io = static_cast<Io*>(user);
    for (size_t b = 0; b < nb && b < 2; b++) memcpy(io->out.data() + b * io->n + period * nframes, bufs[b], nframes * sizeof(float));
}

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s fftSize in.f32 out.f32 nblocks ir0.wav [ir1.wav] [--set half key value] [--cc half ctl val @block]\n", argv[0]);
        return 2;
    }
    Log::quiet(true);
    const size_t fftSize = strtoull(argv[1], nullptr, 10);
    const uint64_t nblocks = strtoull(argv[4], nullptr, 10);
    // --period N anywhere on the command line: frames per JACK period (256 unless given)
    unsigned period = 256;
    for (int i = 5; i + 1 < argc; i++)
        if (!strcmp(argv[i], "--period")) period = (unsigned)atoi(argv[i + 1]);
    // --devices a,b,...: render offline (Convolution::processBatch) on a Convolution over several devices (IR partitions
    // sharded, partial wet blocks summed over RCCL - include/mcconv_group.h; listing a device twice makes virtual ranks)
    std::vector<int> devices;
    for (int i = 5; i + 1 < argc; i++)
        if (!strcmp(argv[i], "--devices"))
            for (const char* p = argv[i + 1]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
    fakejack_configure(44100, period);
    Io io;
    io.n = nblocks * period;
    io.in.resize(2 * io.n);
    io.out.assign(2 * io.n, 0.f);
    FILE* f = fopen(argv[2], "rb");
    assert(f);
    size_t got = fread(io.in.data(), sizeof(float), io.in.size(), f);
    fclose(f);
    assert(got == io.in.size());

    selectGpu();
    Convolution single("demo", devices.size() > 1 ? 4096 : fftSize);  // (placeholder when a device list is given)
    Convolution* multi = devices.size() > 1 ? new Convolution("demo", fftSize, devices, 1024) : nullptr;
    Convolution& conv = multi ? *multi : single;
    RawMidi::Device midi("fake:0");
    midi.handler = &conv;
    io.conv = &conv;
    io.midi = &midi;
    // controller numbers as shipped in the reference's settings.txt:29-36
    for (int i = 0; i < 2; i++) {
        auto& c = conv.cc[i];
        c.device = &midi;
        c.message = 176;
        c.select = 21; c.predelay = 22; c.dry = 23; c.wet = 24; c.speed = 25;
        c.panDry = c.panWet = (uint8_t)(26 + i);
        c.level = 28;
    }
    int a = 5;
    size_t nir = 0;
    for (; a < argc && strncmp(argv[a], "--", 2); a++) {
        WavFile w(argv[a]);
        conv.prepare(nir++, w);
    }
    for (; a < argc; a++) {
        if (!strcmp(argv[a], "--set") && a + 3 < argc) {
            const int half = atoi(argv[a + 1]);
            const std::string key = argv[a + 2];
            const double v = atof(argv[a + 3]);
            auto& val = conv.cc[half].value;
            if (key == "select") val.select = (size_t)v;
            else if (key == "predelay") val.predelay = (size_t)v;
            else if (key == "speed") val.speed = (size_t)v;
            else if (key == "vsteps") val.vsteps = (size_t)v;
            else if (key == "dry") val.dry = (float)v;
            else if (key == "wet") val.wet = (float)v;
            else if (key == "panDry") val.panDry = (float)v;
            else if (key == "panWet") val.panWet = (float)v;
            else if (key == "level") val.level = (float)v;
            else { fprintf(stderr, "unknown key %s\n", key.c_str()); return 2; }
            a += 3;
        } else if ((!strcmp(argv[a], "--period") || !strcmp(argv[a], "--devices")) && a + 1 < argc) {
            a += 1;
        } else if (!strcmp(argv[a], "--cc") && a + 4 < argc) {
            CcEvent e{atoi(argv[a + 1]), atoi(argv[a + 2]), atoi(argv[a + 3]), strtoull(argv[a + 4] + 1, nullptr, 10)};
            io.events.push_back(e);
            a += 4;
        }
    }
    uint64_t done = nblocks;
    double avg = 0.0;
    if (multi) {
        conv.processBatch(io.in.data(), io.in.data() + io.n, io.out.data(), io.out.data() + io.n, nblocks * (period / 256));
    } else {
        conv.start();
        done = fakejack_run(conv.handle, nblocks, feed, drain, &io);
        assert(done == nblocks);
        avg = conv.avgRuntime();
        conv.stop();
    }
    f = fopen(argv[3], "wb");
    assert(f);
    fwrite(io.out.data(), sizeof(float), io.out.size(), f);
    fclose(f);
    printf("blocks %llu avg_runtime_ms %.4f\n", (unsigned long long)done, avg);
    delete multi;
    return 0;
}
