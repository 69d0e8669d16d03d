// main.cpp — application wiring in the order of the reference's main()
// (src/main.cu:18-116): select GPU, read settings.txt, one Convolution per
// pair of channels, per half the MIDI mapping + initial values + IR bank,
// start, connect ports, wait for a key, report the average runtime.
// Built against the fake JACK of this directory it runs offline:
// `mcconv_host --periods N` drives N periods of synthetic input instead of
// waiting on stdin.
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <vector>

#include "conv.h"
#include "settings.h"

int main(int argc, char** argv) {
    uint64_t periods = 0;
    const char* settingsPath = "settings.txt";
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--periods") && i + 1 < argc) periods = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--settings") && i + 1 < argc) settingsPath = argv[++i];
    }
    selectGpu();

    Settings settings;
    settings.open(settingsPath);
    auto count = settings.u32("conv.count");
    assert(count % 2 == 0 && "conv.count must be a multiple of 2");
    count /= 2;

    std::map<std::string, RawMidi::Device*> midiDevices;
    std::vector<Convolution*> instances;
    for (uint32_t n = 0; n < count; n++) {
        const auto fs1 = settings.u32("conv[%d].fftSize", n * 2 + 0);
        const auto fs2 = settings.u32("conv[%d].fftSize", n * 2 + 1);
        assert(fs1 == fs2 && "a convolution pair needs identical fft sizes");
        auto* c = new Convolution(std::string("hipconv_") + char('1' + n), fs1);
        instances.push_back(c);
        for (int i = 0; i < 2; i++) {
            const int idx = n * 2 + i;
            const auto deviceId = settings.str("conv[%d].cc.device", idx);
            if (!deviceId.empty()) {
                auto& dev = midiDevices[deviceId];
                if (!dev) dev = new RawMidi::Device(deviceId);
                c->cc[i].device = dev;
                dev->handler = c;
            }
            // MIDI controller numbers and initial values, table-driven: "conv[<idx>].cc.<name>" / ".value.<name>"
            Convolution::CC& half = c->cc[i];
            struct { const char* name; uint8_t* slot; } controllers[] = {
                {"message", &half.message}, {"select", &half.select},   {"predelay", &half.predelay},
                {"dry", &half.dry},         {"wet", &half.wet},         {"speed", &half.speed},
                {"panDry", &half.panDry},   {"panWet", &half.panWet},   {"level", &half.level}};
            for (auto& k : controllers) *k.slot = settings.u8("conv[%d].cc.%s", idx, k.name);
            struct { const char* name; size_t* slot; } counts[] = {
                {"select", &half.value.select}, {"predelay", &half.value.predelay}, {"speed", &half.value.speed}};
            for (auto& k : counts) *k.slot = settings.u32("conv[%d].value.%s", idx, k.name);
            struct { const char* name; float* slot; } gains[] = {{"dry", &half.value.dry},
                                                                  {"wet", &half.value.wet},
                                                                  {"panDry", &half.value.panDry},
                                                                  {"panWet", &half.value.panWet},
                                                                  {"level", &half.value.level}};
            for (auto& k : gains) *k.slot = settings.f32("conv[%d].value.%s", idx, k.name);

            std::ifstream index(settings.str("conv[%d].index", idx));
            std::string path;
            for (size_t j = 0; std::getline(index, path); j++) {
                if (path.empty()) continue;
                WavFile w(path);
                c->prepare(j, w);
            }
        }
        c->start();
        for (int i = 0; i < 2; i++) {
            const int idx = n * 2 + i;
            jack_connect(c->handle, settings.str("conv[%d].input", idx).c_str(), jack_port_name(c->capture[i]));
            jack_connect(c->handle, jack_port_name(c->playback[i]), settings.str("conv[%d].output", idx).c_str());
            if (auto* d = c->cc[i].device)
                if (!d->isOpen) d->start();
        }
    }

    if (periods) {
        // offline: the fake JACK server pushes seeded noise through every instance
        struct Feed {
            std::mt19937 rng{1234};
            std::uniform_real_distribution<float> u{-0.25f, 0.25f};
        } feed;
        auto fill = [](uint64_t, float** bufs, size_t n, jack_nframes_t nframes, void* user) {
            auto* f = static_cast<Feed*>(user);
            for (size_t b = 0; b < n; b++)
                for (jack_nframes_t s = 0; s < nframes; s++) bufs[b][s] = f->u(f->rng) + 0.01f;
        };
        for (auto* c : instances) fakejack_run(c->handle, periods, fill, nullptr, &feed);
    } else {
        std::cin.get();
    }

    for (auto* c : instances) {
        for (int i = 0; i < 2; i++)
            if (auto* d = c->cc[i].device)
                if (d->isOpen) d->stop();
        if (c->isRunning()) c->stop();
        Log::info(c->name, "Average convolution runtime: %f ms", c->avgRuntime());
        delete c;
    }
    for (auto& kv : midiDevices) delete kv.second;
    return 0;
}
