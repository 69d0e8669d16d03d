// main.cpp — application wiring in the order of the reference's main()
// (src/main.cu:18-116): select GPU, read settings.txt, one Convolution per
// pair of channels, per half the MIDI mapping + initial values + IR bank,
// start, connect ports, wait for a key, report the average runtime.
// Built against the fake JACK of this directory it runs offline:
// `mcconv_host --periods N` drives N periods of synthetic input instead of
// waiting on stdin - all instances AT ONCE, a driver thread per client, as jackd
// runs them (--sequential: one after the other; --period F: frames per period;
// --spacing US: a period clock; --dump PREFIX: every instance's input and output
// as raw float32 files PREFIX<i>.in1 / .in2 / .outL / .outR).
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
    const char* dump = nullptr;
    bool sequential = false;
    double spacing_us = 0.0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--periods") && i + 1 < argc) periods = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--settings") && i + 1 < argc) settingsPath = argv[++i];
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--sequential")) sequential = true;
        else if (!strcmp(argv[i], "--spacing") && i + 1 < argc) spacing_us = atof(argv[++i]);
        else if (!strcmp(argv[i], "--period") && i + 1 < argc) fakejack_configure(44100, (jack_nframes_t)atoi(argv[++i]));
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
        // offline: the fake JACK server pushes seeded noise through every instance (its own stream of noise each), all
        // instances at once unless --sequential
        struct Feed {
            std::mt19937 rng;
            std::uniform_real_distribution<float> u{-0.25f, 0.25f};
            std::vector<float> rec[4];  // in1, in2, outL, outR when dumping
            bool keep = false;
        };
        std::vector<Feed> feeds(instances.size());
        std::vector<void*> users;
        std::vector<jack_client_t*> clients;
        for (size_t i = 0; i < instances.size(); i++) {
            feeds[i].rng.seed(1234 + 17 * (unsigned)i);
            feeds[i].keep = dump != nullptr;
            users.push_back(&feeds[i]);
            clients.push_back(instances[i]->handle);
        }
        auto fill = [](uint64_t, float** bufs, size_t n, jack_nframes_t nframes, void* user) {
            auto* f = static_cast<Feed*>(user);
            for (size_t b = 0; b < n; b++)
                for (jack_nframes_t s = 0; s < nframes; s++) {
                    bufs[b][s] = f->u(f->rng) + 0.01f;
                    if (f->keep && b < 2) f->rec[b].push_back(bufs[b][s]);
                }
        };
        auto keep = [](uint64_t, float** bufs, size_t n, jack_nframes_t nframes, void* user) {
            auto* f = static_cast<Feed*>(user);
            if (!f->keep) return;
            for (size_t b = 0; b < n && b < 2; b++) f->rec[2 + b].insert(f->rec[2 + b].end(), bufs[b], bufs[b] + nframes);
        };
        std::vector<double> us(instances.size(), 0.0);
        if (sequential)
            for (size_t i = 0; i < instances.size(); i++) fakejack_run_all(&clients[i], 1, periods, fill, keep, &users[i], spacing_us, &us[i]);
        else
            fakejack_run_all(clients.data(), clients.size(), periods, fill, keep, users.data(), spacing_us, us.data());
        for (size_t i = 0; i < instances.size(); i++) {
            Log::info(instances[i]->name, "%s: %.2f us per period inside the process callback (%llu periods)", sequential ? "alone" : "concurrent",
                      us[i], (unsigned long long)periods);
            if (dump) {
                static const char* ext[4] = {"in1", "in2", "outL", "outR"};
                for (int k = 0; k < 4; k++) {
                    std::ofstream f(std::string(dump) + std::to_string(i) + "." + ext[k], std::ios::binary);
                    f.write(reinterpret_cast<const char*>(feeds[i].rec[k].data()), (std::streamsize)(feeds[i].rec[k].size() * sizeof(float)));
                }
            }
        }
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
