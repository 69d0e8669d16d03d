// host_tool.cpp — small CLI over the host-side helpers for the CPU tests
// (no GPU call is made by any command here).
//   wavdump <in.wav> <out.f32>        decode with WavFile, write interleaved float32 frames
//   wavwrite <in.f32> <out.wav> <bits> encode interleaved float32 frames with WavFile::write
//   settings <file> <key>...          print "key=value" for each printf-free key
//   midi <hexbytes>                   feed bytes to a RawMidi::Device, print dispatched messages
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "log.h"
#include "midi.h"
#include "settings.h"
#include "wav.h"

struct Printer : RawMidi::MessageHandler {
    void onMidiMessage(const RawMidi::Device*, const uint8_t* b, size_t len) override {
        printf("msg");
        for (size_t i = 0; i < len; i++) printf(" %u", b[i]);
        printf("\n");
    }
};

int main(int argc, char** argv) {
    Log::quiet(true);
    if (argc >= 4 && !strcmp(argv[1], "wavdump")) {
        WavFile w(argv[2]);
        FILE* f = fopen(argv[3], "wb");
        if (!f) return 1;
        fwrite(w.buffer, sizeof(wav_float2), w.numFrames, f);
        fclose(f);
        printf("frames=%zu rate=%u bits=%u\n", w.numFrames, w.sampleRate, w.bitsPerSample);
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "wavwrite")) {
        std::ifstream is(argv[2], std::ifstream::binary);
        std::vector<char> raw((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
        const size_t frames = raw.size() / (2 * sizeof(float));
        return WavFile::write(argv[3], reinterpret_cast<const float*>(raw.data()), frames, (unsigned)atoi(argv[4])) ? 0 : 1;
    }
    if (argc >= 3 && !strcmp(argv[1], "settings")) {
        Settings s;
        s.open(argv[2]);
        for (int i = 3; i < argc; i++) {
            try {
                printf("%s=%s\n", argv[i], s.str("%s", argv[i]).c_str());
            } catch (std::exception&) {
                printf("%s=<error>\n", argv[i]);
            }
        }
        try {
            printf("u32(conv.count)=%u\n", s.u32("conv.count"));
            printf("f32(conv[%d].value.dry)=%g\n", 1, s.f32("conv[%d].value.dry", 1));
        } catch (std::exception& e) {
            printf("throw\n");
        }
        try {
            (void)s.u32("no.such.key");
            printf("missing=ok\n");
        } catch (std::exception&) {
            printf("missing=throw\n");
        }
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "midi")) {
        RawMidi::Device d("fake:0");
        Printer p;
        d.handler = &p;
        std::vector<uint8_t> bytes;
        const char* h = argv[2];
        for (size_t i = 0; i + 1 < strlen(h); i += 2) {
            char t[3] = {h[i], h[i + 1], 0};
            bytes.push_back((uint8_t)strtoul(t, nullptr, 16));
        }
        d.feed(bytes.data(), bytes.size());
        return 0;
    }
    fprintf(stderr, "unknown command\n");
    return 2;
}
