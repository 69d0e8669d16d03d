// host_tool.cpp — small CLI over the host-side helpers for the CPU tests
// (no GPU call is made by any command here).
//   wavdump <in.wav> <out.f32>        decode with WavFile, write interleaved float32 frames
//   wavwrite <in.f32> <out.wav> <bits> encode interleaved float32 frames with WavFile::write
//   settings <file> <key>...          print "key=value" for each printf-free key
//   midi <hexbytes>                   feed bytes to a RawMidi::Device, print dispatched messages
//   operators                         every operator of operators.h on fixed operands, one result per line
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "log.h"
#include "midi.h"
#include "settings.h"
#include "operators.h"
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
    if (argc >= 2 && !strcmp(argv[1], "operators")) {
        // the reference's kernels index with dim3 arithmetic (conv.cu:18-19 and every other kernel) and do their
        // float2 arithmetic with these operators (conv.cu:25-30, 58-70, 95-98, 134-137)
        const host_dim3 blockDim{256, 1, 1}, gridDim{64, 2, 1};
        const host_uint3 blockIdx{5, 1, 0}, threadIdx{17, 0, 0};
        const host_dim3 offset = blockDim * blockIdx + threadIdx, stride = blockDim * gridDim;
        printf("offset=%u,%u,%u stride=%u,%u,%u\n", offset.x, offset.y, offset.z, stride.x, stride.y, stride.z);
        const wav_float2 a{1.5f, -2.0f}, b{0.25f, 4.0f};
        wav_float2 r = a + b;
        printf("add=%g,%g\n", r.x, r.y);
        r = a - b;
        printf("sub=%g,%g\n", r.x, r.y);
        r = a * 2.0f;
        printf("mul=%g,%g\n", r.x, r.y);
        r = 2.0f * a;
        printf("lmul=%g,%g\n", r.x, r.y);
        r = a / 4.0f;
        printf("div=%g,%g\n", r.x, r.y);
        r = a;
        r += b;
        printf("addeq=%g,%g\n", r.x, r.y);
        r += 0.5f;
        printf("addeqs=%g,%g\n", r.x, r.y);
        r = clamp(wav_float2{3.0f, -3.0f}, -1.0f, 1.0f);
        printf("clamp=%g,%g\n", r.x, r.y);
        return 0;
    }
    fprintf(stderr, "unknown command\n");
    return 2;
}
