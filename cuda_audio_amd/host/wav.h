// wav.h — WAV impulse-response loader with the reference's surface
// (reference src/wav.h:5-13: path, numFrames, float2* buffer).  The reference
// decodes on the GPU into a device buffer; this loader decodes on the host
// (load-time only) into a host buffer that Convolution::prepare hands to
// mc_load_ir.  Scaling is the reference's (src/wav.cu:17-57, quirk Q5):
// 16-bit samples / 65536, 24-bit samples / 2^24, i.e. full scale = +-0.5.
// Superset: chunks are searched by id instead of assumed back-to-back.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

struct wav_float2 {
    float x, y;  // left, right
};

class WavFile {
public:
    std::string path;
    size_t numFrames = 0;
    wav_float2* buffer = nullptr;  // host memory, numFrames frames (owned)
    unsigned sampleRate = 0, bitsPerSample = 0;

    explicit WavFile(const std::string& path);
    // from already decoded frames (tests / synthetic IRs)
    WavFile(const std::string& label, const float* interleavedLR, size_t frames);
    ~WavFile() = default;
    WavFile(const WavFile&) = delete;
    WavFile& operator=(const WavFile&) = delete;

    // write a stereo PCM file (16 or 24 bit) — used by tests to make fixtures
    static bool write(const std::string& path, const float* interleavedLR, size_t frames, unsigned bits, unsigned rate = 44100);

private:
    std::vector<wav_float2> storage;
};
