#include "wav.h"

#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>

#include "log.h"

namespace {
uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
}  // namespace

WavFile::WavFile(const std::string& label, const float* lr, size_t frames) : path(label), numFrames(frames) {
    storage.resize(frames);
    for (size_t i = 0; i < frames; i++) storage[i] = {lr[2 * i], lr[2 * i + 1]};
    buffer = storage.data();
}

WavFile::WavFile(const std::string& p) : path(p) {
    std::ifstream is(p, std::ifstream::binary);
    assert(is.good() && "cannot open WAV file");
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    assert(file.size() >= 12 && !memcmp(file.data(), "RIFF", 4) && !memcmp(file.data() + 8, "WAVE", 4));
    unsigned channels = 0, blockAlign = 0, byteRate = 0;
    const unsigned char* data = nullptr;
    size_t dataBytes = 0;
    for (size_t off = 12; off + 8 <= file.size();) {
        const unsigned char* ck = file.data() + off;
        size_t sz = rd32(ck + 4);
        if (off + 8 + sz > file.size()) sz = file.size() - off - 8;
        if (!memcmp(ck, "fmt ", 4) && sz >= 16) {
            channels = rd16(ck + 10);
            sampleRate = rd32(ck + 12);
            byteRate = rd32(ck + 16);
            blockAlign = rd16(ck + 20);
            bitsPerSample = rd16(ck + 22);
        } else if (!memcmp(ck, "data", 4)) {
            data = ck + 8;
            dataBytes = sz;
            break;
        }
        off += 8 + sz + (sz & 1);
    }
    assert(data && channels == 2 && "only stereo PCM is supported (reference src/wav.cu:103-114)");
    assert((bitsPerSample == 16 && blockAlign == 4) || (bitsPerSample == 24 && blockAlign == 6));
    numFrames = dataBytes / blockAlign;
    Log::info("wav", "IR [%0.2f s] %s", byteRate ? dataBytes / (float)byteRate : 0.f, p.c_str());
    storage.resize(numFrames);
    if (bitsPerSample == 16) {
        for (size_t i = 0; i < numFrames; i++) {
            const int16_t l = (int16_t)rd16(data + 4 * i), r = (int16_t)rd16(data + 4 * i + 2);
            storage[i] = {l / 65536.0f, r / 65536.0f};
        }
    } else {
        for (size_t i = 0; i < numFrames; i++) {
            int32_t v[2];
            for (int c = 0; c < 2; c++) {
                const unsigned char* s = data + 6 * i + 3 * c;
                const uint32_t u = ((uint32_t)s[0] << 8) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 24);
                v[c] = (int32_t)u / 256;  // sign-extended 24-bit value
            }
            storage[i] = {v[0] / 16777216.0f, v[1] / 16777216.0f};
        }
    }
    buffer = storage.data();
}

bool WavFile::write(const std::string& p, const float* lr, size_t frames, unsigned bits, unsigned rate) {
    if (bits != 16 && bits != 24) return false;
    const unsigned bytes = bits / 8, align = 2 * bytes;
    const uint32_t dataBytes = (uint32_t)(frames * align);
    std::ofstream os(p, std::ofstream::binary);
    if (!os.good()) return false;
    auto w32 = [&](uint32_t v) { os.write((const char*)&v, 4); };
    auto w16 = [&](uint16_t v) { os.write((const char*)&v, 2); };
    os.write("RIFF", 4);
    w32(36 + dataBytes);
    os.write("WAVEfmt ", 8);
    w32(16);
    w16(1);
    w16(2);
    w32(rate);
    w32(rate * align);
    w16((uint16_t)align);
    w16((uint16_t)bits);
    os.write("data", 4);
    w32(dataBytes);
    // inverse of the loader's scaling: full scale +-0.5
    const double scale = bits == 16 ? 65536.0 : 16777216.0;
    const double hi = bits == 16 ? 32767.0 : 8388607.0, lo = -hi - 1.0;
    for (size_t i = 0; i < 2 * frames; i++) {
        double q = std::nearbyint((double)lr[i] * scale);
        q = q > hi ? hi : (q < lo ? lo : q);
        const int32_t v = (int32_t)q;
        os.write((const char*)&v, bytes);  // little endian
    }
    return os.good();
}
