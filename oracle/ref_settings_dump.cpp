// ref_settings_dump.cpp - test infrastructure (own code, not reference source).
// A dumper over the `Settings` class surface that the reference (src/settings.h:24-37) and this repository's host
// (cuda_audio_amd/host/settings.h) share.  oracle/Makefile compiles it twice:
//   _ref/ref_settings_dump   with -I/root/reference/src and the reference's own settings.cu + log.cu, compiled
//                            host-only where they lie (these two files need no CUDA);
//   _ref/host_settings_dump  with -I../cuda_audio_amd/host and the host's settings.cpp + log.cpp.
// tests/test_reference_pins.py runs both on the same files and requests and compares the outputs byte for byte.
//
//   <dumper> <settings file> <out file>      requests on stdin, one per line: "<type> <key>", type one of
//                                            str u8 u16 u32 f32 isTrue isFalse
// Output: "<type> <key> = <value>" or "<type> <key> ! throw" per request, then the whole map ("entry <key> <value>"):
// a getter on a missing key inserts an empty entry (operator[], settings.cu:49) and the dump shows it.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <iostream>
#include <string>

#include "settings.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    Settings s;
    s.open(argv[1]);
    FILE* out = fopen(argv[2], "w");
    if (!out) return 3;
    std::string type, key;
    while (std::cin >> type >> key) {
        try {
            if (type == "str") {
                const std::string v = s.str("%s", key.c_str());
                fprintf(out, "str %s = [%s]\n", key.c_str(), v.c_str());
            } else if (type == "u8") {
                fprintf(out, "u8 %s = %u\n", key.c_str(), (unsigned)s.u8("%s", key.c_str()));
            } else if (type == "u16") {
                fprintf(out, "u16 %s = %u\n", key.c_str(), (unsigned)s.u16("%s", key.c_str()));
            } else if (type == "u32") {
                fprintf(out, "u32 %s = %u\n", key.c_str(), (unsigned)s.u32("%s", key.c_str()));
            } else if (type == "f32") {
                const float f = s.f32("%s", key.c_str());
                uint32_t bits;
                memcpy(&bits, &f, 4);
                fprintf(out, "f32 %s = %.9g (0x%08x)\n", key.c_str(), (double)f, bits);
            } else if (type == "isTrue") {
                fprintf(out, "isTrue %s = %d\n", key.c_str(), (int)s.isTrue("%s", key.c_str()));
            } else if (type == "isFalse") {
                fprintf(out, "isFalse %s = %d\n", key.c_str(), (int)s.isFalse("%s", key.c_str()));
            } else {
                fprintf(out, "%s %s ? unknown type\n", type.c_str(), key.c_str());
            }
        } catch (std::exception&) {
            fprintf(out, "%s %s ! throw\n", type.c_str(), key.c_str());
        }
    }
    fprintf(out, "size %zu\n", s.size());
    for (auto& kv : s) fprintf(out, "entry %s [%s] key=[%s]\n", kv.first.c_str(), kv.second.value.c_str(), kv.second.key.c_str());
    fclose(out);
    return 0;
}
