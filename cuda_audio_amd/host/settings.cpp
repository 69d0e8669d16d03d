#include "settings.h"

#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <istream>
#include <stdexcept>

#include "log.h"

void Settings::parse(std::istream& is) {
    std::string key, value;
    while (is >> key) {
        if (key[0] == '#') {
            std::string rest;
            std::getline(is, rest);
            continue;
        }
        if (!(is >> value)) value.clear();
        (*this)[key] = Setting{key, value};
        Log::info("Settings", "%-24s " ESC(36;1) "%s", key.c_str(), value.c_str());
    }
}

void Settings::open(const std::string& path) {
    std::ifstream is(path, std::ifstream::binary);
    parse(is);
}

Setting& Settings::lookup(const char* fmt, va_list ap, std::string* keyOut) {
    char buf[256];
    vsnprintf(buf, sizeof(buf), fmt, ap);
    *keyOut = buf;
    return (*this)[*keyOut];  // like the reference: a missing key yields an empty value
}

#define SETTINGS_GETTER(ret, name, expr)                               \
    ret Settings::name(const char* fmt, ...) {                        \
        va_list ap;                                                    \
        va_start(ap, fmt);                                             \
        std::string key;                                               \
        Setting& s = lookup(fmt, ap, &key);                            \
        va_end(ap);                                                    \
        try {                                                          \
            return expr;                                               \
        } catch (std::exception&) {                                    \
            Log::error("Settings", "Error for key %s", key.c_str());   \
            throw;                                                     \
        }                                                              \
    }

SETTINGS_GETTER(bool, isTrue, s.isTrue())
SETTINGS_GETTER(bool, isFalse, s.isFalse())
SETTINGS_GETTER(uint8_t, u8, s.u8())
SETTINGS_GETTER(uint16_t, u16, s.u16())
SETTINGS_GETTER(uint32_t, u32, s.u32())
SETTINGS_GETTER(float, f32, s.f32())
SETTINGS_GETTER(const std::string&, str, s.str())
