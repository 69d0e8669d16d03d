// settings.h — `key value` settings file with printf-keyed typed getters,
// same call surface as the reference (src/settings.h:24-37, settings.cu:4-24):
// whitespace-separated pairs, tokens starting with '#' comment out the rest of
// the line, a missing key makes the typed getters throw (std::stoi/stof on an
// empty string, as in the reference).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <istream>
#include <map>
#include <string>

class Setting {
public:
    std::string key;
    std::string value;
    bool isTrue() const { return value == "yes" || value == "true"; }
    bool isFalse() const { return !isTrue(); }
    uint8_t u8() const { return (uint8_t)(std::stoi(value) & 0xFF); }
    uint16_t u16() const { return (uint16_t)(std::stoi(value) & 0xFFFF); }
    uint32_t u32() const { return (uint32_t)std::stoi(value); }
    float f32() const { return std::stof(value); }
    const std::string& str() const { return value; }
};

class Settings : public std::map<std::string, Setting> {
public:
    void open(const std::string& path);
    void parse(std::istream& is);  // same grammar from any stream (tests)

    bool isTrue(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    bool isFalse(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    uint8_t u8(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    uint16_t u16(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    uint32_t u32(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    float f32(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    const std::string& str(const char* fmt, ...) __attribute__((format(printf, 2, 3)));

private:
    Setting& lookup(const char* fmt, va_list ap, std::string* keyOut);
};
