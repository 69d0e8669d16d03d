// log.h — timestamped ANSI logging with the reference's call surface
// (reference src/log.h:28-47: Log::info / warn / error / newline, ESC()).
#pragma once
#include <string>

#ifndef ESC
#define ESC(n) "\x1b[" #n "m"
#endif

class Log {
public:
    static void info(const std::string& id, const char* fmt, ...) noexcept __attribute__((format(printf, 2, 3)));
    static void warn(const std::string& id, const char* fmt, ...) noexcept __attribute__((format(printf, 2, 3)));
    static void error(const std::string& id, const char* fmt, ...) noexcept __attribute__((format(printf, 2, 3)));
    static void newline() noexcept;
    static void newline(const char* fmt, ...) noexcept __attribute__((format(printf, 1, 2)));
    static void quiet(bool q) noexcept;  // tests: silence stdout logging
};
