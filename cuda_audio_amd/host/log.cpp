#include "log.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <mutex>

namespace {
std::mutex g_mu;
bool g_quiet = false;

void emit(FILE* to, const char* colour, const char* tag, const std::string& id, const char* fmt, va_list ap) {
    if (g_quiet && to == stdout) return;
    char body[512];
    vsnprintf(body, sizeof(body), fmt, ap);
    const auto now = std::chrono::system_clock::now();
    const std::time_t tt = std::chrono::system_clock::to_time_t(now);
    const long ms = (long)(std::chrono::duration_cast<std::chrono::milliseconds>(now.time_since_epoch()).count() % 1000);
    std::tm tmv;
    localtime_r(&tt, &tmv);
    std::lock_guard<std::mutex> lk(g_mu);
    fprintf(to, "%s%02d:%02d:%02d.%03ld %-5s" ESC(0) " [%s] %s" ESC(0) "\n", colour, tmv.tm_hour, tmv.tm_min, tmv.tm_sec, ms,
            tag, id.c_str(), body);
    fflush(to);
}
}  // namespace

void Log::quiet(bool q) noexcept { g_quiet = q; }

void Log::info(const std::string& id, const char* fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    emit(stdout, ESC(36), "info", id, fmt, ap);
    va_end(ap);
}

void Log::warn(const std::string& id, const char* fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    emit(stderr, ESC(33;1), "warn", id, fmt, ap);
    va_end(ap);
}

void Log::error(const std::string& id, const char* fmt, ...) noexcept {
    va_list ap;
    va_start(ap, fmt);
    emit(stderr, ESC(31;1), "error", id, fmt, ap);
    va_end(ap);
}

void Log::newline() noexcept {
    if (g_quiet) return;
    std::lock_guard<std::mutex> lk(g_mu);
    fputc('\n', stdout);
}

void Log::newline(const char* fmt, ...) noexcept {
    if (g_quiet) return;
    char body[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(body, sizeof(body), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_mu);
    fprintf(stdout, "                   %s" ESC(0) "\n", body);
}
