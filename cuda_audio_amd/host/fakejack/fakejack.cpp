// fakejack.cpp — in-process stand-in for jackd (see fakejack/jack/jack.h).
#include <jack/jack.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct fake_jack_port {
    std::string full_name;
    unsigned long flags = 0;
    std::vector<float> buf;
};

struct fake_jack_client {
    std::string name;
    JackProcessCallback process = nullptr;
    void* process_arg = nullptr;
    JackShutdownCallback shutdown = nullptr;
    void* shutdown_arg = nullptr;
    bool active = false;
    jack_nframes_t rate = 44100, period = 256;
    std::vector<fake_jack_port*> ports;
    std::vector<std::pair<std::string, std::string>> connections;
};

namespace {
jack_nframes_t g_rate = 44100, g_period = 256;
}

extern "C" {

void fakejack_configure(jack_nframes_t sample_rate, jack_nframes_t period) {
    g_rate = sample_rate;
    g_period = period;
}

jack_client_t* jack_client_open(const char* name, jack_options_t, jack_status_t* status, ...) {
    auto* c = new fake_jack_client();
    c->name = name ? name : "client";
    c->rate = g_rate;
    c->period = g_period;
    if (status) *status = 0;
    return c;
}

int jack_client_close(jack_client_t* c) {
    if (!c) return -1;
    for (auto* p : c->ports) delete p;
    delete c;
    return 0;
}

int jack_set_process_callback(jack_client_t* c, JackProcessCallback cb, void* arg) {
    if (!c) return -1;
    c->process = cb;
    c->process_arg = arg;
    return 0;
}

void jack_on_shutdown(jack_client_t* c, JackShutdownCallback cb, void* arg) {
    if (!c) return;
    c->shutdown = cb;
    c->shutdown_arg = arg;
}

jack_nframes_t jack_get_sample_rate(jack_client_t* c) { return c ? c->rate : 0; }
jack_nframes_t jack_get_buffer_size(jack_client_t* c) { return c ? c->period : 0; }

jack_port_t* jack_port_register(jack_client_t* c, const char* port_name, const char*, unsigned long flags, unsigned long) {
    if (!c || !port_name) return nullptr;
    auto* p = new fake_jack_port();
    p->full_name = c->name + ":" + port_name;
    p->flags = flags;
    p->buf.assign(c->period, 0.f);
    c->ports.push_back(p);
    return p;
}

void* jack_port_get_buffer(jack_port_t* p, jack_nframes_t nframes) {
    if (!p) return nullptr;
    if (p->buf.size() < nframes) p->buf.resize(nframes, 0.f);
    return p->buf.data();
}

const char* jack_port_name(const jack_port_t* p) { return p ? p->full_name.c_str() : ""; }

int jack_activate(jack_client_t* c) {
    if (!c) return -1;
    c->active = true;
    return 0;
}

int jack_connect(jack_client_t* c, const char* src, const char* dst) {
    if (!c || !src || !dst) return -1;
    c->connections.emplace_back(src, dst);
    return 0;
}

uint64_t fakejack_run(jack_client_t* c, uint64_t nperiods, fakejack_io_fn feed, fakejack_io_fn drain, void* user) {
    if (!c || !c->active || !c->process) return 0;
    std::vector<float*> ins, outs;
    for (auto* p : c->ports) {
        if (p->buf.size() < c->period) p->buf.resize(c->period, 0.f);
        (p->flags & JackPortIsInput ? ins : outs).push_back(p->buf.data());
    }
    uint64_t done = 0;
    for (; done < nperiods; done++) {
        if (feed) feed(done, ins.data(), ins.size(), c->period, user);
        if (c->process(c->period, c->process_arg) != 0) break;
        if (drain) drain(done, outs.data(), outs.size(), c->period, user);
    }
    return done;
}

uint64_t fakejack_run_all(jack_client_t** clients, size_t nclients, uint64_t nperiods, fakejack_io_fn feed, fakejack_io_fn drain,
                          void** users, double spacing_us, double* us_per_period) {
    if (!clients || !nclients) return 0;
    using clock = std::chrono::steady_clock;
    std::vector<uint64_t> done(nclients, 0);
    std::vector<double> busy(nclients, 0.0);
    std::atomic<int> ready{0};
    const auto t0 = clock::now() + std::chrono::milliseconds(2);
    auto drive = [&](size_t i) {
        jack_client_t* c = clients[i];
        if (!c || !c->active || !c->process) return;
        std::vector<float*> ins, outs;
        for (auto* p : c->ports) {
            if (p->buf.size() < c->period) p->buf.resize(c->period, 0.f);
            (p->flags & JackPortIsInput ? ins : outs).push_back(p->buf.data());
        }
        ready.fetch_add(1);
        while (ready.load() < (int)nclients) std::this_thread::yield();  // all clients start together
        for (uint64_t k = 0; k < nperiods; k++) {
            if (spacing_us > 0) {
                const auto due = t0 + std::chrono::nanoseconds((int64_t)(k * spacing_us * 1e3));
                while (clock::now() < due) {
                }  // (a period clock: spin, as the benchmark's spaced mode does)
            }
            if (feed) feed(k, ins.data(), ins.size(), c->period, users ? users[i] : nullptr);
            const auto a = clock::now();
            const int rc = c->process(c->period, c->process_arg);
            busy[i] += std::chrono::duration<double, std::micro>(clock::now() - a).count();
            if (rc != 0) break;
            if (drain) drain(k, outs.data(), outs.size(), c->period, users ? users[i] : nullptr);
            done[i] = k + 1;
        }
    };
    std::vector<std::thread> th;
    for (size_t i = 0; i < nclients; i++) th.emplace_back(drive, i);
    for (auto& t : th) t.join();
    uint64_t all = nperiods;
    for (size_t i = 0; i < nclients; i++) {
        all = done[i] < all ? done[i] : all;
        if (us_per_period) us_per_period[i] = done[i] ? busy[i] / (double)done[i] : 0.0;
    }
    return all;
}

void fakejack_shutdown(jack_client_t* c) {
    if (c && c->shutdown) c->shutdown(c->shutdown_arg);
}

}  // extern "C"
