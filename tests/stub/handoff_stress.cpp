// handoff_stress.cpp - test infrastructure: csrc/params_handoff.h alone, under -fsanitize=thread (no HIP, no GPU).
// Two writer threads publish pairs whose every field is a function of one counter; a reader thread samples as fast as it
// can (the process path) and checks that every pair it sees is ONE published pair (no mixture of two), that generations
// never go back, and that the vsteps count-down is a compare-exchange against the sampled value: a reset that arrives
// between sample and count-down survives.  Prints "ok <samples> <generations>" or a diagnostic and exits non-zero.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../cuda_audio_amd/csrc/params_handoff.h"

static void fill(mc_cc_value& v, uint64_t k, int half) {
    v.select = k;
    v.predelay = k * 3 + (uint64_t)half;
    v.speed = k ^ 0x5555;
    v.dry = (float)(k & 0xffff) * 0.5f;
    v.wet = (float)(k & 0xffff) + 1.0f;
    v.panDry = -(float)(k & 0xfff);
    v.panWet = (float)((k >> 3) & 0xffff);
    v.level = (float)(k & 0xff) * 0.25f;
}
static bool consistent(const mc_cc_value& v, int half) {
    mc_cc_value w;
    fill(w, v.select, half);
    return v.predelay == w.predelay && v.speed == w.speed && v.dry == w.dry && v.wet == w.wet && v.panDry == w.panDry && v.panWet == w.panWet &&
           v.level == w.level;
}

int main(int argc, char** argv) {
    const long rounds = argc > 1 ? atol(argv[1]) : 200000;
    ParamHandoff ph;
    ph.update([](mc_cc_value (&cc)[2]) {
        fill(cc[0], 0, 0);
        fill(cc[1], 0, 1);
    });
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> counter{1};
    auto writer = [&](int id) {
        while (!stop.load(std::memory_order_relaxed)) {
            const uint64_t k = counter.fetch_add(1, std::memory_order_relaxed);
            if (id == 0)
                ph.update([&](mc_cc_value (&cc)[2]) {  // both halves from one counter: a pair is one publish
                    fill(cc[0], k, 0);
                    fill(cc[1], k, 1);
                });
            else
                ph.update([&](mc_cc_value (&cc)[2]) {  // a select with its reset (conv.cu:260-261), the other half untouched
                    fill(cc[1], cc[1].select, 1);
                    cc[0].vsteps = 1000 + (k & 7);
                });
        }
    };
    std::thread w0(writer, 0), w1(writer, 1);
    uint64_t last_gen = 0, samples = 0;
    int bad = 0;
    for (long r = 0; r < rounds && !bad; r++) {
        mc_cc_value cc[2];
        const uint64_t g = ph.sample(cc);
        samples++;
        if (g < last_gen) {
            fprintf(stderr, "generation went back: %llu after %llu\n", (unsigned long long)g, (unsigned long long)last_gen);
            bad = 1;
        }
        last_gen = g;
        if (!consistent(cc[0], 0) || !consistent(cc[1], 1) || cc[0].select != cc[1].select) {
            fprintf(stderr, "mixed pair at generation %llu: selects %llu / %llu\n", (unsigned long long)g, (unsigned long long)cc[0].select,
                    (unsigned long long)cc[1].select);
            bad = 1;
        }
        ph.count_down(0, cc[0].vsteps, cc[0].vsteps ? 1 : 0);  // the process path's count-down beside the writers' resets
    }
    stop.store(true);
    w0.join();
    w1.join();
    // the compare-exchange, deterministically: a reset between sample and count-down survives
    {
        mc_cc_value cc[2];
        ph.update([](mc_cc_value (&c)[2]) { c[0].vsteps = 7; });
        ph.sample(cc);
        if (cc[0].vsteps != 7) bad = 1, fprintf(stderr, "vsteps not handed over\n");
        ph.update([](mc_cc_value (&c)[2]) { c[0].vsteps = 100; });  // a select arrives while the period runs
        ph.count_down(0, 7, 1);                                      // the period that sampled 7
        ph.sample(cc);
        if (cc[0].vsteps != 100) bad = 1, fprintf(stderr, "the reset was undone: %llu\n", (unsigned long long)cc[0].vsteps);
        ph.count_down(0, 100, 1);
        ph.sample(cc);
        if (cc[0].vsteps != 99) bad = 1, fprintf(stderr, "count-down lost: %llu\n", (unsigned long long)cc[0].vsteps);
        ph.count_down(0, 99, 200);  // a batch longer than the steps left
        ph.sample(cc);
        if (cc[0].vsteps != 0) bad = 1, fprintf(stderr, "count-down below zero: %llu\n", (unsigned long long)cc[0].vsteps);
    }
    if (bad) return 1;
    printf("ok %llu %llu\n", (unsigned long long)samples, (unsigned long long)last_gen);
    return 0;
}
