// params_handoff.h - how controller values reach the process path ("mc_set_params callable concurrently,
// double-buffered POD swap, no locks on the process path").  Plain C++17, no HIP: tests/test_params_handoff.py builds
// it alone under -fsanitize=thread.
//
// The reference has no synchronisation at all here: main() and the MIDI thread store into the public cc[i].value
// fields (main.cu:49-70, conv.cu:255-276, midi.cu:22-59) while the JACK thread reads them inside onProcess
// (conv.cu:339-353, 386-427).  Round 2 put a std::mutex around the engine's copy, taken twice per period on the JACK
// thread - a real-time thread that can block behind a preempted controller thread.
//
// Here the process path takes no lock and never waits for a writer:
//   * writers (mc_set_params, mc_handle_cc; any thread) serialise among themselves with a mutex, build the new pair of
//     values in the slot that is NOT current, and publish it by flipping `cur` (release);
//   * the reader (the start of every process call) copies the current slot and checks the slot's sequence word
//     before and after: it retries only if a writer went round TWICE during the copy - a writer that is stalled in the
//     middle of a write holds the other slot and is never looked at;
//   * every published pair carries a generation number, returned with the sample (tests: every period ran on exactly
//     one published pair, never a mixture);
//   * vsteps - the one field the process path writes (its count-down, conv.cu:345,353) - lives outside the slots as an
//     atomic: the count-down is a compare-exchange against the value the period sampled, so a select that arrived
//     meanwhile (vsteps = speed, conv.cu:261) survives, as the reference's in-place decrement would let it.
// All slot words are relaxed atomics (no data race in the C++ sense, nothing for a sanitizer to flag); the ordering
// comes from the sequence words and `cur`.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>

#include "../../include/mcconv.h"

struct ParamHandoff {
    static constexpr int kWords = (2 * sizeof(mc_cc_value) + 7) / 8;
    struct Slot {
        std::atomic<uint32_t> seq{0};  // odd while a writer is inside
        std::atomic<uint64_t> gen{0};
        std::atomic<uint64_t> w[kWords];
    };
    Slot slot[2];
    std::atomic<uint32_t> cur{0};
    std::atomic<uint64_t> vsteps[2];
    std::mutex wmu;  // writers only

    ParamHandoff() {
        for (auto& s : slot)
            for (auto& x : s.w) x.store(0, std::memory_order_relaxed);
        vsteps[0].store(0, std::memory_order_relaxed);
        vsteps[1].store(0, std::memory_order_relaxed);
    }

    // ---- writers -----------------------------------------------------------------------------------------------
    // f(cc[2]) edits the current pair in place; fields it leaves alone keep their values.  Returns the generation
    // of the pair it published (an edit that changes nothing publishes nothing).  vsteps: if f changed cc[i].vsteps, the new value is stored (a select's reset, or
    // mc_set_params handing the host's count over); otherwise the process path's count-down is left alone.
    template <class F>
    uint64_t update(F&& f) {
        std::lock_guard<std::mutex> lk(wmu);
        const uint32_t c = cur.load(std::memory_order_relaxed);
        mc_cc_value cc[2];
        load_words(slot[c], cc);
        uint64_t vs_before[2];
        for (int i = 0; i < 2; i++) cc[i].vsteps = vs_before[i] = vsteps[i].load(std::memory_order_acquire);
        mc_cc_value before[2] = {cc[0], cc[1]};
        f(cc);
        if (same(before[0], cc[0]) && same(before[1], cc[1])) return slot[c].gen.load(std::memory_order_relaxed);  // nothing moved: nothing to publish
        for (int i = 0; i < 2; i++)
            if (cc[i].vsteps != vs_before[i]) vsteps[i].store(cc[i].vsteps, std::memory_order_release);  // before the pair: a reader that sees the new select sees the reset
        Slot& n = slot[1 - c];
        const uint64_t g = slot[c].gen.load(std::memory_order_relaxed) + 1;
        n.seq.fetch_add(1, std::memory_order_acq_rel);  // odd: being written
        std::atomic_thread_fence(std::memory_order_release);
        uint64_t raw[kWords] = {};
        mc_cc_value tmp[2] = {cc[0], cc[1]};
        tmp[0].vsteps = tmp[1].vsteps = 0;  // (not part of the pair)
        std::memcpy(raw, tmp, sizeof(tmp));
        for (int k = 0; k < kWords; k++) n.w[k].store(raw[k], std::memory_order_relaxed);
        n.gen.store(g, std::memory_order_relaxed);
        n.seq.fetch_add(1, std::memory_order_release);  // even: complete
        cur.store(1 - c, std::memory_order_release);
        return g;
    }

    // ---- reader: the process path (one thread at a time per engine), and mc_get_params from anywhere ---------------
    // lock-free; returns the generation of the pair copied into cc (cc[i].vsteps = the live count)
    uint64_t sample(mc_cc_value (&cc)[2]) const {
        for (;;) {
            const uint32_t c = cur.load(std::memory_order_acquire);
            const Slot& s = slot[c];
            const uint32_t s0 = s.seq.load(std::memory_order_acquire);
            if (s0 & 1u) continue;  // (the writer came round to this slot again: cur has moved on, look again)
            load_words(s, cc);
            const uint64_t g = s.gen.load(std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_acquire);
            if (s.seq.load(std::memory_order_relaxed) != s0) continue;
            for (int i = 0; i < 2; i++) cc[i].vsteps = vsteps[i].load(std::memory_order_acquire);
            return g;
        }
    }

    // the period that sampled `sampled` steps used `used` of them: written back only if nobody reset the count meanwhile
    void count_down(int half, uint64_t sampled, uint64_t used) {
        if (!used) return;
        uint64_t expect = sampled;
        vsteps[half].compare_exchange_strong(expect, sampled - (used < sampled ? used : sampled), std::memory_order_acq_rel, std::memory_order_relaxed);
    }

private:
    static bool same(const mc_cc_value& a, const mc_cc_value& b) {  // (field by field: the struct has padding)
        return a.select == b.select && a.predelay == b.predelay && a.speed == b.speed && a.vsteps == b.vsteps && a.dry == b.dry && a.wet == b.wet &&
               a.panDry == b.panDry && a.panWet == b.panWet && a.level == b.level;
    }
    static void load_words(const Slot& s, mc_cc_value (&cc)[2]) {
        uint64_t raw[kWords];
        for (int k = 0; k < kWords; k++) raw[k] = s.w[k].load(std::memory_order_relaxed);
        std::memcpy(cc, raw, sizeof(mc_cc_value) * 2);
    }
};
