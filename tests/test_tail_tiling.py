"""The tiling of the JACK tail's direct convolution (csrc/jack_tail.hip.h: td_unit, td_window, td_tile), checked on the CPU.

Partition 0 of a 256-frame period is summed in the time domain: out[m] = sum_{j <= m} h[j] x[m - j].  A unit is NO consecutive output
frames x NT consecutive taps, one unit per thread; the constants are read from the header, the index arithmetic is restated here as
the kernel has it, and the test checks what the kernel relies on: every (frame, tap) pair of the triangle is summed exactly once, the
units fit the threads that take them, a unit's window stays inside the zero padding in front of the period, the partial sums of a
frame fit the slots, and the same holds for the second half of the segment (frames 256 + r: taps j > r), which the kernel computes
as the first half's triangle on the reversed taps and the reversed period."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "cuda_audio_amd", "csrc", "jack_tail.hip.h")).read()
B = 256


def const(name):
    m = re.search(r"#define\s+%s\s+(\d+)" % name, SRC)
    assert m, name
    return int(m.group(1))


def units(nt, no):
    """td_unit: unit u -> (a, c) in the order the kernel enumerates them."""
    out = []
    for a in range(B // no):
        for c in range((no * a + no + nt - 1) // nt):
            out.append((a, c))
    return out


def run_tiling(nt, no, nthreads, h, x, pad):
    us = units(nt, no)
    assert len(us) <= nthreads, (len(us), nthreads)
    xpad = np.concatenate([np.zeros(pad), x])
    slots = {}
    count = np.zeros((B, B), int)  # (frame, tap) pairs summed
    for a, c in us:
        base = no * a - nt * c - (nt - 1)  # frame of window entry 0 (td_window)
        assert -pad <= base and base + nt + no - 2 <= B - 1
        w = xpad[pad + base:pad + base + nt + no - 1]
        hh = np.array([h[nt * c + jj] if nt * c + jj < B else 0.0 for jj in range(nt)])
        acc = np.zeros(no)
        for jj in range(nt):
            for o in range(no):
                acc[o] += hh[jj] * w[nt - 1 + o - jj]  # td_tile
                m, j = no * a + o, nt * c + jj
                if j < B and 0 <= m - j:
                    count[m, j] += 1
        for o in range(no):
            slots.setdefault(no * a + o, []).append((c, acc[o]))
    tri = np.tril(np.ones((B, B), int))
    assert np.array_equal(count, tri)
    nslots = max(len(v) for v in slots.values())
    y = np.array([sum(v for _, v in sorted(slots[m])) for m in range(B)])
    return y, nslots


def test_first_and_second_half_tilings_cover_the_triangle_once():
    rng = np.random.default_rng(5)
    h, x = rng.standard_normal(B), rng.standard_normal(B)
    full = np.convolve(x, h)  # 511 frames: the period's own term of the segment
    pad, slots, threads = const("TD_PAD"), (B + const("TD_NT1") - 1) // const("TD_NT1"), const("TAIL1_THREADS")
    y1, n1 = run_tiling(const("TD_NT1"), const("TD_NO1"), threads, h, x, pad)
    assert n1 <= slots
    assert np.abs(y1 - full[:B]).max() < 1e-12
    # second half on the helper threads (256 .. 511): reversed taps, reversed period; frame 510 - m' of the segment is output m'
    y2, n2 = run_tiling(const("TD_NT2"), const("TD_NO2"), threads - B, h[::-1].copy(), x[::-1].copy(), pad)
    assert n2 <= slots
    second = np.array([y2[B - 2 - r] if r < B - 1 else 0.0 for r in range(B)])
    assert np.abs(second - np.concatenate([full[B:], [0.0]])).max() < 1e-12


def test_unit_counts_stated_in_the_header():
    n1 = len(units(const("TD_NT1"), const("TD_NO1")))
    n2 = len(units(const("TD_NT2"), const("TD_NO2")))
    assert "%d units" % n1 in SRC and "%d units" % n2 in SRC  # the comments quote them
    assert n1 <= const("TAIL1_THREADS") and n2 <= const("TAIL1_THREADS") - B
