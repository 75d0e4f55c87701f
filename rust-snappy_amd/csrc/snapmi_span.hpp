// snapmi: the walk of k_compress_spans (snapmi_compress.hip) - the uniform
// (scalar) half of a wavefront's step over a WINDOW of 63 consecutive
// positions of its block.
//
// The wavefront-per-block kernel of rounds 1-3 (k_compress_blocks) evaluates
// 64 probes of the reference's schedule per step and stops at the first hit:
// one LDS atomic + one candidate gather + one ballot PER EMITTED COPY, ~2 100
// cycles each, ~7 000 copies per 64 KiB of text.  But everything a step needs
// from memory - "what did the table hold for the hash at position P" and "how
// many bytes match there" - can be fetched for ALL positions of a window at
// once, whether the parse will look them up or not; what is sequential in the
// reference (src/compress.rs:195-317: which positions are looked up, which
// are inserted) is then a walk over per-lane results that are already in
// registers: a handful of scalar instructions per event instead of a memory
// round trip.  Lane L of a step holds position base - 1 + L:
//
//   old(L)  what ONE lane-ordered LDS exchange returned for the slot of that
//           position's hash: the table's entry, or - if a lower lane of this
//           very step has the same hash - that lane's position ("C bit");
//   m(L)    common prefix (0..16) of the 16 bytes at the position and the 16
//           bytes at old(L); hit = m >= 4.
//
// The exchange has written EVERY position of the window into the table; the
// walk decides which of them the reference really inserts (`touched`), and one
// lane-ordered store afterwards puts every slot right: touched lanes write
// their position, untouched lanes without a C bit give back what they
// displaced.  That is exact as long as a lane with a C bit is only touched
// when the lower lane it collided with was touched too (then that lane's
// position IS the reference's candidate, and the bytes compared were the
// right ones); otherwise the walk stops in front of it and the next step
// starts there with a fresh exchange ("cut": 1-2 % of the steps on text).
//
// Runs of misses are walked with one find-first-set over (hits | C bits).  A
// run of more than 32 misses leaves the window regime (the reference's stride
// grows, src/compress.rs:207-211): the caller continues it with the
// schedule-ordered step of k_compress_blocks.  A match of 16 bytes or more is
// finished by the caller (extend_match), then the next step starts behind it.
//
// The same text runs on the host: tests/span_wave_host.cpp emulates the 64
// lanes (exchange and store in ascending lane order) around this walk and
// tests/test_span_wave_cpu.py checks its bytes against the reference's.
#ifndef SNAPMI_SPAN_HPP
#define SNAPMI_SPAN_HPP

#include <stdint.h>

#include "snapmi_tiny.hpp" // SNAPMI_LANE_FN

namespace snapmi {

// probes 0 .. kSpanRun of a run lie at consecutive positions (skip reaches 64
// after 32 probes, src/compress.rs:207-211); the window regime handles probes
// 0 .. kSpanRun - 1, whose limit check is "position + 1 <= s_limit"
constexpr uint32_t kSpanRun = 32;

struct SpanState {
    uint32_t s;         // position of the next lookup
    uint32_t q;         // chain == 0: probes of the current run already done
    uint32_t chain;     // 1: a copy ended at s; s - 1 is not inserted yet
    uint32_t next_emit; // first byte not yet covered by a token
};

enum : uint32_t {
    kSpanCont = 0, // state says where the next step starts
    kSpanLong = 1, // lane `at` hit with >= 16 equal bytes: the caller extends
                   // the match, emits the token and restarts behind it
    kSpanDone = 2, // the block's parse is over (src/compress.rs:212-214,275-277)
};

// LN: uniform accessors of per-lane results, LN::m(lane), LN::old(lane).
// SINK: SINK::token(literal_len, copy_len, offset).
// hits / cbits: bit L = lane L hit / has a C bit (only lanes that took part in
// the exchange).  touched: bit L set = the reference inserted lane L's
// position (bit 0 = the insert of s - 1 when st.chain was set).
template <class LN, class SINK>
SNAPMI_LANE_FN uint32_t span_walk(SpanState &st, const uint64_t hits,
                                  const uint64_t cbits, const uint32_t s_limit,
                                  const LN &ln, SINK &sink, uint64_t &touched,
                                  uint32_t &at)
{
    const uint32_t base = st.s; // lane L = position base - 1 + L
    uint64_t T = st.chain ? 1ull : 0ull;
    uint32_t L = 1, q = st.q;
    bool chain = st.chain != 0;
    const uint64_t stop = hits | cbits;
    uint32_t rc = kSpanCont;
    for (;;) {
        uint32_t P = base - 1 + L;
        if (!chain) {
            // a run of probes, one position apart: lanes L, L+1, .. are probes
            // q, q+1, ..; plain misses (no hit, no C bit) up to the first
            // lane that is neither, the window's end, probe kSpanRun or the
            // block's limit (the probe at P needs P + 1 <= s_limit)
            uint32_t room = 64 - L;
            if (kSpanRun - q < room)
                room = kSpanRun - q;
            const uint32_t lim = s_limit > P ? s_limit - P : 0;
            if (lim < room)
                room = lim;
            const uint64_t ahead = L < 64 ? stop >> L : 0;
            uint32_t k = ahead ? (uint32_t)__builtin_ctzll(ahead) : 64;
            if (k > room)
                k = room;
            if (k) {
                T |= (k == 64 ? ~0ull : ((1ull << k) - 1)) << L;
                q += k;
                L += k;
                P += k;
            }
            if (k == room) { // a limit, whatever lane L holds
                st.s = P;
                st.q = q;
                st.chain = 0;
                rc = P + 1 > s_limit ? kSpanDone : kSpanCont;
                break;
            }
        }
        // lane L (< 64): a chain check, or a probe that hits or has a C bit
        const uint64_t bit = 1ull << L;
        if (cbits & bit) {
            const uint32_t pred = ln.old(L) - (base - 1);
            if (!((T >> pred) & 1)) { // cut: the next step starts here
                st.s = P;
                st.q = q;
                st.chain = chain ? 1 : 0;
                break;
            }
        }
        T |= bit;
        if (hits & bit) {
            const uint32_t m = ln.m(L);
            if (m >= 16) {
                at = L;
                st.s = P;
                rc = kSpanLong;
                break;
            }
            sink.token(P - st.next_emit, m, P - ln.old(L));
            const uint32_t e = P + m;
            st.next_emit = e;
            st.s = e;
            st.q = 0;
            st.chain = 1;
            if (e >= s_limit) {
                rc = kSpanDone;
                break;
            }
            // src/compress.rs:290-297: insert e - 1, then the check at e
            const uint32_t Li = L + m - 1;
            if (Li >= 63)
                break; // both belong to the next step
            const uint64_t ibit = 1ull << Li;
            if (cbits & ibit) {
                const uint32_t pred = ln.old(Li) - (base - 1);
                if (!((T >> pred) & 1))
                    break;
            }
            T |= ibit;
            L = Li + 1;
            chain = true;
        } else {
            // a miss: behind a chain check the run starts anew
            // (src/compress.rs:310-312)
            q = chain ? 0 : q + 1;
            chain = false;
            L++;
        }
    }
    touched = T;
    return rc;
}

} // namespace snapmi
#endif
