// snapmi_profile.hpp -- the instrumentation of the experiment build
// (`make -C rust-snappy_amd/csrc profile`: -DSNAPMI_PROFILE=1 or =2 into
// libsnapmi_profile*.so, read by tests/hw/prof_*.py).  This header is the ONE
// place where that build differs: the kernels say PROF(...), TICK(i) and
// COUNT(x), which are nothing in the product and test builds.
//
//   PROF(code)   code that exists in the experiment build only (counters'
//                declarations, their flush into CompressArgs::prof)
//   TICK(i)      add the cycles since the last TICK to phase counter pt[i]
//                (SNAPMI_PROFILE=1: after waiting for outstanding memory
//                operations, so a phase owns its waits; =2: without, so a
//                wait is counted where the product build has it - in the
//                phase that first needs the data)
//   COUNT(x)     x++
#pragma once

#ifdef SNAPMI_PROFILE
#define PROF(...) __VA_ARGS__
#if SNAPMI_PROFILE == 2
#define TICK_WAIT()
#else
#define TICK_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define TICK(i)                                                               \
    do {                                                                      \
        TICK_WAIT();                                                          \
        const uint64_t _t = __builtin_readcyclecounter();                     \
        pt[i] += _t - t_last;                                                 \
        t_last = _t;                                                          \
    } while (0)
#define COUNT(x) (x)++
#else
#define PROF(...)
#define TICK(i)                                                               \
    do {                                                                      \
    } while (0)
#define COUNT(x)                                                              \
    do {                                                                      \
    } while (0)
#endif
