"""The host's arithmetic of the token path (csrc/snapmi_pool.hpp), on the CPU:
how a batch's blocks are cut into launches, how many pages a launch's token
pool gets, how the pool grows behind a batch that spilled.  The kernels that
live by these numbers are covered by the GPU suite (the `*_spill`
configurations, test_token_pool_spills_are_compressed_again_and_the_pool_grows);
here the rules themselves: what the header and DESIGN 4.1 promise."""
import ctypes as C
import random
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def P(tmp_path_factory):
    so = tmp_path_factory.mktemp("pool") / "pool_host.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC",
                           str(ROOT / "tests" / "pool_host.cpp"), "-o",
                           str(so)])
    L = C.CDLL(str(so))
    u64, u32 = C.c_uint64, C.c_uint32
    L.t_segment_blocks.restype = u64
    L.t_segment_blocks.argtypes = [u64, u64]
    L.t_pool_worst_pages.restype = u64
    L.t_pool_worst_pages.argtypes = [u64, u64, u64]
    L.t_pool_pages.restype = u64
    L.t_pool_pages.argtypes = [u64, u64, u64, u32, u32, u64]
    L.t_pool_grow.restype = u32
    L.t_pool_grow.argtypes = [u32, u64, u64]
    return L


def test_launches_of_a_batch_are_equal_and_within_the_limit(P):
    rng = random.Random(6)
    cases = [(146700, 262144), (146700, 98304), (146700, 65536),
             (262145, 262144), (1, 64), (64, 64), (65, 64), (1 << 20, 262144)]
    cases += [(rng.randrange(1, 1 << 22), rng.choice([64, 1000, 98304,
                                                      262144]))
              for _ in range(2000)]
    for blocks, mx in cases:
        seg = P.t_segment_blocks(blocks, mx)
        assert 1 <= seg <= mx or blocks <= mx, (blocks, mx, seg)
        launches = -(-blocks // seg)
        assert launches == -(-blocks // mx), (blocks, mx, seg)  # no more of them
        # equal but for rounding: the last launch is short by less than one
        # block per launch
        assert launches * seg - blocks < launches, (blocks, mx, seg)
    assert P.t_segment_blocks(146700, 98304) == 73350
    assert P.t_segment_blocks(100, 262144) == 100


def test_pool_of_a_hundred_per_cent_holds_every_block(P):
    """100 means "no block can spill": 33 token pages and 4 exception pages
    for every block of the launch and what the launch keeps in hand,
    whatever the batch's bytes and however they are spread over launches."""
    for blocks, seg, lanes, nbytes in [(146700, 146700, 65536, 8592427314),
                                      (146700, 73350, 65536, 8592427314),
                                      (1000, 1000, 1024, 1000 * 4096),
                                      (10, 10, 64, 0)]:
        pages = P.t_pool_pages(nbytes, blocks, seg, lanes, 100, 0)
        assert pages == seg * 37 + lanes + lanes // 2


def test_default_pool_is_under_half_the_input_and_over_what_cfg2_asks(P):
    """cfg2: 143 717 blocks of 8.59e9 bytes, 65 536 lanes in flight.  At the
    default 39 per cent the pool, its page tables and the staging arrays of
    k_match_both's window wavefronts are under half the input - and over the
    1.927 M pages the launch asks for (profiles/r6_token_pool.txt)."""
    nbytes, blocks, lanes = 8589640742, 143717, 65536
    pages = P.t_pool_pages(nbytes, blocks, blocks, lanes, 39, 32768)
    scratch = ((pages + 1) * 2048 + blocks * (160 + 4 + 4)
               + 256 * 2 * 18496 * 4)
    assert scratch < 0.5 * nbytes, scratch / nbytes
    assert pages > 1_927_000 * 1.04, pages
    # a caller that does not say how many bytes: full blocks are assumed
    assert P.t_pool_pages(0, blocks, blocks, lanes, 39, 32768) >= pages


def test_small_batches_never_spill(P):
    """The floor of 32 768 pages: a batch of up to 750 blocks gets the bound
    of every block whatever the percentage."""
    for blocks in (1, 17, 100, 750):
        lanes = (blocks + 63) // 64 * 64
        pages = P.t_pool_pages(blocks * 65536, blocks, blocks, lanes, 1, 32768)
        worst = P.t_pool_worst_pages(blocks * 65536, blocks, blocks)
        assert worst >= blocks * 37
        # (the launch's own worst case and what it has in hand: no more)
        assert pages == worst + lanes + lanes // 2
    # ... and the batch behind the floor does
    blocks, lanes = 2000, 2048
    assert P.t_pool_pages(blocks * 65536, blocks, blocks, lanes, 1,
                          32768) == 32768 < blocks * 37


def test_worst_case_covers_what_a_block_can_fill(P):
    """n bytes hold at most n / 4 + 1 tokens and n / 65 exceptions: the pages
    of the formula cover the pages those fill, for every block length."""
    for n in list(range(1, 2000)) + [4096, 8192, 8193, 65535, 65536]:
        worst = P.t_pool_worst_pages(n, 1, 1)
        tokens, exc = n // 4 + 1, n // 65
        assert worst >= -(-tokens // 512) + -(-exc // 256), n


def test_pool_grows_behind_a_batch_that_spilled_and_not_otherwise(P):
    assert P.t_pool_grow(39, 0, 1000) == 39
    assert P.t_pool_grow(39, 10, 1000) == 39          # a hundredth: as it is
    assert P.t_pool_grow(39, 11, 1000) == 46          # under a tenth: a sixth
    assert P.t_pool_grow(39, 100, 1000) == 46
    assert P.t_pool_grow(39, 101, 1000) == 59         # beyond: by half
    assert P.t_pool_grow(59, 500, 1000) == 89
    assert P.t_pool_grow(89, 500, 1000) == 100        # never over 100
    assert P.t_pool_grow(100, 500, 1000) == 100
    assert P.t_pool_grow(39, 5, 0) == 39              # nothing was launched
    # from any start, a batch that keeps spilling reaches 100 in a few steps
    now, steps = 1, 0
    while now < 100:
        now = P.t_pool_grow(now, 900, 1000)
        steps += 1
    assert steps <= 12
