import sys
from pathlib import Path

import pytest

import os

# The process loads the TEST build of the library (libsnapmi_test.so: the same
# sources with the knobs of include/snapmi_test.h and the cross-check kernels
# compiled in; rust-snappy_amd/_lib.py picks it when SNAPMI_TESTING is set) -
# and the SHIPPED library beside it (libsnapmi.so, _lib.load_product()): the
# "product*" parameters of the ctx / cctx fixtures below make their contexts
# with that one, so every test written against those fixtures - the
# reference's suite, the bounds, the parity files - also runs through the
# library that smoke(), bench.py and the tools load.  A test that needs a
# knob of the test build is skipped there (TestOnlyOption).
os.environ.setdefault("SNAPMI_TESTING", "1")

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by gpurun)")


@pytest.hookimpl(hookwrapper=True)
def pytest_pyfunc_call(pyfuncitem):
    """A test that asks a product-library context for a test-build knob is
    skipped (it is covered on the test build's parameters)."""
    outcome = yield
    if outcome.excinfo is not None:
        from rust_snappy_amd.raw import TestOnlyOption
        if issubclass(outcome.excinfo[0], TestOnlyOption):
            try:
                pytest.skip("needs a knob of the test build; this parameter "
                            "runs the shipped library")
            except BaseException:  # noqa: BLE001 - the Skipped exception
                outcome.force_exception(sys.exc_info()[1])


def product_context():
    """A context of the SHIPPED library (libsnapmi.so) in this process."""
    import rust_snappy_amd as R
    return R.raw.Context(0, lib=R._lib.load_product())


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session", params=["dec3", "dec2", "product"])
def ctx(request, built):
    """A context per decoder kernel: k_decompress_streams3 (element per lane,
    128-byte windows; the default) and the second-generation
    k_decompress_streams2 kept as a cross-check, so every decoder parity test
    runs through both - and "product": the shipped library with its default
    options."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import rust_snappy_amd as R
    if request.param == "product":
        c = product_context()
    else:
        c = R.raw.Context(0)
        c.set_option("decode_kernel", {"dec3": 3, "dec2": 2}[request.param])
    yield c
    c.close()


@pytest.fixture(scope="session",
                params=["spans", "spans_lds", "waves", "waves_lds", "lanes",
                        "lanes_segmented", "lanes_overlap", "both",
                        "spans_match", "small_tables", "small_tables_lanes",
                        "coresident", "spans_sched", "lanes_spill",
                        "lanes_overlap_spill", "spans_match_spill",
                        "coresident_spill", "small_tables_spill", "product",
                        "product-lanes",
                        "product-spans_lds", "product-small_tables",
                        "product-coresident", "product-lanes_spill",
                        "product-coresident_spill"])
def cctx(request, built):
    """A context per compressor kernel: the wavefront-per-block kernels (window
    steps and, as the cross-check, one copy per step; five tables per CU and
    one block per CU), the lane-per-block kernel (one launch, and split into segments of 64
    blocks) and both at once, each forced for every batch size, so every
    parity test of the encoder runs through all of them.  "lanes" and
    "lanes_segmented" encode every block at its final position
    (lane_direct_encode); "lanes_overlap" and "both" go through the scratch
    slots and k_compact.  "small_tables*": every block of at most 8 KiB
    goes to the window kernel with 16 KiB tables (k_match_spans_8k) however
    few there are, the larger blocks to the window kernel as
    match finder or to the lane kernel (both skip the other classes' blocks)
    - the configurations above switch those kernels off, so that they keep
    testing the kernels they name on blocks of every size.
    "<name>_spill": configuration <name> with a token pool of a hundredth of
    the worst case and no floor - a page or two for a batch of this suite, so
    nearly every block finds the pool empty and is compressed a second time
    by k_redo_spilled (and the first blocks of a batch are not).
    "product": the SHIPPED library (libsnapmi.so) with its default options -
    the routing a user gets; "product-<name>": the shipped library forced
    into configuration <name> (those that need no knob of the test build)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import rust_snappy_amd as R
    if request.param == "product":
        c = product_context()
        yield c
        c.close()
        return
    product = request.param.startswith("product-")
    if product:
        request = type("P", (), {"param": request.param[len("product-"):]})
        c = product_context()
    else:
        c = R.raw.Context(0)
    if request.param.endswith("_spill"):
        request = type("P", (), {"param": request.param[:-len("_spill")]})
        c.set_option("token_pool_pct", 1)
        c.set_option("token_pool_min_pages", 0)
    if request.param == "spans_sched":
        # the window kernel with the order of its blocks chosen as the launch
        # goes (SpanSched), however few blocks there are
        request = type("P", (), {"param": "spans"})
        c.set_option("span_schedule", 2)
    c.set_option("compress_mode", {"spans": 0, "spans_lds": 0, "waves": 0,
                                   "waves_lds": 0, "lanes": 1,
                                   "lanes_segmented": 1, "lanes_overlap": 1,
                                   "both": 2, "spans_match": 1,
                                   "small_tables": 1,
                                   "small_tables_lanes": 1, "coresident": 1}[
        request.param])
    # three lane wavefronts and two window wavefronts per CU on one two-ended
    # ticket (k_match_both), however few blocks there are
    c.set_option("lane_coresident", 1 if request.param == "coresident" else 0)
    c.set_option("lane_coresident_min_blocks", 1)
    small = request.param.startswith("small_tables")
    c.set_option("small_table_kernel", 1 if small else 0)
    c.set_option("small_table_min_blocks", 1)
    # the token path's match finder: the lane kernel in the "lanes*"
    # configurations whatever the last batch compressed to (the default picks
    # by that), the window kernel (k_match_spans) in "spans_match"
    c.set_option("match_kernel", 1 if request.param == "spans_match" else 0)
    # spans / waves: five tables per CU, input from L2; *_lds: one block per
    # CU, table and input block in LDS (the kernel of the smallest batches).
    # spans*: a window of 63 positions per step (k_compress_spans, the
    # default); waves*: one copy per step (k_compress_blocks, rounds 1-3)
    c.set_option("small_batch_kernel",
                 2 if request.param in ("waves_lds", "spans_lds") else 0)
    c.set_option("span_kernel",
                 0 if request.param in ("waves", "waves_lds") else 1)
    # (small_tables: blocks of more than 8 KiB by the window kernel as match
    # finder; small_tables_lanes and the others: by what lane_min_blocks 1
    # and compress_mode select)
    c.set_option("lane_min_blocks",
                 1 << 30 if request.param == "small_tables" else 1)
    # streams under 256 bytes / under 2 KiB are k_compress_tiny's /
    # k_compress_small's by default; two of the six configurations keep them
    # with the block kernels, so the block kernels' handling of small blocks
    # stays covered
    c.set_option("tiny_stream_kernel",
                 0 if request.param in ("waves", "lanes_segmented") else 1)
    if request.param == "lanes_segmented":
        c.set_option("lane_segment_blocks", 64)
        # (launches with no more blocks than lanes - every launch of this
        # suite's batches - run k_match_blocks_spec; this configuration keeps
        # the plain kernel covered)
        c.set_option("lane_speculate", 0)
    # matched in two halves, the first half encoded on the side stream
    if not product:
        c.set_test_option("lane_overlap_encode",
                          2 if request.param == "lanes_overlap" else 0)
    yield c
    c.close()


# The long-stream scan's geometry (segment size, segments per scan wavefront)
# follows the size of the call (stream_seg_log2 / stream_scan_segs in
# snapmi_api.hip): 1 KiB and 8 for everything a test can afford.  The tests of
# that path run in what large calls get, too: 4 KiB segments, full groups of
# 64, and a group size in between.
SCAN_GEOMETRIES = [pytest.param((10, 0), id="1k"),
                   pytest.param((12, 0), id="4k"),
                   pytest.param((10, 64), id="1k-64"),
                   pytest.param((12, 16), id="4k-16")]


def set_scan_geometry(ctx, geom):
    """geom = (log2 of the segment size, segments per scan wavefront; 0 = by
    size) or None (back to the library's own choice); returns the log2."""
    seg, groups = geom if geom else (0, 0)
    ctx.set_test_option("stream_seg_log2", seg)
    ctx.set_test_option("stream_scan_segs", groups)
    return seg
