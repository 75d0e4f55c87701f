#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): the 12-stream
zflat/uflat round of the reference's bench (bench/src/bench.rs:83-114), tiled
to 8 GiB per GPU as independent raw streams.  One step = one compress pass
and one decompress pass of the raw block codec over the whole batch, inputs
and outputs resident in HBM.  value = uncompressed bytes through both
directions / time, whole job, GiB/s.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gib G]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N

Multi-GPU: streams are independent, so each rank compresses its own shard;
no data-path collective (scaling = weak: G GiB per GPU).
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_round():
    """The 12 bench inputs laid out back to back (16-byte aligned)."""
    import kats
    import oracle_lib  # only for the corpus file list (no oracle compute)
    rnd = oracle_lib.corpus_round()
    offs, pos = [], 0
    for _, d in rnd:
        offs.append(pos)
        pos += (max(len(d), 1) + 15) // 16 * 16
    host = np.zeros(pos, dtype=np.uint8)
    for (_, d), o in zip(rnd, offs):
        host[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    lens = [len(d) for _, d in rnd]
    shas = [kats.CORPUS_SHA256[b] for b, _ in rnd]
    return rnd, host, np.array(offs, dtype=np.int64), np.array(
        lens, dtype=np.int64), shas


def cpu_baseline(rnd, seconds=8.0):
    """Oracle (the C restatement of the reference, kind 'port') timed on the
    host cores on a bounded sample: the 12-stream round, repeated by every
    thread (pthreads inside oracle/liboracle.so) for `seconds` per
    direction."""
    import ctypes as C
    import oracle_lib as O
    L = O.lib()
    cores = os.cpu_count() or 1
    datas = [d for _, d in rnd]
    comps = [O.compress(d) for d in datas]
    n = len(datas)
    PP = C.c_char_p * n
    SZ = C.c_size_t * n
    L.snapo_bench.restype = C.c_double
    L.snapo_bench.argtypes = [PP, SZ, PP, SZ, C.c_int, C.c_int, C.c_int,
                              C.c_double, C.POINTER(C.c_uint64)]
    res = {}
    for direction, key in ((0, "c"), (1, "d")):
        rounds = C.c_uint64(0)
        t0 = time.perf_counter()
        bps = L.snapo_bench(PP(*datas), SZ(*[len(d) for d in datas]),
                            PP(*comps), SZ(*[len(c) for c in comps]), n,
                            direction, cores, seconds, C.byref(rounds))
        res[key] = (bps / GIB, rounds.value, time.perf_counter() - t0)
    c, d = res["c"][0], res["d"][0]
    combined = 2.0 / (1.0 / c + 1.0 / d)
    out = {
        "value": round(combined, 4), "unit": "GiB/s", "cores": cores,
        "kind": "port",
        "compress_gibs": round(c, 4), "decompress_gibs": round(d, 4),
        "sample": (f"12-stream zflat/uflat round (2928571 B) x "
                   f"{res['c'][1]} (compress, {res['c'][2]:.1f}s) / x "
                   f"{res['d'][1]} (decompress, {res['d'][2]:.1f}s) on "
                   f"{cores} pthreads, oracle/snappy_oracle.c -O3"),
    }
    # one thread, for comparison with the reference README (1 core)
    r1 = C.c_uint64(0)
    c1 = L.snapo_bench(PP(*datas), SZ(*[len(d) for d in datas]), PP(*comps),
                       SZ(*[len(c) for c in comps]), n, 0, 1, 2.0,
                       C.byref(r1)) / GIB
    d1 = L.snapo_bench(PP(*datas), SZ(*[len(d) for d in datas]), PP(*comps),
                       SZ(*[len(c) for c in comps]), n, 1, 1, 2.0,
                       C.byref(r1)) / GIB
    out["one_thread"] = {"compress_gibs": round(c1, 4),
                         "decompress_gibs": round(d1, 4)}
    if O.libsnappy() is not None:  # informational: Google libsnappy 1.1.8
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 2.0:
            for dd in datas:
                O.libsnappy_compress(dd)
            k += 1
        ubytes = sum(len(d) for d in datas)
        out["libsnappy_1_1_8_compress_gibs_1thread"] = round(
            k * ubytes / (time.perf_counter() - t0) / GIB, 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gib", type=float, default=8.0,
                    help="uncompressed GiB per GPU (BASELINE cfg2: 8)")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true",
                    help="experiment builds only: skip the parity gate "
                         "(the JSON line is then marked invalid)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as g
    g.build()
    import rust_snappy_amd as R
    from rust_snappy_amd import batch, raw

    rnd, host_round, r_offs, r_lens, shas = build_round()
    round_stride = int(host_round.size)
    round_ubytes = int(r_lens.sum())
    rounds = int(np.ceil(args.gib * GIB / round_ubytes))
    n = 12 * rounds
    ubytes = rounds * round_ubytes
    if rank == 0:
        log(f"[bench] {rounds} rounds x 12 streams = {n} streams, "
            f"{ubytes / GIB:.3f} GiB uncompressed per GPU, world={world}")

    # ---- inputs resident in HBM, tiled on the device --------------------
    ctx = raw.Context(local_rank)
    d_round = torch.from_numpy(host_round).to(dev)
    data = d_round.repeat(rounds)
    offs = (np.arange(rounds, dtype=np.int64)[:, None] * round_stride
            + r_offs[None, :]).reshape(-1)
    lens = np.tile(r_lens, rounds)
    src = batch.StreamBatch(data, offs, lens)
    caps = np.array([raw.max_compress_len(int(x)) for x in r_lens],
                    dtype=np.int64)
    comp = batch.StreamBatch.empty(np.tile(caps, rounds), dev)
    comp_lens = torch.zeros(n, dtype=torch.int64, device=dev)
    comp_errs = torch.zeros(32 * n, dtype=torch.uint8, device=dev)
    back = batch.StreamBatch.empty(lens, dev)
    back_lens = torch.zeros(n, dtype=torch.int64, device=dev)
    back_errs = torch.zeros(32 * n, dtype=torch.uint8, device=dev)

    def do_compress():
        raw.compress_batch(ctx, src.d_ptrs, src.d_lens, comp.d_ptrs,
                           comp.d_lens, comp_lens, comp_errs,
                           host_in_lens=src.h_lens)
        return ctx.last_timing()  # waits for this batch's last event

    def do_decompress():
        raw.decompress_batch(ctx, comp.d_ptrs, comp_lens, back.d_ptrs,
                             back.d_lens, back_lens, back_errs)
        return ctx.last_timing()

    # ---- parity gate before any number is reported ----------------------
    do_compress()
    do_decompress()
    ctx.synchronize()
    cl = comp_lens.cpu().numpy()
    if args.no_verify:
        shas, rounds_checked = [], 0
    def kinds(t):
        return np.frombuffer(t.cpu().numpy().tobytes(),
                             dtype="<i4").reshape(n, 8)[:, 0]
    assert (kinds(comp_errs) == 0).all(), "compress reported errors"
    assert args.no_verify or (kinds(back_errs) == 0).all(), \
        "decompress reported errors"
    for j in range(12 if not args.no_verify else 0):  # round 0 vs sha256
        got = comp.stream_bytes(j, cl[j])
        n_in, n_out, sha = shas[j]
        assert len(got) == n_out and hashlib.sha256(got).hexdigest() == sha, \
            f"compressed bytes of stream {j} differ from the oracle's"
    assert args.no_verify or (cl.reshape(rounds, 12) == cl[:12][None, :]).all()
    c_stride = int(comp.offsets[12]) if rounds > 1 else 0
    if rounds > 1 and not args.no_verify:  # every round equals round 0
        per = comp.data[:rounds * c_stride].view(rounds, c_stride)
        for j in range(12):
            o, m = int(comp.offsets[j]), int(cl[j])
            assert bool((per[:, o:o + m] == per[0:1, o:o + m]).all()), j
    per = back.data[:rounds * round_stride].view(rounds, round_stride)
    for j in range(12 if not args.no_verify else 0):  # round trip == input
        o, m = int(r_offs[j]), int(r_lens[j])
        assert bool((per[:, o:o + m] == d_round[None, o:o + m]).all()), \
            f"round trip of stream {j} differs from the input"
    cbytes = int(cl.sum())
    ratio = cbytes / ubytes
    if rank == 0:
        log(f"[bench] parity ok; compressed {cbytes / GIB:.3f} GiB "
            f"(ratio {ratio:.4f})")

    # ---- timed region ----------------------------------------------------
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        do_compress()
        do_decompress()
    barrier()
    k_comp_ms, k_dec_ms, t_comp, t_dec, compact_ms, plan_ms = [], [], 0.0, \
        0.0, [], []
    k_dom_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        tm = do_compress()
        tb = time.perf_counter()
        td = do_decompress()
        tc = time.perf_counter()
        k_comp_ms.append(tm["codec_ms"])
        k_dom_ms.append(tm["dominant_ms"])
        compact_ms.append(tm["compact_ms"])
        plan_ms.append(tm["plan_ms"])
        k_dec_ms.append(td["codec_ms"])
        t_comp += tb - ta
        t_dec += tc - tb
    barrier()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        log("[bench] compress kernel ms per step: "
            + " ".join(f"{x:.1f}/{y:.1f}" for x, y in zip(k_dom_ms, k_comp_ms))
            + " | decompress: " + " ".join(f"{x:.1f}" for x in k_dec_ms))
    if world > 1:
        t = torch.tensor([elapsed, t_comp, t_dec], dtype=torch.float64,
                         device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, t_comp, t_dec = t.tolist()

    if rank == 0:
        K = args.steps
        total_u = ubytes * world
        value = 2.0 * total_u * K / elapsed / GIB
        comp_gibs = total_u * K / t_comp / GIB
        dec_gibs = total_u * K / t_dec / GIB
        # roofline of the dominant kernel (k_compress_blocks): algorithmic
        # bytes per launch = U read + C written ((1+rho) B per
        # uncompressed byte), over the HIP-event duration of that launch.
        kc = float(np.mean(k_comp_ms)) * 1e-3   # all compress-side kernels
        kdom = float(np.mean(k_dom_ms)) * 1e-3  # the dominant kernel alone
        kd = float(np.mean(k_dec_ms)) * 1e-3
        alg = ubytes + cbytes
        # k_match_blocks reads the input (U) and writes 8-byte tokens, the
        # encoder kernel writes C; the algorithmic bytes of the compress
        # direction (U + C) are charged to the dominant kernel's duration
        ach = alg / kdom / 1e9
        ach_d = alg / kd / 1e9
        dom_name = ("k_match_blocks" if abs(kdom - kc) > 1e-9
                    else "k_compress_blocks")
        traffic = traffic_d = None
        pmc = ROOT / "profiles" / "r1_pmc_traffic.json"
        if pmc.exists() and abs(args.gib - 8.0) < 1e-9:
            pj = json.loads(pmc.read_text())["kernels"]
            if dom_name in pj:
                traffic = pj[dom_name]["traffic_bytes_fetch_x2"]
            if "k_decompress_streams" in pj:
                traffic_d = pj["k_decompress_streams"][
                    "traffic_bytes_fetch_x2"]
        line = {
            "metric": "GiB/s uncompressed (compress + decompress) on "
                      "zflat/uflat corpus",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": (f"raw block codec, 12-stream zflat/uflat round "
                             f"tiled x{rounds} = {ubytes / GIB:.3f} GiB per "
                             f"GPU ({n} independent raw streams), compress "
                             f"then decompress, HBM-resident"),
                "streams_per_gpu": n, "ratio": round(ratio, 4),
                "parallelism": f"shard-by-stream x{world}"},
            "compress_gibs": round(comp_gibs, 3),
            "decompress_gibs": round(dec_gibs, 3),
            "roofline": {
                "kernel": dom_name, "bound": "hbm",
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": "profiles/r1_pmc_traffic.json "
                                  "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                  "separate passes, FETCH_SIZE x2)",
                "alg_bytes_per_launch": alg,
                "avg_launch_ms": round(kdom * 1e3, 3)},
            "roofline_decompress": {
                "kernel": "k_decompress_streams", "bound": "hbm",
                "achieved": round(ach_d, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach_d / HBM_PEAK_GBS, 5),
                "traffic": traffic_d, "alg_bytes_per_launch": alg,
                "avg_launch_ms": round(kd * 1e3, 3)},
            # median and min over the timed steps (HIP events)
            "kernel_ms_median": {
                "compress": round(float(np.median(k_comp_ms)), 3),
                "decompress": round(float(np.median(k_dec_ms)), 3)},
            "kernel_ms_min": {
                "compress": round(float(np.min(k_comp_ms)), 3),
                "decompress": round(float(np.min(k_dec_ms)), 3)},
            "kernel_ms": {"plan": round(float(np.mean(plan_ms)), 3),
                          "compress": round(kc * 1e3, 3),
                          "compress_dominant": round(kdom * 1e3, 3),
                          "compact": round(float(np.mean(compact_ms)), 3),
                          "decompress": round(kd * 1e3, 3)},
        }
        if args.no_verify:
            line["INVALID"] = "experiment build, parity gate skipped"
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(rnd)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
