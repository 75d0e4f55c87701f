#!/usr/bin/env python3
"""After `tests/hw/final_profile.sh r6_final` on the GPU box (gpurun merges
gpurun_out/ back): copy the set into profiles/r6_final_*, rewrite
r6_pmc_traffic.json from the line's own roofline.traffic, and print the
figures the documents quote.

  python profiles/take_final.py [tag] [gpu suite log]
"""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_final"
G, P = ROOT / "gpurun_out", ROOT / "profiles"
pairs = {f"bench_{tag}.json": f"{tag}_bench.json",
         f"kernel_stats_{tag}.md": f"{tag}_kernel_stats.md",
         f"prof_{tag}_bench.json": f"{tag}_bench_under_rocprof.json",
         f"scalar_latency_{tag}.txt": f"{tag}_scalar_latency.txt",
         f"stream_{tag}_kernel_stats.md": f"{tag}_stream_kernel_stats.md"}
for a, b in pairs.items():
    shutil.copy(G / a, P / b)
if len(sys.argv) > 2:
    shutil.copy(sys.argv[2], P / f"{tag}_gpu_suite.txt")
line = [x for x in (P / f"{tag}_bench.json").read_text().splitlines()
        if x.startswith("{")][-1]
(P / f"{tag}_bench.json").write_text(line + "\n")
d = json.loads(line)
json.dump({
    "workload": "bench.py default (cfg2, 8.002 GiB), round 6",
    "method": "measured by bench.py itself (two child runs of one step under "
              "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, "
              "separate passes; FETCH_SIZE x2 + WRITE_SIZE per launch): the "
              f"roofline.traffic fields of profiles/{tag}_bench.json",
    "kernels": {
        d["roofline"]["kernel"]: {
            "traffic_bytes_fetch_x2": d["roofline"]["traffic"]},
        d["roofline_decompress"]["kernel"]: {
            "traffic_bytes_fetch_x2": d["roofline_decompress"]["traffic"]}}},
    open(P / "r6_pmc_traffic.json", "w"), indent=1)
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
ex = d["extras"]
print("source_sha16", d["source_sha16"], "tree", bench.source_sha16(),
      "git", d["git_sha"])
print("value", d["value"], "compress", d["compress_gibs"], "decompress",
      d["decompress_gibs"], "ms/step", d["ms_per_step"], "first call",
      d["first_compress_call_ms"])
print("kernel_ms", d["kernel_ms"])
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"],
      d["roofline"]["whole_side"]["frac"], "| dec",
      d["roofline_decompress"]["frac"], d["roofline_decompress"]["traffic"])
print("placement", d["placement"])
c = d["cpu_baseline"]
print("cpu value", c["value"], "cores", c["cores"], "fast", c["compress_gibs"],
      c["decompress_gibs"], "libsnappy", c["libsnappy_1_1_8"]["all_cores"],
      "oracle", c["oracle_plain_loops"]["all_cores"], "1 thread",
      c["port_fast"]["one_thread"], c["libsnappy_1_1_8"]["one_thread"])
r = [v["ratio"] for v in c["per_file_mbs_compress_decompress"].values()]
print("port / libsnappy per file: compress", min(x[0] for x in r),
      max(x[0] for x in r), "decompress", min(x[1] for x in r),
      max(x[1] for x in r))
for k, v in ex["sweep"]["sizes"].items():
    print("sweep", k, v["compress_gibs"], v["decompress_gibs"],
          v["compress_ms"], v["decompress_ms"], v["first_call_ms"])
print("sweep wall", ex["sweep"]["wall_s"])
for k, v in ex["budget"]["budgets"].items():
    print("budget", k, v["compress_gibs"], v["context_bytes"],
          v.get("first_call_ms"), v.get("placement", "")[:50])
c5 = ex["cfg5"]
print("cfg5", c5["compress_ms"], c5["decompress_ms"], c5["compress_gibs"],
      c5["decompress_gibs"], c5["compress_hbm_frac"],
      c5["decompress_hbm_frac"],
      {k: (v["compress_ms"], v["decompress_ms"]) for k, v in c5.items()
       if isinstance(v, dict)})
print("cfg3", ex["cfg3"]["frame_encode_gibs"], ex["cfg3"]["frame_decode_gibs"],
      ex["cfg3"]["frame_decode_no_index_gibs"])
print("files", {k[6:]: (v["compress_gibs"], v["decompress_gibs"])
                for k, v in ex["files"]["files"].items()})
s = ex["seam"]["snapmi"]
print("seam jpg200", s["zflat03_jpg_200"], "txt1", s["zflat06_txt1"], "html",
      s["zflat00_html"], "| libsnappy txt1",
      ex["seam"]["libsnappy_1_1_8"]["zflat06_txt1"])
t = ex["tiny"]
print("tiny", t["compress_gibs"], t["decompress_gibs"],
      {k: (v["compress_gibs"], v["decompress_gibs"]) for k, v in t.items()
       if isinstance(v, dict)})
print("pcie", ex["pcie"]["frame_encode_gibs"], ex["pcie"]["frame_decode_gibs"])
a = ex["adapters"]
print("adapters", a["frame_encoder_write_all_gibs"],
      a["frame_decoder_readinto_pinned_from_pinned_gibs"],
      a["frame_decoder_read_to_end_gibs"])
print("stream", ex["stream"]["decompress_stream_gibs"],
      ex["stream"]["batch_of_one_gibs"])
print("cfg4", ex["cfg4"]["frame_encode_gibs_no_gather"],
      ex["cfg4"]["encode_ms"])
rp = json.loads([x for x in (P / f"{tag}_bench_under_rocprof.json").read_text()
                 .splitlines() if x.startswith("{")][-1])
print("under rocprof kernel_ms", rp["kernel_ms"])
print((P / f"{tag}_kernel_stats.md").read_text().splitlines()[2][:120])
print((P / f"{tag}_kernel_stats.md").read_text().splitlines()[3][:120])
