#!/usr/bin/env python3
"""profiles/<round>/pmc_calibration_<tag>.json and pmc_traffic_<tag>.json from a tools/profile_cal.sh run (gpurun_out/prof_cal_<tag>/summary.json):
    python tools/summarize_cal.py gpurun_out/prof_cal_<tag>/summary.json profiles/r05 r05
(`library_source_id` = odgi_amd/lib/BUILD_ID of the profiled library: bench.py says "stale" when the sources have changed since.)

Calibration: for each kernel of known traffic (tools/microbench.hip, MICROBENCH_CAL=1) the memory-side read requests of the
L2 by size class (TCC_EA0_RDREQ_32B / _64B / _128B) and its write requests (TCC_EA0_WRREQ, _64B) against the bytes and
requests the kernel is known to make.  Traffic of the tile kernel: the same counters of the bench command's launches,
weighted as the driver's window weighs them (20 warm + 20 cooling launches of iterations 5..24)."""
import json, sys, os
src = sys.argv[1]
out_dir = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05"
tag = sys.argv[3] if len(sys.argv) > 3 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    build_id = open(os.path.join(ROOT, "odgi_amd", "lib", "BUILD_ID")).read().strip()   # hash of the sources the profiled library was built from
except OSError:
    build_id = "unknown"
d = json.load(open(src))
C, K = d["counters"], d["known"]
def cnt(k, c):
    return C.get(k, {}).get(c, {}).get("mean", 0.0)
def traffic(k):
    r32, r64, r128 = cnt(k, "TCC_EA0_RDREQ_32B_sum"), cnt(k, "TCC_EA0_RDREQ_64B_sum"), cnt(k, "TCC_EA0_RDREQ_128B_sum")
    rd = cnt(k, "TCC_EA0_RDREQ_sum")
    if r64 + r128 + r32 == 0:           # size classes not collected: every request counted as 64 bytes (a lower bound)
        r64 = rd
    w, w64 = cnt(k, "TCC_EA0_WRREQ_sum"), cnt(k, "TCC_EA0_WRREQ_64B_sum")
    return {"rdreq": rd, "rdreq_32B": r32, "rdreq_64B": r64, "rdreq_128B": r128, "wrreq": w, "wrreq_64B": w64,
            "bytes_read": 32 * r32 + 64 * r64 + 128 * r128, "bytes_written": 64 * w64 + 32 * (w - w64),
            "fetch_size_bytes": 1024 * cnt(k, "FETCH_SIZE"), "write_size_bytes": 1024 * cnt(k, "WRITE_SIZE"),
            "l2_hit": cnt(k, "TCC_HIT_sum"), "l2_miss": cnt(k, "TCC_MISS_sum"),
            "rdreq_dram": cnt(k, "TCC_EA0_RDREQ_DRAM_sum"), "wrreq_dram": cnt(k, "TCC_EA0_WRREQ_DRAM_sum"),
            "mean_duration_ms": next((v["mean_duration_ns"] for v in C.get(k, {}).values()), 0.0) / 1e6}
cal = {}
for name, kn in K.items():
    key = "mb:" + name
    if key not in C:
        continue
    t = traffic(key)
    disp = next(iter(C[key].values()))["dispatches"]
    # cal_stream_read is launched twice (1 GiB warm-up + 2 GiB): the counters are the mean of the two
    known_read = kn["bytes_read"] * (0.75 if name == "cal_stream_read" and disp == 2 else 1.0)
    cal[name] = {"known": kn, "dispatches": disp, "counters": t,
                 "counted_read_over_known": t["bytes_read"] / known_read if known_read else None,
                 "counted_written_over_known": t["bytes_written"] / kn["bytes_written"] if kn["bytes_written"] else None,
                 "fetch_size_over_known": t["fetch_size_bytes"] / known_read if known_read else None,
                 "read_requests_per_known_request": t["rdreq"] / (kn["requests"] * (0.75 if name == "cal_stream_read" and disp == 2 else 1.0)) if kn["bytes_read"] else None}
ceil = max((K[n]["G_requests_per_s"] for n in ("cal_gather16", "cal_gather32") if n in K), default=0.0) * 1e9
json.dump({"source": src, "library_source_id": build_id, "random_64B_request_ceiling_per_s": ceil, "kernels": cal}, open(os.path.join(out_dir, f"pmc_calibration_{tag}.json"), "w"), indent=1)
warm, cool = traffic("bench:sgd_tile_kernel_warm"), traffic("bench:sgd_tile_kernel_cooling")
mix = {k: 0.5 * (warm[k] + cool[k]) for k in warm}
drain = traffic("bench:far_drain_kernel")
res = {"library_source_id": build_id,
       "source": "rocprofv3 --pmc passes (TCC_EA0_RDREQ by size class, TCC_EA0_WRREQ, TCC_HIT/MISS; one group per pass) of `bench.py --cpu-seconds 0 --steps 20 "
                 "--warmup 5`, tools/profile_cal.sh + tools/summarize_cal.py; " + src,
       "note": "Calibrated on kernels of known traffic (pmc_calibration.json): the L2's memory-side request COUNTS are exact (a random 16- or 32-byte "
               "gather = 1.00 read request, a 64-byte line write = 1.00 write request of 64 bytes, a streaming read = one request per 128 bytes), and a "
               "read request moves a whole 128-BYTE LINE, as its size class says: with few lanes in flight the other half of a gathered line is an L2 "
               "hit (+385 ns against +640 ns for another random record, microbench_wave_local.jsonl: line fill), lanes of one instruction that cover "
               "a line cost one request, and random gathers top out at 51-55 G requests/s = 6.5-7.0 TB/s of lines — the HBM ceiling, not a request "
               "ceiling.  (Earlier in round 4 the second half's misses under 524 288 lanes were read as 64-byte sector fills; they were capacity "
               "misses.  FETCH_SIZE counts 64 bytes per read request: half, the guide's gfx950 correction.)  hbm_bytes_per_launch = 128 B x 128-byte "
               "read requests + 64 B x 64-byte ones + 32 B x 32-byte ones + the same for writes; hbm_bytes_if_64B_sectors is the earlier reading, "
               "kept for comparison.  Mean of the warm and the cooling instance, as the driver's window (iterations 5..24) weighs them.",
       "hbm_bytes_per_launch": mix["bytes_read"] + mix["bytes_written"], "hbm_bytes_if_64B_sectors": 64.0 * mix["rdreq"] + mix["bytes_written"],
       "hbm_bandwidth_TB_per_s": {"warm": (warm["bytes_read"] + warm["bytes_written"]) / warm["mean_duration_ms"] / 1e9 if warm["mean_duration_ms"] else None,
                                  "cooling": (cool["bytes_read"] + cool["bytes_written"]) / cool["mean_duration_ms"] / 1e9 if cool["mean_duration_ms"] else None},
       "requests_per_launch": mix["rdreq"] + mix["wrreq"], "read_requests_per_launch": mix["rdreq"], "write_requests_per_launch": mix["wrreq"],
       "read_request_rate_per_s": {"warm": warm["rdreq"] / warm["mean_duration_ms"] * 1e3 if warm["mean_duration_ms"] else None,
                                   "cooling": cool["rdreq"] / cool["mean_duration_ms"] * 1e3 if cool["mean_duration_ms"] else None},
       "random_request_ceiling_per_s": ceil, "tile_kernel_warm": warm, "tile_kernel_cooling": cool, "tile_kernel_window_mean": mix, "far_drain_kernel": drain}
json.dump(res, open(os.path.join(out_dir, f"pmc_traffic_{tag}.json"), "w"), indent=1)
for k in ("tile_kernel_warm", "tile_kernel_cooling", "far_drain_kernel"):
    t = res[k]
    ms = t["mean_duration_ms"]
    print("%-20s %.3f ms  read %.2f GB (64B %.3g, 128B %.3g)  written %.2f GB  requests %.4g = %.1f G/s  -> %.2f TB/s  L2 hit %.3f" %
          (k, ms, t["bytes_read"] / 1e9, t["rdreq_64B"], t["rdreq_128B"], t["bytes_written"] / 1e9, t["rdreq"] + t["wrreq"],
           (t["rdreq"] + t["wrreq"]) / ms / 1e6 if ms else 0, (t["bytes_read"] + t["bytes_written"]) / ms / 1e9 if ms else 0,
           t["l2_hit"] / max(1.0, t["l2_hit"] + t["l2_miss"])))
print("random 64-byte request ceiling %.1f G/s" % (ceil / 1e9))
for n, c in cal.items():
    print("%-24s counted/known read %s written %s  requests/known %s" % (n, c["counted_read_over_known"], c["counted_written_over_known"], c["read_requests_per_known_request"]))
