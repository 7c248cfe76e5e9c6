#!/usr/bin/env python3
"""bench.py — SGD node-pair updates/s of the MI355X path-guided layout kernel (BASELINE.json metric).

Workload (BASELINE.json configs[3], the one the HBM-roofline claim is quoted on; it fits one GPU):
synthetic linearised pangenome, 1,000,000 nodes / 50 haplotype paths / ~4.6e7 path steps, seed 42,
reference defaults (iter_max 30, 10*S terms per iteration, theta 0.99, cooling from iteration 15).
The first half of the schedule draws half of the partners uniformly over the path (more far
partners, slower), the cooling half draws all of them by Zipf; the default window times both.
A "step" is one SGD iteration (one learning-rate step): 10*S node-pair updates, sharded 1/G per
GPU (tile kernel: every G-th tile with its whole share of terms), followed by the coordinate-delta
all-reduce when G > 1 (strong scaling: total terms fixed).

  python bench.py --gpus 1 --steps 28 --warmup 2        # = the reference's whole 30-iteration schedule
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

Prints ONE JSON line on rank 0.  `roofline` is computed from HIP-event kernel durations measured
inside the library on the launch stream; `cpu_baseline` times the CPU oracle (a restatement of the
reference's Hogwild loop — the upstream binary cannot be built here) on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_TERM = 68.0      # SURVEY 8(d) / BASELINE.md 3: 52 B read + 16 B written per term
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=28, help="timed iterations (default: with --warmup 2 the whole 30-iteration schedule)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--paths", type=int, default=50)
    ap.add_argument("--streams", type=int, default=0, help="sampler streams per GPU (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--stress", action="store_true", help="also report sampled path stress of the final layout")
    ap.add_argument("--no-tiles", action="store_true", help="force the per-lane kernel (PGSGD_FLAG_NO_TILES)")
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0, help="extra PGSGD_FLAG_* bits (A/B runs: 0x2000 exact math, 0x4000 no partner pairs)")
    ap.add_argument("--seed", type=int, default=None, help="sampler seed (default: the reference's 9399220)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-rank exchange (prepare, all-reduce, merge) even at world size 1: under "
                         "`torchrun --nproc-per-node 1` this executes the RCCL path on a single-GPU box")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the layout kernels have no CPU fallback")
    # one rank per GPU; PGSGD_DIST_BACKEND=gloo lets several ranks share one GPU (used to exercise this
    # multi-rank path on the single-GPU test box — RCCL refuses two ranks on one device)
    backend = os.environ.get("PGSGD_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or (args.force_exchange and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")

    import odgi_amd as oa
    from odgi_amd.distributed import DistributedLayout, HipEngine

    t0 = time.time()
    g = oa.Graph.synthetic(args.nodes, args.paths, seed=42)
    iters = max(30, args.warmup + args.steps)
    from odgi_amd import _lib
    p = oa.LayoutParams.defaults(g, iter_max=iters, n_streams=args.streams, device=local_rank,
                                 flags=(_lib.FLAG_NO_TILES if args.no_tiles else 0) | args.flags)
    # several ranks: regions of 128 nodes where 256-node windows would not fill the devices — the sessions' rule then shards by
    # region with the exact exchange (G ranks hold one GPU's layout bit for bit); --flags 0x80000: by tile, faster and lossy
    p.flags |= oa.shard_flags(g, world, p.flags)
    if args.seed is not None:
        p.seed = args.seed
    X0, Y0 = oa.initial_layout(g, "d", seed=42)
    if rank == 0:
        log(f"[bench] graph N={g.n_nodes} S={g.n_steps} P={g.n_paths} terms/iter={p.min_term_updates} "
            f"built in {time.time() - t0:.1f}s")
    p.stream_offset = rank * (1 << 20)   # disjoint sampler stream ids per rank (a GPU runs < 2^20 streams)
    eng = HipEngine(g, p, X0, Y0)
    p.n_streams = eng.session.n_streams
    drv = DistributedLayout(p, eng, force_exchange=args.force_exchange)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        drv.step(it)
    if args.warmup and hasattr(eng, "flush"):
        eng.flush()   # the warm-up's last far pulls are delivered outside the window (the next launch would deliver them first otherwise)
    eng.session.kernel_time(reset=True)
    counts0 = eng.session.launch_counts()
    terms_on_device0 = eng.session.terms_executed() if hasattr(eng.session, "terms_executed") else 0
    fence()
    per_iter = []          # (iteration, update-kernel ms, launches, snapshot ms, drain ms) — host bookkeeping after each
    k_prev = (0.0, 0, 0.0, 0.0)   # step's own sync (the product's run loop syncs every iteration for delta_max too)
    t1 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        drv.step(it)
        k = eng.session.kernel_time() + eng.session.aux_time()
        per_iter.append((it, k[0] - k_prev[0], k[1] - k_prev[1], k[2] - k_prev[2], k[3] - k_prev[3]))
        k_prev = k
    # The far pulls a tile launch collects are delivered right before the NEXT launch: every timed step pays for the drain
    # of the step before it, and the timed window closes with the flush of its own last launch (and, at world > 1, the
    # exchange that merges it): the window holds exactly the drains of its own launches (the warm-up's were flushed above).
    drv.finish()
    fence()
    elapsed = time.perf_counter() - t1
    clock_mhz, clock_launch_ms = eng.session.shader_clock() if hasattr(eng.session, "shader_clock") else (0.0, 0.0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = eng.session.kernel_time()
    snapshot_ms, drain_ms = eng.session.aux_time()
    terms_on_device = eng.session.terms_executed() - terms_on_device0 if hasattr(eng.session, "terms_executed") else None
    n_kernels, n_copies = (a - b for a, b in zip(eng.session.launch_counts(), counts0))

    total_terms = float(p.min_term_updates) * args.steps
    value = total_terms / elapsed
    # one launch per exchange block: terms per launch = this rank's terms in the timed region / launches
    my_terms = drv.my_terms() * args.steps / max(launches, 1)
    avg_kernel_s = (kernel_ms / 1e3) / max(launches, 1)
    achieved = BYTES_PER_TERM * my_terms / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0

    out = {
        "metric": "SGD node-pair updates/sec",
        "value": value,
        "unit": "terms/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32 (f64 sampler, q32 coords)",
        "data": "synthetic",
        "config": {"workload": f"synthetic linearised pangenome N={g.n_nodes} S={g.n_steps} P={g.n_paths} seed 42 "
                               f"({'BASELINE configs[3]' if (g.n_nodes, g.n_paths) == (1_000_000, 50) else 'BASELINE configs[4] size' if (g.n_nodes, g.n_paths) == (10_000_000, 50) else 'not a BASELINE configuration'}); {p.min_term_updates} terms per iteration, iter_max {iters}, "
                               f"theta {p.theta}, timed iterations {args.warmup}..{args.warmup + args.steps - 1}",
                   "streams_per_gpu": int(p.n_streams),
                   # everything a step puts on the stream, and the launches of the dominant kernel among them
                   "kernel_launches_per_step": n_kernels / args.steps, "memset_and_copy_ops_per_step": n_copies / args.steps,
                   "update_kernel_launches_per_step": launches / args.steps,
                   # `value` divides the terms the schedule ASKS for by the time; what this rank's tile launches EXECUTED in the timed
                   # window, counted by the kernel's waves on the device (0 / None: per-lane kernel, whose lanes loop over their share)
                   "terms_requested_this_rank": drv.my_terms() * args.steps, "terms_executed_on_device_this_rank": terms_on_device,
                   "parallelism": f"{getattr(eng, 'shard_mode', 'terms') if drv.engine_sharded else 'terms'}-sharded x{world}, graph replicated, "
                                  + ("2 integer delta all-reduces per eta step (one per region colour; the ranks hold one GPU's coordinates bit for bit)"
                                     if drv.engine_sharded and getattr(eng, "shard_mode", "") == "regions-exact" and drv.exchanging
                                     else f"{drv.blocks} fused delta all-reduce(s) per eta step"),
                   "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0,
                   "collective_backend": dist.get_backend() if dist.is_initialized() else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "pgsgd::sgd_tile_kernel" if eng.session.tile_info()["tiled"] else "pgsgd::sgd_iteration_kernel",
                     "avg_kernel_ms": 1e3 * avg_kernel_s,
                     "terms_per_launch": my_terms, "bytes_per_term": BYTES_PER_TERM},
    }
    # The same arithmetic over everything a step costs, not only its dominant kernel: with the streaming kernels
    # around every tile launch (coordinate snapshot, far-update drain), and against the wall clock of the step.
    info = eng.session.tile_info()
    rf = out["roofline"]
    step_terms = drv.my_terms()
    all_ms = (kernel_ms + snapshot_ms + drain_ms) / args.steps
    beside_on, beside_ms = eng.session.drain_beside() if hasattr(eng.session, "drain_beside") else (False, 0.0)
    rf["aux_kernels_ms_per_step"] = {"snapshot_kernel": snapshot_ms / args.steps,
                                     # on the launch stream: far_drain_kernel (+ far_combine_kernel) in front of the next launch — or, for a session that
                                     # drains BESIDE its launches, only far_combine_kernel; that session's far_drain_kernel runs on a second stream
                                     "far_drain_kernel" if not beside_on else "far_combine_kernel": drain_ms / args.steps}
    if beside_on:
        rf["aux_kernels_ms_per_step"]["far_drain_kernel_beside_the_launches"] = beside_ms / args.steps
        rf["drain_beside_the_next_launch"] = True
    rf["frac_all_kernels"] = BYTES_PER_TERM * step_terms / (all_ms / 1e3) / 1e9 / HBM_PEAK_GBS if all_ms > 0 else 0.0
    rf["frac_wall"] = BYTES_PER_TERM * step_terms / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS
    first_cooling = p.first_cooling_iteration()
    phases = {}
    for name, sel in (("warm", lambda it: it < first_cooling), ("cooling", lambda it: it >= first_cooling)):
        rows = [r for r in per_iter if sel(r[0])]
        if rows:
            ms = sum(r[1] for r in rows) / len(rows)
            ms_all = sum(r[1] + r[3] + r[4] for r in rows) / len(rows)
            phases[name] = {"iterations": [rows[0][0], rows[-1][0]], "update_kernel_ms_per_step": ms, "all_kernels_ms_per_step": ms_all,
                            "frac": BYTES_PER_TERM * step_terms / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                            "frac_all_kernels": BYTES_PER_TERM * step_terms / (ms_all / 1e3) / 1e9 / HBM_PEAK_GBS}
    rf["phases"] = phases
    rf["whole_schedule"] = bool(args.warmup + args.steps >= iters and args.warmup <= 2)
    out["config"]["kernel_plan"] = ("per-lane kernel" if not info["tiled"] else
                                    "per-lane kernel until cooling, tile kernel after" if info["warm_per_lane"] else
                                    "tile kernel (snapshot_kernel -> sgd_tile_kernel -> far_drain_kernel per region colour)"
                                    + (": the drain on a second stream beside the other colour's launch, far_combine_kernel in front of the same colour's next" if eng.session.drain_beside()[0] else ""))
    if info["tiled"]:
        out["config"]["tile_plan"] = {"region_nodes": info["region_nodes"], "tile_steps": info["tile_steps"], "windows": info["n_work_items"],
                                      "parts_per_window": info["parts"], "work_items_launched": info["n_launch_items"],
                                      "order": "node order, one run per XCD" if info["xcd_runs"] else "one run by size"}
    # the shader clock the last tile launch of the window ran at (s_memtime against the 100 MHz s_memrealtime, read by the
    # kernel's first workgroup): a 20-step window is 0.2 s long and every figure above moves with the clock
    rf["shader_clock_mhz"] = clock_mhz
    # HBM traffic per launch: NOT measured in this run — the memory-side request counters of the L2 (TCC_EA0_RDREQ by size
    # class, TCC_EA0_WRREQ) from the committed rocprofv3 PMC passes of this same command, with the request sizes calibrated on
    # kernels of known traffic (tools/profile_cal.sh -> profiles/r06/pmc_traffic_r06.json, pmc_calibration*.json), labelled so —
    # and labelled STALE when the library's sources have changed since the passes were taken (library_source_id)
    prof = os.path.join(ROOT, "profiles", "r06", "pmc_traffic_r06.json")
    same_workload = world == 1 and args.nodes == 1_000_000 and args.paths == 50 and args.streams == 0 and not args.no_tiles and not args.flags
    if os.path.exists(prof) and same_workload:
        try:
            with open(prof) as f:
                pj = json.load(f)
            import __graft_entry__ as ge
            current = pj.get("library_source_id") == ge.built_id()
            rf["traffic"] = pj.get("hbm_bytes_per_launch")
            rf["traffic_over_algorithmic"] = pj.get("hbm_bytes_per_launch") / (BYTES_PER_TERM * my_terms) if my_terms else None
            rf["hbm_bandwidth_TB_per_s"] = pj.get("hbm_bandwidth_TB_per_s")
            rf["memory_requests_per_launch"] = pj.get("requests_per_launch")
            rf["memory_requests_per_term"] = pj.get("requests_per_launch") / my_terms if my_terms else None
            rf["memory_read_requests_per_launch"] = pj.get("read_requests_per_launch")
            rf["memory_read_request_rate_per_s"] = pj.get("read_request_rate_per_s")
            rf["random_request_ceiling_per_s"] = pj.get("random_request_ceiling_per_s")
            rf["traffic_note"] = pj.get("note")
            rf["traffic_source"] = ("profiled offline, not in this run" + ("" if current else " — STALE: taken with library sources " + str(pj.get("library_source_id")) +
                                    ", this run's are " + ge.built_id()) + ": " + str(pj.get("source")))
            rf["traffic_source_current"] = bool(current)
        except Exception as e:  # noqa: BLE001
            log(f"[bench] could not read {prof}: {e}")

    if args.stress and rank == 0:
        X, Y = eng.result()
        out["stress_sampled"] = oa.path_stress(g, X, Y, 2_000_000)
        # the near pairs' exact contribution to that figure's expectation (no sampling error: two layouts of equal quality differ by
        # up to +-25 % in the sampled figure at 1e7 nodes, profiles/r06/NOTES.md section 1)
        out["stress_near_exact"] = oa.path_stress_near(g, np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64), zmax=4)["near"]
        out["stress_initial"] = oa.path_stress(g, X0, Y0, 2_000_000)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        # CPU baseline: the oracle's Hogwild restatement of path_sgd_layout.cpp:165-377 (fp64, 1 ms
        # controller), reference Release flags minus -march=native, on a time-bounded sample of the
        # same workload (same graph, same parameters, same initial layout).
        from oracle import oracle as orc
        cpu_flags = orc.build_fast_native()   # the reference's Release flags incl. -march=native, compiled on the host that is timed
        og = orc.Graph.from_product(g)
        cores = os.cpu_count() or 1
        # the loop's shared coordinates are seq-cst atomic<double>s (path_sgd_layout.cpp:316-363): more threads than one
        # socket holds can be slower than fewer, so the baseline is the BEST of a quarter, half and all of the host's
        # hardware threads (a third of the budget each), and the line says which
        scan = []
        for t in sorted({max(1, cores // 4), max(1, cores // 2), cores}):
            _, _, st_t = orc.layout_hogwild(og, orc.params_from(p), t, X0, Y0, max_seconds=max(1.0, args.cpu_seconds / 3), fast=True)
            scan.append((st_t["terms"] / st_t["seconds"] if st_t["seconds"] > 0 else 0.0, t, st_t))
        best_rate, best_threads, st = max(scan)
        # the same loop on ONE thread (`-t 1`), a quarter of the budget
        _, _, st1 = orc.layout_hogwild(og, orc.params_from(p), 1, X0, Y0, max_seconds=max(1.0, args.cpu_seconds / 4), fast=True)
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
        except Exception:  # noqa: BLE001
            pass
        out["cpu_baseline"] = {
            "value": best_rate,
            "unit": "terms/s", "cores": best_threads, "kind": "port",
            "sample": f"{st['terms']} terms in {st['seconds']:.1f} s of the same workload "
                      f"({best_threads} Hogwild threads — the best of {[t for _, t, _ in scan]} on a host of {cores} hardware threads — fp64, "
                      f"CPU restatement of the reference; upstream is unbuildable here)",
            "thread_scan": [{"threads": t, "value": r} for r, t, _ in scan], "host_threads": cores,
            "single_thread": {"value": st1["terms"] / st1["seconds"] if st1["seconds"] > 0 else 0.0, "cores": 1,
                              "sample": f"{st1['terms']} terms in {st1['seconds']:.1f} s, same loop, one thread (-t 1)"},
            "cpu_model": cpu_model, "build_flags": cpu_flags,
        }
    eng.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
