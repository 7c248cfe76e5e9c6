#!/usr/bin/env python3
"""Where a tile-kernel workgroup's time goes at config 4 (profiling instance 5 of the kernel):

    make -C odgi_amd/csrc ../lib/libpgsgd_x5.so
    PGSGD_DEBUG=1 PGSGD_LIB=libpgsgd_x5.so PGSGD_TILE_TAIL=1 python tools/gpu_tile_phases.py > profiles/rNN/tile_phases.jsonl

Thread 0 of every workgroup sums, in 100 MHz ticks: taking an item .. its first tile (queue, item record, predecessor's flag, window
staging, the lane table of the item's tiles, the first tile's step records), a tile's start .. its term loop (barriers, staging of
the prefetched records, stream seeds), the term loop, the loop's end .. the tile's end (snapshot pieces), the item's end (queue flush,
window write-back), and everything.  Shares of the last colour's launch of the listed iterations."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import odgi_amd as oa
# gpu_tile_phases.py [NODES [FLAGS]]: another size (300000: a launch of 586 windows on 1 280 workgroup slots — the under-occupied
# regime of mid-sized graphs and of multi-GPU ranks), extra PGSGD_FLAG_* bits (0x40000: 128-node regions)
g = oa.Graph.synthetic(int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000, 50, seed=42)
p = oa.LayoutParams.defaults(g, flags=int(sys.argv[2], 0) if len(sys.argv) > 2 else 0)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
etas = oa.path_linear_sgd_layout_schedule(p)
with oa.LayoutSession(g, p) as s:
    s.upload(X0, Y0)
    prev = [0] * 12
    for it in range(p.iter_max):
        s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
        s.sync()
        w = s.probe_words()
        if it in (6, 10, 14, 16, 20, 28):
            # words 4, 5 are cumulative over both launches of the iteration, 6..9 belong to the last launch: scale the former by a half
            item_pro, tile_pro = (w[4] - prev[4]) / 2, (w[5] - prev[5]) / 2
            loop, tile_epi, item_epi, total = w[6], w[7], w[8], w[9]
            print(json.dumps({"iteration": it, "cooling": it >= p.first_cooling_iteration(), "ticks_per_launch": {"item_start": item_pro, "tile_start": tile_pro,
                              "term_loop": loop, "tile_end": tile_epi, "item_end": item_epi, "workgroup_lifetimes": total},
                              "share": {k: round(v / max(1, total), 4) for k, v in (("item_start", item_pro), ("tile_start", tile_pro), ("term_loop", loop),
                                                                                   ("tile_end", tile_epi), ("item_end", item_epi))}}))
        prev = w
