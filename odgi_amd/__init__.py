"""odgi_amd — MI355X-native `odgi layout` (path-guided SGD 2D layout).

Host-side mirror of the reference interface for this one path:
  Graph                      <- graph_t + xp::XP lowered to the flat path-step index
  LayoutParams.defaults()    <- src/subcommand/layout_main.cpp:198-204,251-266
  path_linear_sgd_layout_schedule  <- src/algorithms/path_sgd_layout.cpp:433-468
  path_linear_sgd_layout_gpu       <- src/algorithms/path_sgd_layout.hpp:59-80 (the `--gpu` entry)
  LayoutSession              <- one eta step at a time (multi-GPU driver, benchmark)
  Layout                     <- src/algorithms/layout.hpp:24-40 (.lay / TSV)
  main_layout(argv)          <- the subcommand itself
  path_linear_sgd, path_linear_sgd_order  <- src/algorithms/path_sgd.hpp (the 1D sibling behind `odgi sort -Y`)
Everything computes through libpgsgd.so (HIP, gfx950); nothing here falls back to the CPU.
"""
from .graph import Graph
from .layout import (Layout, LayoutParams, LayoutSession, initial_layout, main_layout,
                     path_linear_sgd_layout_gpu, path_linear_sgd_layout_schedule, zeta_table,
                     path_stress, path_stress_near, path_distance, shard_flags)

from .sort import path_linear_sgd, path_linear_sgd_order, sort_params_defaults, sort_stress

__all__ = ["path_linear_sgd", "path_linear_sgd_order", "sort_params_defaults", "sort_stress",
           "Graph", "Layout", "LayoutParams", "LayoutSession", "initial_layout", "main_layout",
           "path_linear_sgd_layout_gpu", "path_linear_sgd_layout_schedule", "zeta_table",
           "path_stress", "path_stress_near", "path_distance", "shard_flags"]
