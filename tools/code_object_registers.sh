#!/bin/bash
# register allocation of every kernel of libpgsgd.so from the gfx950 code object's metadata (.amdhsa):
#   tools/code_object_registers.sh [extra hipcc flags] > profiles/rNN/code_object_registers.txt
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 --cuda-device-only -S "$@" -o /tmp/pgsgd_session.s odgi_amd/csrc/pgsgd_session.hip || exit 1
python3 - <<'PY'
import re
txt = open("/tmp/pgsgd_session.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
# the metadata block lists, per kernel: .name, .sgpr_count, .sgpr_spill_count, .vgpr_count, .vgpr_spill_count, .private_segment_fixed_size
meta = txt[txt.find("amdhsa.kernels"):]
for blk in meta.split("  - .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print(g("name"), "vgpr_count", g("vgpr_count"), "sgpr_count", g("sgpr_count"), "sgpr_spill_count", g("sgpr_spill_count"),
          "vgpr_spill_count", g("vgpr_spill_count"), "private_segment_fixed_size", g("private_segment_fixed_size"))
PY
