#!/bin/bash
# CPU-side sanitizer pass (GPU AddressSanitizer is not available on this pool): the tile emulation -- the same phase functions the
# kernels run -- and the two oracles under ASan + UBSan, driven by the host-logic tests.   bash tools/sanitize_host.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -o $T/emul.so $R/tests/host/tile_emul.cpp
gcc -O1 -g -fsanitize=address,undefined -fPIC -shared -o $T/oracle.so $R/oracle/sam2pairs_oracle.c $R/oracle/krmdup_oracle.c
cat > $T/run.py <<PY
import sys
sys.path.insert(0, "$R/tests"); sys.path.insert(0, "$R")
import util
util.EMUL_SO = "$T/emul.so"; util.ORACLE_SO = "$T/oracle.so"
import test_host_logic as t, test_krmdup as k
for name in ("edge_unc.sam", "edge_flash.sam"):
    for cfg in (0, 1, 3, 4, 5, 10, 12, 13, 14, 15): t.test_tile_phases_edge_fixtures(name, cfg)
for cfg in (0, 10, 13, 15): t.test_tile_phases_synthetic("stress", 13, 4000, ("unc", "flash"), cfg)
for cfg in (0, 10, 13): t.test_filtered_stranger_inside_a_group_never_splits_it(cfg)
t.test_ragged_and_empty_inputs(); t.test_last_line_without_newline_and_crlf(); t.test_long_fields_take_the_generic_parser()
k.test_krmdup_oracle_matches_golden()
print("sanitizer run clean")
PY
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 $T/run.py 2>&1 | grep -v "^+ " | tail -5
rm -rf $T
