#!/bin/bash
# usage: isa_report.sh file.hip  -> instruction-kind census and per-kernel register use (gfx950)
set -e
HERE=$(dirname "$(readlink -f "$0")")
F=$1; OUT=/tmp/$(basename "$F" .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -S --cuda-device-only "$HERE/$F" -o "$OUT" 2>&1 | grep -v "warning\|^$" || true
grep -o "global_atomic[a-z_0-9]*\|flat_atomic[a-z_0-9]*\|flat_load[a-z_0-9]*\|flat_store[a-z_0-9]*\|scratch_[a-z_0-9]*\|v_mfma[a-z_0-9]*" "$OUT" | sort | uniq -c
grep -E "^\s+\.(vgpr_count|sgpr_count|name:|private_segment_fixed_size)" "$OUT" | paste - - - - | awk '{print $2, "scratch="$4, "sgpr="$6, "vgpr="$8}'
