#!/bin/bash
# Offline view of the run-time compiled pre-pass: config + ISA + resource counts of piece_spec.hip for one aligner.
#   tools/jit/spec_offline.sh [adapter] [e] [flags] [min_overlap] [max_len] [ragged] [extra -D flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
AD=${1:-AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC}; E=${2:-0.1}; FL=${3:-14}; MO=${4:-3}; LEN=${5:-150}; RG=${6:-0}
shift 6 2>/dev/null || true
OUT=${SPEC_OUT:-/tmp/w/spec}
mkdir -p "$OUT" "$ROOT/build/csrc"
CS=$ROOT/atropos_amd/csrc
make -s -C "$CS" "$ROOT/build/csrc/jit_sources.inc" >/dev/null 2>&1 || (cd "$CS" && python3 embed_sources.py ../../build/csrc/jit_sources.inc piece_spec.hip piece_filter.hpp fast_work.hpp piece_core.hpp filter_core.hpp locate_core.hpp ../../include/atropos_hip.h)
if [ ! -x "$OUT/spec_offline" ] || [ "$ROOT/tools/jit/spec_offline.cpp" -nt "$OUT/spec_offline" ] || [ "$CS/jit.hpp" -nt "$OUT/spec_offline" ] || [ "$CS/piece_core.hpp" -nt "$OUT/spec_offline" ]; then
  /opt/rocm/bin/hipcc -O1 -std=c++17 --offload-arch=gfx950 -x hip -I"$CS" -I"$ROOT/include" -I"$ROOT/build/csrc" "$ROOT/tools/jit/spec_offline.cpp" -o "$OUT/spec_offline" -ldl 2>&1 | grep -v warning || true
fi
"$OUT/spec_offline" "$AD" "$E" "$FL" "$MO" "$LEN" "$RG" "$OUT" $SPEC_RTC
NW=$(( (LEN + 31) / 32 ))
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -include cstring -include algorithm -I"$OUT" -I"$CS" -I"$ROOT/include" -DATR_SPEC=1 -DATR_SPEC_NW=$NW -DATR_SPEC_RAGGED=$RG "$@" "$CS/piece_spec.hip" -o "$OUT/spec.s" 2>&1 | grep -v "hip-link" || true
python3 "$ROOT/tools/kstats.py" "$OUT/spec.s"
