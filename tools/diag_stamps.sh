#!/bin/sh
# builds a stamped copy of the library into tools/dbg/ and the harness next to it (developer aid)
set -e
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Inumpywren_amd/csrc"
mkdir -p tools/dbg
for f in numpywren_amd/csrc/*.hip; do $H $F -DNPW_DIAG_STAMPS -c $f -o tools/dbg/$(basename $f .hip).o 2>/dev/null; done
$H --offload-arch=gfx950 -shared -fPIC tools/dbg/*.o -o tools/dbg/libnpw_hip.so
$H $F tools/diag_stamps.hip -Ltools/dbg -lnpw_hip -Wl,-rpath,'$ORIGIN' -o tools/dbg/diag_stamps
