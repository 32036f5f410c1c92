#!/bin/sh
# Developer aid: builds a copy of the library with wall-clock stamps inside qr_panel3_kernel (-DNPW_QR_STAMPS) into
# tools/dbg/qs/ and the harness tools/dbg/qr_stamps that prints where a column's time goes (run it on the GPU box).
set -e
cd "$(dirname "$0")/.."
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Inumpywren_amd/csrc"
mkdir -p tools/dbg/qs
for f in numpywren_amd/csrc/*.hip; do $H $F -DNPW_QR_STAMPS -c $f -o tools/dbg/qs/$(basename $f .hip).o 2>/dev/null; done
$H --offload-arch=gfx950 -shared -fPIC tools/dbg/qs/*.o -o tools/dbg/qs/libnpw_hip.so
$H $F tools/qr_stamps.hip -Ltools/dbg/qs -lnpw_hip -Wl,-rpath,'$ORIGIN/qs' -o tools/dbg/qr_stamps 2>/dev/null
echo "built tools/dbg/qr_stamps"
