#!/bin/bash
# Developer aid: where a column of the QR panel kernel spends its time (workgroup 0; -DNPW_QR_STAMPS build of qr.hip
# into tools/dbg/, the product library is not touched).    bash tools/qr_stamps.sh   (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/dbg/stamps
cd $R/numpywren_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNPW_QR_STAMPS -I$R/include -I. -c $f -o $R/tools/dbg/stamps/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/tools/dbg/stamps/*.o -o $R/tools/dbg/stamps/libnpw_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include $R/tools/qr_stamps.hip -L$R/tools/dbg/stamps -lnpw_hip -Wl,-rpath,$R/tools/dbg/stamps -o $R/tools/dbg/stamps/qr_stamps
$R/tools/dbg/stamps/qr_stamps
