# What the config-4 distance GEMM costs WITHOUT writing its 4 GB matrix (upper bound of what a fused distmat + ranking launch
# could save): a second build of the library with the epilogue's stores compiled out, timed against the shipped one.
cd "$(dirname "$0")/.." && mkdir -p /tmp/nostore && cp -r ieee_amd/csrc include /tmp/nostore/ 2>/dev/null
mkdir -p /tmp/nostore/ieee_amd && rm -rf /tmp/nostore/ieee_amd/csrc && mv /tmp/nostore/csrc /tmp/nostore/ieee_amd/csrc
(cd /tmp/nostore/ieee_amd/csrc && rm -f evaluator.o && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DIEEE_DIST_NOSTORE_PROBE -c evaluator.hip -o evaluator.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/nostore/lib_nostore.so *.o)
for i in 1 2; do
  echo "shipped:"; python scripts/distmat_probe.py 2>/dev/null | tail -1
  echo "no stores:"; IEEE_AMD_LIB=/tmp/nostore/lib_nostore.so python scripts/distmat_probe.py 2>/dev/null | tail -1
done
