# the PSD cone kernels of cones beyond 64 (n = 72 / 96 / 128): tests, then a rocprofv3 kernel summary:  bash tools/r05_psd96.sh <tag>
TAG=${1:-r05_psd}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q -k "psd or c5_24 or e2e_reference or device_resident_ipm" > $O/${TAG}_pytest.log 2>&1
tail -4 $O/${TAG}_pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o p -- python -m pytest $R/tests -m gpu -x -q -k "psd_cone_operations and 96 or psd_large_cone or (c5_psd_scaling and 72)" > $O/${TAG}_prof.log 2>&1
f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/${TAG}_n96_kernel_stats.csv && grep -E "k_psd" $f | cut -d, -f1-4 | sed 's/chip::dev::(anonymous namespace):://' | cut -c1-150
rm -rf $O/${TAG}_prof
