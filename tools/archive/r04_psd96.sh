# the PSD cone kernels at n = 96 with and without the matrix-core products:  bash tools/r04_psd96.sh <tag>
TAG=${1:-r04_psd}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "psd or c5_24 or genpow or e2e_reference" > $O/${TAG}_pytest.log 2>&1
tail -4 $O/${TAG}_pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for mode in mfma scalar; do
  rm -rf $O/${TAG}_$mode
  if [ $mode = scalar ]; then export CHIP_NO_PSD_MFMA=1; else unset CHIP_NO_PSD_MFMA; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_$mode -o p -- python -m pytest $R/tests -m gpu -x -q -k "psd_cone_operations and 96 or psd_large_cone" > $O/${TAG}_$mode.log 2>&1
  f=$(find $O/${TAG}_$mode -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_n96_${mode}_kernel_stats.csv && grep -E "k_psd" $f | cut -d, -f1-4 | sed 's/chip::dev::(anonymous namespace):://' | cut -c1-150
  rm -rf $O/${TAG}_$mode
done
