R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_probe $R/tools/micro/mfma_f64_probe.hip 2> /dev/null || exit 1
/tmp/mfma_f64_probe | tee $O/mfma_f64_probe.txt
