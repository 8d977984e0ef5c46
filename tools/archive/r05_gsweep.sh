# persistent sweeps (k_snode_gsweep): tests, then same-box A/B on configs 2 and 5:  bash tools/r05_gsweep.sh <tag>
TAG=${1:-r05_gs}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "persistent_sweeps or supernode_substitution or paired_solves or chain_supernodes or structure_fuzz" > $O/${TAG}_pytest.log 2>&1
tail -5 $O/${TAG}_pytest.log | cut -c1-300
bash tools/r05_ab.sh $TAG c2 "CHIP_NO_SWEEP_PERSIST"
bash tools/r05_ab.sh $TAG c5 "CHIP_NO_SWEEP_PERSIST"
