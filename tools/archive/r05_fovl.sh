# bundle contributions beside the supernode chain: tests, then same-box A/B on configs 2 and 5:  bash tools/r05_fovl.sh <tag>
TAG=${1:-r05_fo}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "chain_supernodes or supernode_substitution or paired_solves or structure_fuzz or ancestor_updates or persistent_sweeps or c5_24 or e2e_reference" > $O/${TAG}_pytest.log 2>&1
tail -5 $O/${TAG}_pytest.log | cut -c1-300
bash tools/r05_ab.sh $TAG c2 "CHIP_NO_FACTOR_OVERLAP"
bash tools/r05_ab.sh $TAG c5 "CHIP_NO_FACTOR_OVERLAP"
