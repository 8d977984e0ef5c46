# entry-parallel bundle factorisation on unfused handles: tests, then same-box A/B on configs 2 and 5:  bash tools/r05_bff.sh <tag>
TAG=${1:-r05_bff}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "bundle_sweeps_entry or c2_ or c5_ or chain_supernodes or supernode_substitution or persistent_sweeps or paired_solves or e2e_reference or structure_fuzz or faer or refactor or fallback" > $O/${TAG}_pytest.log 2>&1
tail -5 $O/${TAG}_pytest.log | cut -c1-300
bash tools/r05_ab.sh $TAG c2 "CHIP_NO_FACTOR_FLAT"
bash tools/r05_ab.sh $TAG c5 "CHIP_NO_FACTOR_FLAT"
