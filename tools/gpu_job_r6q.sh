cd ${GRAFT_REPO_ROOT:?}
for args in "19 rows 0" "19 tiled 0" "3 rows 1" "3 tiled 1" "19 rows 0 18"; do echo "== $args"; python abtest/edge_diag.py $args 2>&1 | grep -v amdgpu.ids | tail -17; done
