cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export DET6D_KNOBS_LIB=1
DET6D_ROWS_RESIDENT=1 python3 scripts/r06/dbg_rows.py 32768 /tmp/a 2>&1 | grep -v amdgpu.ids
DET6D_ROWS_RESIDENT=2 python3 scripts/r06/dbg_rows.py 32768 /tmp/b 2>&1 | grep -v amdgpu.ids
python3 - <<'PY'
import numpy as np
ra, rb = np.load('/tmp/a_rows.npy'), np.load('/tmp/b_rows.npy')
sa, sb = np.load('/tmp/a_scores.npy'), np.load('/tmp/b_scores.npy')
print('agg rows equal', np.array_equal(ra, rb), 'mismatch rows', np.unique(np.nonzero(ra != rb)[0])[:40], 'cols', np.unique(np.nonzero(ra != rb)[1])[:70])
bad = np.nonzero(sa[:, 0] != sb[:, 0])[0]
print('scores mismatches', len(bad), bad[:64])
print(sa[:40, 0]); print(sb[:40, 0])
PY
