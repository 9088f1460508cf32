cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 scripts/r06/dbg_coop.py 2>&1 | grep -v amdgpu.ids
