cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/pcie_inclusive.py 22 | cut -c1-700
python tools/pcie_inclusive.py 26 | cut -c1-700
timeout 2900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python3 tools/stress_modes.py 200 5 2>&1 | tail -2
python3 tools/stress.py 60 2>&1 | tail -2
