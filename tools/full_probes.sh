# The check campaigns at their full sizes (the -m gpu suite runs reduced versions): every NTT size up to 2^26, the dense walk over MSM
# sizes up to 2^24, the skewed scalar distributions, random NTT call sequences, the mode stress.  ~10 minutes of a box.
cd $GRAFT_REPO_ROOT
echo "== ntt_sizes_probe 26";  timeout 900 python3 tests/probes/ntt_sizes_probe.py 26 2>&1 | tail -4
echo "== msm_sizes_probe 24 21"; timeout 900 python3 tests/probes/msm_sizes_probe.py 24 21 2>&1 | tail -4
echo "== msm_skew_probe 22 19"; timeout 600 python3 tests/probes/msm_skew_probe.py 22 19 2>&1 | tail -4
echo "== ntt_monkey 200 5";   timeout 600 python3 tests/probes/ntt_monkey.py 200 5 2>&1 | tail -3
echo "== stress_modes 1500 43"; timeout 900 python3 tests/probes/stress_modes.py 1500 43 2>&1 | tail -3
echo "== oom_probe";          timeout 600 python3 tests/probes/oom_probe.py 2>&1 | tail -3
echo "== teardown_probe";     timeout 300 python3 tests/probes/teardown_probe.py 2>&1 | tail -3
