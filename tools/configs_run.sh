export CHECK=0 REPS=3
f() { "$@" 2>&1 | grep -E "rep 2|^B" | cut -c1-330; }
echo "== BLS377 2^26"; CURVE=BLS377 f timeout 600 python tests/probes/gpu_big.py 26
echo "== BN254 2^26"; CURVE=BN254 f timeout 600 python tests/probes/gpu_big.py 26
echo "== BN254 pf=8 2^26"; CURVE=BN254 PF=8 f timeout 900 python tests/probes/gpu_big.py 26
echo "== BLS381 pf=8 2^24"; PF=8 f timeout 600 python tests/probes/gpu_big.py 24
echo "== BLS381 2^22 2^23"; f timeout 600 python tests/probes/gpu_big.py 22 23
echo "== pcie 22"; timeout 600 python tools/pcie_inclusive.py 22
echo "== pcie 26"; timeout 900 python tools/pcie_inclusive.py 26
