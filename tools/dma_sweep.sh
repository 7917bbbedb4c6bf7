# Dev tool: DMA-mode (host buffers) timings against the number of pieces a task is enqueued in (BLAZE_MSM_PIECES).
# usage: bash tools/dma_sweep.sh <logn> <pieces...>
logn=$1; shift
for k in "$@"; do
  echo "== 2^$logn pieces=$k"
  BLAZE_MSM_PIECES=$k python tools/pcie_inclusive.py $logn 2>&1 | tail -1
done
