# Round 4: one rank's task of a 2^26 job on one GPU, every candidate layout, scalars resident and from host memory.
for host in 0 1; do
  for w in 2 4 8; do
    for R in 1 2 4 8; do
      if [ $((w % R)) -eq 0 ]; then HOST=$host RANGES=$R python3 tools/shard_probe.py 26 $w 0 10 2>&1 | tail -1; fi
    done
    echo -n "pick: "; HOST=$host python3 tools/shard_probe.py 26 $w 0 10 2>&1 | tail -1
  done
done
